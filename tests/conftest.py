import os
import sys

import pytest

os.environ.setdefault("STAG_AMD_CACHE", "off")      # tests leave nothing in ~/.cache; the cache test sets its own
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# The scenarios that drive classify() / classify_genome() end to end take this fixture: "oracle" swaps the
# GPU engine for the oracle-backed stand-in of tests/fake_engine.py (host logic, runs without a GPU), "gpu"
# leaves the real stag_amd.engine.Database in place and carries the gpu mark, so `pytest -m gpu` runs the
# very same scenarios on the HIP kernels.
RTOL = {"oracle": 1e-12, "gpu": 1e-9}      # north_star: node probabilities within 1e-9 relative


@pytest.fixture(params=["oracle", pytest.param("gpu", marks=pytest.mark.gpu)])
def engine(request, monkeypatch):
    if request.param == "oracle":
        import stag_amd.classify as sc
        from fake_engine import OracleDatabase
        monkeypatch.setattr(sc, "Database", OracleDatabase)
    return request.param
