"""Helpers shared by the tests: read tests/golden/*.json back into the reference's data shapes."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def bits(s):
    return np.frombuffer(s.encode(), dtype=np.uint8) == ord("1")


def db_from_json(case):
    taxonomy = {k: list(v) for k, v in case["taxonomy"].items()}
    tax_function = {k: np.array([float(x) for x in row], dtype=np.float64)
                    for k, row in case["tax_function"].items()}
    classifiers = {k: (v if isinstance(v, str) else np.array([float(x) for x in v], dtype=np.float64))
                   for k, v in case["classifiers"].items()}
    return taxonomy, tax_function, classifiers


def floats(text, sep="/"):
    return [float(t) for t in text.split(sep)] if text else []


def rows_match(got, exp, rtol=1e-9):
    """Compare two 7-field rows: text fields exactly, probabilities within rtol."""
    assert got[0] == exp[0]
    assert got[1] == exp[1], (got, exp)
    assert got[2] == exp[2], (got, exp)
    assert str(got[3]) == str(exp[3]) and type(got[3]) is type(exp[3]), (got[3], exp[3])
    g, e = floats(got[4]), floats(exp[4])
    assert len(g) == len(e)
    np.testing.assert_allclose(g, e, rtol=rtol, atol=0)
    gl, el = got[5].split("/"), exp[5].split("/")
    assert [t.split(":")[0] for t in gl] == [t.split(":")[0] for t in el]
    np.testing.assert_allclose([float(t.split(":")[1]) for t in gl],
                               [float(t.split(":")[1]) for t in el], rtol=rtol, atol=0)
    assert got[6] == exp[6]


ERROR_OF_STATUS = {1: "KeyError", 2: "TypeError", 3: "ValueError", 4: "IndexError", 5: "ZeroDivisionError"}
