#!/usr/bin/env python3
"""Prints the host-boundary legs of bench lines kept as gpurun_out/pipe/r<records>.json (STAG_PIPE_RECORDS sweeps)."""
import json
import sys
for r in sys.argv[1:]:
    d = json.load(open("gpurun_out/pipe/r%s.json" % r))
    t = d["a2m_file"]["host_to_host_trimmed"]
    print(r, "trimmed", t["value"], t["ms_per_step"], t["equals_device_resident_results"], "a2m_file", d["a2m_file"]["value"],
          "full text", d["host_to_host"]["value"], d["host_to_host"]["equals_device_resident_results"])
