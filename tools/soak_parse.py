#!/usr/bin/env python3
"""Soak of the device front end on a GPU box (not a pytest): for a number of seeds, tests/js/parse_cases.js writes fresh
streams of every kind (clean, malformed, garbage) and parses them with the JavaScript front end; aacg_parse_batch must
return the same records bit for bit.  Usage: python tools/soak_parse.py [seeds=20]"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import aacgpu  # noqa: E402
import test_parse_device as T  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
frames = refused = 0
for seed in range(1, n_seeds + 1):
    d = tempfile.mkdtemp()
    r = subprocess.run(["node", os.path.join(ROOT, "tests", "js", "parse_cases.js"), d, "standard"], capture_output=True, text=True,
                       env=dict(os.environ, AACG_CASE_SEED=str(seed * 2654435761 % (1 << 31))))
    assert r.returncode == 0, r.stdout + r.stderr
    entries, counts = T.codebooks(d)
    for case in json.load(open(os.path.join(d, "manifest.json"))):
        data, table, exp = T.load_case(d, case)
        p = aacgpu.Parser(entries, counts, sample_index=case["sampleIndex"])
        got = p.parse_batch(data, table, case["maxUnits"], case["maxChannels"], case["options"], case["wantTns"])
        T.compare(case, got, exp)
        frames += len(table)
        refused += int((exp["results"]["status"] != 0).sum())
        p.close()
    print("seed %d ok" % seed, flush=True)
print("soak_parse OK: %d seeds, %d frames (%d of them refused, identically)" % (n_seeds, frames, refused))
