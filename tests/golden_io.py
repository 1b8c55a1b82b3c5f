"""Loader for tests/golden/golden.{bin,json} (vectors produced by running the reference,
see tests/golden/gen/gen_golden.js)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_golden():
    with open(os.path.join(HERE, "golden", "golden.json")) as f:
        man = json.load(f)
    blob = np.fromfile(os.path.join(HERE, "golden", "golden.bin"), dtype=np.uint8)
    out = {}
    for name, a in man["arrays"].items():
        dt = np.dtype("<" + a["dtype"])
        n = int(np.prod(a["shape"]))
        out[name] = blob[a["offset"]:a["offset"] + n * dt.itemsize].view(dt).reshape(a["shape"]).copy()
    return out
