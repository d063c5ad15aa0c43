import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


def golden_index():
    with open(os.path.join(GOLDEN_DIR, "index.json")) as f:
        return {c["name"]: c for c in json.load(f)["cases"]}


def golden_arrays(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


_DT = {"float16": torch.float16, "bfloat16": torch.bfloat16, "float32": torch.float32,
       "float64": torch.float64}


def golden_inputs(meta):
    """Regenerate the inputs of a golden case from its (kind, seed) recipe and
    check them against the recorded SHA-256, so a drifting RNG is caught."""
    import hashlib
    from oracle import crossclr_oracle as orc
    v, t = orc.make_inputs(meta["kind"], meta["B"], meta["D"], meta["seed"], _DT[meta["dtype"]])
    if meta.get("mutate") == "zero_row":
        v[min(3, meta["B"] - 1)] = 0
    if meta.get("mutate") == "tiny_row":
        v[min(3, meta["B"] - 1)] *= 1e-14 / float(v[min(3, meta["B"] - 1)].double().norm())
    h = hashlib.sha256()
    for x in (v, t):
        h.update(x.contiguous().view(torch.uint8).numpy().tobytes())
    assert h.hexdigest() == meta["input_sha256"], "synthetic input generator drifted"
    return v, t


@pytest.fixture(scope="session")
def goldens():
    return golden_index()
