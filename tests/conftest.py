import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The CPU oracle is where the GPU suite's time goes.  torch defaults to one thread per core; on the GPU box (128 cores, 2 sockets) its CPU ops stop
# scaling — and start tripping over each other — long before that: bench.py's cpu_baseline runs the same oracle at 1.9 s per pair on 32 threads, the
# suite measured ~12 s per pair at the default.  Same cap here.
if (os.cpu_count() or 1) > 32:
    torch.set_num_threads(32)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in d.files:
        a = d[k]
        out[k] = torch.from_numpy(a) if a.ndim else a.item()
    return out


@pytest.fixture
def golden():
    return load_golden


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
