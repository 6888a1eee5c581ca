import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("g_"))


def load_golden(name):
    """Load one fixture made by tests/golden/make_golden.py (numpy.load, no pickle)."""
    import numpy as np
    import torch
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    out = {"sd0": {}, "grad0": {}, "sd_final": {}}
    for k in z.files:
        if "/" in k and k.split("/", 1)[0] in out:
            grp, key = k.split("/", 1)
            out[grp][key] = torch.from_numpy(z[k].copy())
        else:
            out[k] = z[k]
    return out


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
