import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "low-cost-mocap_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def core():
    """The HIP core through its C ABI; fails loudly (never falls back) when missing on a GPU box."""
    # PyTorch bundles its own libamdhip64 with the same SONAME as /opt/rocm's: whichever is loaded first
    # serves both.  A process that uses both must load torch first (bench.py does); tests that touch
    # torch.cuda / RCCL later in the session would otherwise find torch bound to the other runtime.
    import torch  # noqa: F401
    from mocap_core import capi
    return capi.MocapCore(device_id=0)
