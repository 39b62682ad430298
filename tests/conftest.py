import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gsx_lib():
    """libgsx.so, built in-tree by __graft_entry__.build(); GPU tests fail loudly if it is missing."""
    from wgpu_3dgs_viewer_app_amd import _lib

    return _lib.load()


def pytest_collection_finish(session):
    """GPU runs: bring PyTorch's HIP context up BEFORE libgsx creates its first viewer.  The other order works too — until a
    test file that needs torch happens to run after gigabytes of libgsx allocations, where torch's lazy initialisation was
    seen to fail with "no ROCm-capable device" (observed when running test_gpu_speculation.py before test_gpu_overflow.py)."""
    if any(item.get_closest_marker("gpu") for item in session.items):
        try:
            import torch

            if torch.cuda.is_available():
                torch.cuda.init()
                torch.zeros(1, device="cuda")
        except Exception:  # noqa: BLE001 — no torch / no GPU: the GPU tests will say so themselves
            pass
