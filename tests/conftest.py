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
