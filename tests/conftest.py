import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def built_lib():
    """libchordvis.so, built in-tree if missing (hipcc cross-compiles without a GPU)."""
    from chord_amd import build
    if not os.path.exists(build.LIB):
        build.build(verbose=False)
    from chord_amd import lib
    return lib


@pytest.fixture(scope="session")
def gpu(built_lib):
    if not _gpu_available():
        pytest.fail("test marked gpu but no GPU is visible: the product path has no CPU fallback")
    return built_lib
