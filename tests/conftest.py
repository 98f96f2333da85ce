import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A per-test time limit for the GPU tests (pytest-timeout, thread method: dumps every thread's stack, names the test and ends
    the run): a kernel that never returns then fails ONE named test instead of holding the box until the caller's limit.  The
    slowest GPU test takes ~10 s; NGPDE_TEST_TIMEOUT overrides the 300 s."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    limit = float(os.environ.get("NGPDE_TEST_TIMEOUT", "300"))
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(limit, method="thread"))


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(autouse=True)
def _seeded(request):
    """Every test draws its torch random inputs from a seed derived from its own id (+ NGPDE_TEST_SEED, default 0): a failure
    reproduces by name, and `NGPDE_TEST_SEED=k pytest ...` sweeps other draws."""
    import os
    import zlib
    try:
        import torch
    except ImportError:
        yield
        return
    torch.manual_seed((zlib.crc32(request.node.nodeid.encode()) + int(os.environ.get("NGPDE_TEST_SEED", "0"))) & 0x7FFFFFFF)
    yield
