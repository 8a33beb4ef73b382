import os
import sys
from pathlib import Path

import pytest

# The checkers (OpenMP C oracle, float64 numpy twin on OpenBLAS) would each start one thread per
# core of a 256-core GPU box and leave them spinning: two oversubscribed pools starving each other
# turned a 2 s test into 270 s on a busy box. Bounded, passive pools; must be set before numpy loads.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("OMP_NUM_THREADS", str(min(32, os.cpu_count() or 1)))
os.environ.setdefault("OPENBLAS_NUM_THREADS", str(min(16, os.cpu_count() or 1)))

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


def pytest_collection_modifyitems(config, items):
    """On a host with no ROCm device node at all (a CPU-only container) the gpu tests are skipped
    instead of failing one by one. On a GPU box nothing is ever skipped: if /dev/kfd exists but no
    device is usable the tests must fail loudly."""
    if os.path.exists("/dev/kfd"):
        return
    skip = pytest.mark.skip(reason="no ROCm device node (/dev/kfd): CPU-only host")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the native pieces exist (hipcc cross-compiles on CPU-only hosts)."""
    import __graft_entry__ as g

    lib = ROOT / "lean-explore_amd" / "libleansearch.so"
    ora = ROOT / "oracle" / "_build" / "liboracle.so"
    if not lib.exists() or not ora.exists():
        g.build()
    yield


@pytest.fixture(scope="session")
def gpu_available():
    from lean_explore_amd import native

    return native.device_count() > 0
