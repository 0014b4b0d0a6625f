import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
GLB = os.path.join(GOLDEN, "testroomopt.glb")
ROUTE = os.path.join(GOLDEN, "lange_route.xml")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the libraries when a fresh checkout has none (hipcc cross-compiles without a GPU)."""
    pk = os.path.join(ROOT, "small-project-uv-robot-ray-tracer_amd")
    need = [os.path.join(pk, "libuvrt_hip.so"), os.path.join(pk, "libuvrt_host.so"),
            os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(f) for f in need):
        import __graft_entry__ as g
        g.build()
    yield


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_first(_built):
    """On a GPU box initialise torch's HIP runtime BEFORE libuvrt_hip.so brings in its own
    libamdhip64: a process that initialises HIP through /opt/rocm first leaves torch's bundled
    runtime without devices ("No HIP GPUs are available").  bench.py has the same order."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield


@pytest.fixture(scope="session")
def orc():
    import __graft_entry__ as g
    return g.load_oracle()


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def oscene(orc):
    """The test scene through the ORACLE's loader / floor / BVH restatement."""
    return orc.Scene(GLB)


@pytest.fixture(scope="session")
def oroute(orc):
    return orc.load_route(ROUTE)
