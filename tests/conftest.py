import os
import sys

import pytest

try:    # One HIP runtime per process: torch brings its own copy of libamdhip64; loaded first, libcvo_hip.so binds to that copy too.
    import torch  # noqa: F401  (the other order -- /opt/rocm's runtime first, torch's second -- leaves torch without a GPU)
except Exception:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.build()
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def hiplib():
    """The product library; built in-tree, loaded through the C ABI.  GPU tests fail loudly if it is absent."""
    import cvo_slam_amd as ca
    if not os.path.exists(ca.lib_path()):          # fresh checkout: compile it (hipcc cross-compiles gfx950 without a GPU)
        from cvo_slam_amd import build
        build.build()
    ca.load_library()
    return ca
