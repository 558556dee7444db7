import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def _has_gpu():
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
    except OSError:
        return False
    n = ctypes.c_int(0)
    return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; when someone runs the whole suite on a machine with no
    # GPU they are skipped instead of failing.
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


@pytest.fixture(scope="session")
def rtmi():
    import raytracer_amd
    raytracer_amd.abi.load()
    return raytracer_amd
