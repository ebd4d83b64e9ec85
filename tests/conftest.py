import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_visible():
    try:
        import ctypes as C

        import relearn_amd as ra
        n = C.c_int32(0)
        return ra.lib().rl_device_count(C.byref(n)) == ra.OK and n.value > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """tests marked `gpu` are skipped (not errored) on a box without a device or without the built library"""
    if any(item.get_closest_marker("gpu") for item in items) and not _gpu_visible():
        skip = pytest.mark.skip(reason="needs a gfx950 device and the built librelearn_hip.so")
        for item in items:
            if item.get_closest_marker("gpu"):
                item.add_marker(skip)


@pytest.fixture(scope="session")
def engine():
    import relearn_amd as ra
    eng = ra.Engine(0)
    yield eng
    eng.close()
