import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "tools"), os.path.join(REPO, "tests", "emu")):
    if p not in sys.path:
        sys.path.insert(0, p)


os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # as procgen_amd/__init__.py: before anything below initialises the HIP runtime


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _hip_device_count():
    """number of HIP devices, asked of the runtime directly (no torch import, no library of ours)"""
    import ctypes

    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


def pytest_collection_modifyitems(config, items):
    # a CPU-only box running the files without -m "not gpu": the library is fatal() without a device (there is no fallback), which
    # would take the xdist worker down; skip the gpu-marked tests instead
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if gpu_items and _hip_device_count() == 0:
        skip = pytest.mark.skip(reason="no HIP device (gpu-marked test)")
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")
