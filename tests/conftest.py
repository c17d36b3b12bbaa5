import os
import sys

import pytest

# torch first: it bundles its own copy of the HIP runtime, and whichever copy initialises the GPU first is the one the
# process can use -- a test that brings torch up after libtaxor_gpu.so has used the GPU finds "No HIP GPUs are available"
# (bench.py and the profile scripts import torch first for the same reason).  The library also reads GPU_MAX_HW_QUEUES
# from the environment the runtime is started with.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
try:
    import torch  # noqa: F401
except Exception:      # a box without torch still runs the C-ABI tests
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
