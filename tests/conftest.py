import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# what a caller of the batch pipeline sets: one hardware queue per stream.  Here, before anything touches the GPU:
# tests make native handles long before they make an engine, and the engine's own request (_native.want_hw_queues) would be too late.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load
