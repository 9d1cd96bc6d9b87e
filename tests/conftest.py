import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """The oracle's batch-1 forwards are many small torch-CPU ops: on the GPU box's 128-core host the default (one thread per core) makes
    them 5-10 x SLOWER than 16 threads (bench.py's cpu_baseline probe: bf16 0.19 s per step at 16 threads, 4.7 s at 128).  The
    thread count can move the last bits of the oracle's fp32 reductions; every comparison with the oracle carries a tolerance or a margin
    rule, and the golden fixtures are files, so nothing depends on it."""
    try:
        import torch
        if (os.cpu_count() or 1) > 32:
            torch.set_num_threads(16)
    except Exception:
        pass
    yield
