import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer test (CPU suite: skipped by nothing; GPU suite: the extra oracle-heavy coverage, e.g. second order for all 8 tasks)")
    # The torch-CPU oracle is a long chain of small ops: on the GPU box's 256 host cores torch's default intra-op pool (one thread per
    # core) makes it 3-5x SLOWER than 16 threads (bench.py's cpu_baseline sweep: 0.24 s per inner step at 16 threads, 0.80 s at 64).
    try:
        import torch
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    except Exception:  # noqa: BLE001
        pass


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
