import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the fp64 CPU oracle is the slow part of the GPU parity tests: a GPU box exposes all host cores (256 logical) to every
    # job on it, and torch's default of one thread per visible core oversubscribes a shared host badly (a 67 s suite was
    # seen to take > 7 min).  16 threads = the CPU share of one GPU slot; QEA_TEST_THREADS overrides.
    try:
        import torch
        n = int(os.environ.get("QEA_TEST_THREADS", "0")) or min(16, os.cpu_count() or 1)
        torch.set_num_threads(n)
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
