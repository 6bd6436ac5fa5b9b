import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:        # the oracle (torch fp32 on the host) is several times slower with one thread per core of a 128-core box than with 16
        import torch
        torch.set_num_threads(min(16, os.cpu_count() or 16))
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

