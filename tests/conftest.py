import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The CPU oracle (torch) would start one thread per LOGICAL cpu of the host (128+ on a GPU box whose share is 16 cores):
    # on the small test problems that oversubscription made the oracle, not the device, the bulk of the suite's time
    # (three-scale oracle step: 60 s with 128 threads).
    try:
        import torch
        torch.set_num_threads(min(16, os.cpu_count() or 16))
    except ImportError:
        pass


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
