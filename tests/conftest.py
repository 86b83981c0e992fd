import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the GPU boxes show 256 CPUs but grant a 16-core share: keep torch-CPU (the oracle) within it
    import torch
    from pfst_amd.hostinfo import usable_cpus
    torch.set_num_threads(usable_cpus())


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
