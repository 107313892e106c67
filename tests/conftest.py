import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the oracle is many small torch CPU ops: on the GPU box's 256 host threads torch's default pool makes each of them slower, not faster
    # (bench.py measured 0.3 utt/s with all 256 against 29 with 32) -- the r14a GPU session ran into its 900 s limit on that alone
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))


def pytest_collection_modifyitems(config, items):
    """Without a device (the build container) the GPU parity tests are skipped instead of failing at their first launch."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no ROCm device: -m gpu tests run on the MI355X box')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
