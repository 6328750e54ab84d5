import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available() or os.environ.get('DL_TEST_DRYRUN') == '1':
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
