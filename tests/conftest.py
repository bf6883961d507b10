import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'reference: needs /root/reference (build container only)')


def pytest_collection_modifyitems(config, items):
    import torch
    has_gpu = torch.cuda.is_available()
    from oracle import ref_shims
    has_ref = ref_shims.reference_available() and os.path.exists(ref_shims.REF_EXT_SO)
    for item in items:
        if 'gpu' in item.keywords and not has_gpu:
            item.add_marker(pytest.mark.skip(reason='no GPU in this container'))
        if 'reference' in item.keywords and not has_ref:
            item.add_marker(pytest.mark.skip(reason='reference tree / oracle/_ref not available'))


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
