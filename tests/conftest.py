import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


PARITY_BUILDS = ['fp32', 'bf16x3']


@pytest.fixture
def parity_build(request):
    """the two builds held to the 1e-3 parity bar: 'fp32' (exact fp32 matrix instructions) and 'bf16x3' (fp32 tensors, the trunk
    contractions on the bf16 matrix instruction over hi / lo pairs of the fp32 operands); use with
    @pytest.mark.parametrize('parity_build', PARITY_BUILDS, indirect=True)"""
    import importlib
    E = importlib.import_module('single-image-super-resolution_amd.engine')
    E.set_precision(request.param)
    yield request.param
    E.set_precision('fp32')
