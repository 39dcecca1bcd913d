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


F32_TENSOR_BUILDS = ['fp32', 'bf16x3']


@pytest.fixture
def f32_build(request):
    """the two builds over fp32 tensors: 'fp32' (exact fp32 matrix instructions: THE 1e-3 parity build) and 'bf16x3' (the trunk
    contractions on the bf16 matrix instruction over hi / lo pairs of the fp32 operands: forward inside 1e-3, gradients held to its
    own measured bounds where a test says so); use with @pytest.mark.parametrize('f32_build', F32_TENSOR_BUILDS, indirect=True)"""
    import importlib
    E = importlib.import_module('single-image-super-resolution_amd.engine')
    E.set_precision(request.param)
    yield request.param
    E.set_precision('fp32')
