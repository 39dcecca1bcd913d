import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'gpu_slow: GPU tests left out of the default -m gpu run (the suite stays under 600 s there): the '
                                       'full-size LR 96 generator comparisons of the frozen split build, whose fp32 twins stay in; '
                                       'run with -m "gpu or gpu_slow" or SISR_SLOW_TESTS=1')


def pytest_collection_modifyitems(config, items):
    """tests marked gpu_slow run only when asked for by name: -m with gpu_slow in it, or SISR_SLOW_TESTS=1"""
    if 'gpu_slow' in (config.getoption('-m') or '') or os.environ.get('SISR_SLOW_TESTS') == '1':
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker('gpu_slow') else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


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
