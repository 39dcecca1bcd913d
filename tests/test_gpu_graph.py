"""GPU: graph.GraphedStep -- replay equals eager, segmented capture runs host work between segments, and a FAILED
capture (a host synchronisation inside the captured function: what broke round 1's bench, gpurun_out/bench1.log)
leaves the process usable: the next eager launch of the library must succeed instead of inheriting the
invalidated capture's error code."""
import pytest
import torch

from gpu_helpers import pkg

pytestmark = pytest.mark.gpu


def _x():
    return (torch.rand(4, 3, 16, 16, generator=torch.Generator().manual_seed(9)) * 2 - 1).cuda()


def test_failed_capture_leaves_the_process_usable():
    from oracle import ops as oo                      # checker only
    G, ut = pkg('graph'), pkg('utils')
    x = _x()

    def bad():
        y = ut.lr_from_hr(x, (8, 8))
        float(y.sum())                                # host synchronisation: illegal inside a stream capture
        return y
    with pytest.raises(G.GraphCaptureError):
        G.GraphedStep(bad, warmup=1)
    y = ut.lr_from_hr(x, (8, 8))                      # the very next eager launch on the ordinary stream
    torch.cuda.synchronize()
    assert float((y.cpu() - oo.lr_from_hr(x.cpu(), (8, 8))).abs().max()) < 5e-6
    good = G.GraphedStep(lambda: ut.lr_from_hr(x, (8, 8)))       # and capturing still works afterwards
    assert torch.equal(good(), y)


def test_segmented_capture_runs_host_work_between_segments():
    G, ut = pkg('graph'), pkg('utils')
    x = _x()
    seen = []

    def fn():
        a = ut.lr_from_hr(x, (8, 8))
        G.segment_boundary('mid')                     # no-op in the eager warm-up runs, a graph cut while capturing
        b = ut.lr_from_hr(a, (4, 4))
        return a, b
    ea, eb = fn()
    step = G.GraphedStep(fn, between=seen.append)
    assert len(step.graphs) == 2
    a, b = step()
    a2, b2 = step()
    assert seen == ['mid', 'mid'] and torch.equal(a, ea) and torch.equal(b, eb) and torch.equal(b2, eb)
