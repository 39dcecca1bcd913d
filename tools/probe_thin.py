"""Launch time of the thin-layer kernels against the batch size (fixed cost vs per-tile cost): HIP-event timing of the
engine calls for N = 16, 32, 48 at the bench geometry (LR 96, HR 192).  usage: python tools/probe_thin.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from gpu_helpers import pkg, FakeConv, nhwc
E, L = pkg('engine'), pkg('_lib')
os.environ['SISR_STORAGE'] = 'bf16'
E.set_precision('bf16')
dev = 'cuda'
rnd = lambda *s: torch.rand(s, device=dev) * 2 - 1


def timed(fn, iters=int(os.environ.get('PROBE_ITERS', '60'))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for n in ([int(os.environ['PROBE_N'])] if 'PROBE_N' in os.environ else (16, 32, 48)):
    row = []
    # first conv (9x9, 3 -> 64, LR 96): forward + weight gradient
    ref = FakeConv(rnd(64, 3, 9, 9) * 0.1, rnd(64) * 0.1, E.ConvGeom(3, 64, 9, 1, 4))
    p = E.prepare_weights([(ref, n, 96, 96)], training=True, need_dgrad=False)[0][0]
    x = rnd(n, 3, 96, 96)
    x_op = E.Operand.plain(x, dims=(n, 96, 96, 3), mode=L.X_NCHW)
    row.append(('first fwd', timed(lambda: E.conv_forward(p, x_op, bias=ref.bias))))
    g = rnd(n, 96, 96, 64).bfloat16()
    pre = rnd(n, 96, 96, 64).bfloat16()
    dy_op = E.Operand(g, (n, 96, 96, 64), pro=L.PRO_ACT_BWD, x2=pre, slope=torch.tensor([0.25], device=dev))
    row.append(('first wgrad+reduce', timed(lambda: E.conv_wgrad(p, x_op, dy_op))))
    # last conv (3x3, 64 -> 3, HR 192): forward, data gradient, weight gradient
    ref2 = FakeConv(rnd(3, 64, 3, 3) * 0.05, rnd(3) * 0.1, E.ConvGeom(64, 3, 3, 1, 1))
    p2 = E.prepare_weights([(ref2, n, 192, 192)], training=True)[0][0]
    xa = rnd(n, 192, 192, 64).bfloat16()
    xa_op = E.Operand.act(xa, torch.tensor([0.25], device=dev))
    y = E.conv_forward(p2, xa_op, bias=ref2.bias, y_mode=L.Y_NCHW, epi=L.EPI_TANH)[0]
    row.append(('last fwd', timed(lambda: E.conv_forward(p2, xa_op, bias=ref2.bias, y_mode=L.Y_NCHW, epi=L.EPI_TANH))))
    gy = rnd(n, 3, 192, 192)
    dy2 = E.Operand(gy, (n, 192, 192, 3), pro=L.PRO_TANH_BWD, mode=L.X_NCHW, x2=y)
    row.append(('last dgrad', timed(lambda: E.conv_dgrad(p2, dy2))))
    row.append(('last wgrad+reduce', timed(lambda: E.conv_wgrad(p2, xa_op, dy2))))
    print('N=%d  ' % n + '  '.join('%s %.1f us' % r for r in row), flush=True)
