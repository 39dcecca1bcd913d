import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from gpu_helpers import pkg
E = pkg('engine')
import test_gpu_kernels as T
n, cin, cout, k, stride, h, w = (2, 64, 64, 3, 1, 12, 12)
x = T._rand((n, cin, h, w), 1); wt = T._rand((cout, cin, k, k), 2, 0.07); b = T._rand((cout,), 3, 0.1)
y_ref = F.conv2d(x, wt, b, stride=stride, padding=1)
E.set_precision('bf16')
ref = T.FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(cin, cout, k, stride, 1))
preps, keep = E.prepare_weights([(ref, n, h, w)], training=True)
for stats in (False, True):
    out = E.conv_forward(preps[0], E.Operand.plain(T.nhwc(x).cuda()), bias=ref.bias, stats=stats)
    y = out[0]
    yy = T.nchw(y).cpu()
    err = (yy - y_ref).abs()
    print('stats', stats, 'max err', err.max().item(), 'frac bad', (err > 0.05).float().mean().item())
    bad = (err > 0.05).nonzero()
    print(bad[:10].tolist())
    print('per-channel bad', (err > 0.05).float().mean(dim=(0, 2, 3))[:8].tolist(), 'per-row bad', (err > 0.05).float().mean(dim=(0, 1, 3)).tolist())
