"""where does the deep data-gradient kernel differ from the reference?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import torch.nn.functional as F
from gpu_helpers import FakeConv, nchw, nhwc, pkg
E, L = pkg('engine'), pkg('_lib')
E.set_precision('bf16')


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


bf = lambda t: t.to(torch.bfloat16).float()
n, cin, cout, h, w = 2, 64, 64, 24, 24
wt = rnd((cout, cin, 3, 3), 2, 0.07)
ref = FakeConv(wt.cuda(), torch.zeros(cout).cuda(), E.ConvGeom(cin, cout, 3, 1, 1))
p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
x = bf(rnd((n, cin, h, w), 31))
xd = nhwc(x).cuda().to(torch.bfloat16)
r = F.conv2d(x.double(), bf(wt).double(), padding=1)
for stats in (False, True, False, False):
    y, _, _ = E.conv_forward(p, E.Operand.plain(xd), bias=None, stats=stats)
    torch.cuda.synchronize()
    err = (nchw(y.float()).cpu().double() - r).abs()
    em = err.amax(dim=1)
    bad = (em > 0.05).nonzero()
    print('forward stats=%s max err %.4f bad pixels %d %s' % (stats, float(err.max()), len(bad), bad[:8].tolist()))
    for b in bad[:2].tolist():
        print('   got', y[b[0], b[1], b[2], :6].float().cpu().tolist(), '\n   ref', r[b[0], :6, b[1], b[2]].tolist())
        # is it some other pixel's value?
        d = (nchw(y.float()).cpu().double()[b[0], :, b[1], b[2]][None, :, None, None] - r).abs().amax(dim=1)
        k = d.flatten().argmin()
        print('   closest reference pixel', divmod(int(k), h * w)[0], divmod(int(k) % (h * w), w), 'dist %.4f' % float(d.flatten()[k]))
