import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch, torch.nn.functional as F
from gpu_helpers import pkg, FakeConv, nhwc
E, L = pkg('engine'), pkg('_lib')
os.environ['SISR_STORAGE'] = 'bf16'
n, h, w = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (2, 16, 32))]
g_ = torch.Generator().manual_seed(1)
rnd = lambda s, sc=1.0: (torch.rand(s, generator=g_) * 2 - 1) * sc
bf = lambda t: t.bfloat16().float()
x = rnd((n, 3, h, w)); wt = rnd((64, 3, 9, 9), 0.1); b = rnd((64,), 0.1)
pre = bf(F.conv2d(bf(x), wt, b, padding=4)); g = bf(rnd((n, 64, h, w)))
E.set_precision('bf16')
ref = FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(3, 64, 9, 1, 4))
p = E.prepare_weights([(ref, n, h, w)], training=True, need_dgrad=False)[0][0]
x_op = E.Operand.plain(x.cuda(), dims=(n, h, w, 3), mode=L.X_NCHW)
dy_op = E.Operand(nhwc(g).cuda().bfloat16(), (n, h, w, 64), pro=L.PRO_ACT_BWD, x2=nhwc(pre).cuda().bfloat16(), slope=torch.tensor([0.25], device='cuda'))
red = {}
for sw in ('1', '0'):
    os.environ['SISR_THIN'] = sw
    red[sw] = E.conv_wgrad(p, x_op, dy_op).cpu()
d = (red['1'] - red['0']).abs()
print('max', float(d.max()), 'of', float(red['0'].abs().max()))
bad = (d > 0.1).nonzero().flatten()
print('n bad', len(bad))
import collections
c = collections.Counter()
for i in bad.tolist()[:100000]:
    row, co = divmod(i, 64); ky, krow = divmod(row, 28); kx, ci = divmod(krow, 3)
    c[(ky, kx)] += 1
print(sorted(c.items()))
for i in bad.tolist()[:10]:
    row, co = divmod(i, 64); ky, krow = divmod(row, 28)
    print(i, 'ky', ky, 'krow', krow, 'co', co, float(red['1'][i]), float(red['0'][i]))
