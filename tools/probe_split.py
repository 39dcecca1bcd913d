"""trunk conv (3x3 64->64, fp32 tensors, (16, 96, 96)) forward + data-gradient roles: exact fp32 MFMA against the split (hi / lo bf16) MFMA"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
E, L = bench.sub('engine'), bench.sub('_lib')
dev = torch.device('cuda', 0)
B, LR = 16, 96
torch.manual_seed(1)
w = (torch.rand(64, 64, 3, 3, device=dev) - 0.5) * 0.1
bias = torch.zeros(64, device=dev)
class Ref: pass
ref = Ref(); ref.weight, ref.bias, ref.u, ref.v, ref.geom = w, bias, None, None, E.ConvGeom(64, 64, 3, 1, 1)
x = torch.rand(B, LR, LR, 64, device=dev) * 2 - 1
c = torch.rand(B, LR, LR, 64, device=dev) * 2 - 1
sc, sh = torch.rand(64, device=dev) + 0.5, torch.rand(64, device=dev) - 0.5
slope = torch.full((1,), 0.25, device=dev)
outs = {}
for prec in ('fp32', 'bf16x3'):
    E.set_precision(prec)
    p = E.prepare_weights([(ref, B, LR, LR)], training=True)[0][0]
    op = E.Operand.affine_act(x, sc, sh, slope)
    out = torch.empty_like(x)
    ms = bench._time_launches(lambda: E.conv_forward(p, op, bias=bias, stats=True, out=out), 40)
    dy = E.Operand(x, tuple(c.shape), pro=L.PRO_BNACT_BWD, x2=c, pa=sc, pb=sh, pd=sh, ps=sc, pt=sh, slope=slope)
    msd = bench._time_launches(lambda: E.conv_dgrad(p, dy, res=c), 40)
    msw = bench._time_launches(lambda: E.conv_wgrad(p, op, dy), 40)
    outs[prec] = (out.clone(), E.conv_dgrad(p, dy, res=c), E.conv_wgrad(p, op, dy))
    print('%-7s fwd %.2f us  dgrad %.2f us  wgrad + slab sum %.2f us' % (prec, ms * 1e3, msd * 1e3, msw * 1e3))
for i, nm in enumerate(('fwd', 'dgrad', 'wgrad')):
    a, b = outs['fp32'][i], outs['bf16x3'][i]
    print(nm, 'split vs exact: max |diff| / max |ref| = %.2e' % float((a - b).abs().max() / b.abs().max()))
