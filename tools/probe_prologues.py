"""forward trunk conv (3x3 64->64 at (16, 96, 96), bf16 build) timed per input prologue: how much of the launch is the producers'
transform arithmetic (NONE: raw copy; ACT; AFFINE_ACT; + statistics on / off)"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
E, L = bench.sub('engine'), bench.sub('_lib')
E.set_precision('bf16')
dev = torch.device('cuda', 0)
B, LR = 16, 96
torch.manual_seed(1)
w = (torch.rand(64, 64, 3, 3, device=dev) - 0.5) * 0.1
bias = torch.zeros(64, device=dev)
class Ref: pass
ref = Ref(); ref.weight, ref.bias, ref.u, ref.v, ref.geom = w, bias, None, None, E.ConvGeom(64, 64, 3, 1, 1)
p = E.prepare_weights([(ref, B, LR, LR)], training=True)[0][0]
x = (torch.rand(B, LR, LR, 64, device=dev) * 2 - 1).to(torch.bfloat16)
sc, sh = torch.rand(64, device=dev) + 0.5, torch.rand(64, device=dev) - 0.5
slope = torch.full((1,), 0.25, device=dev)
out = torch.empty_like(x)
ops = {'none': E.Operand.plain(x), 'act': E.Operand.act(x, slope), 'affine_act': E.Operand.affine_act(x, sc, sh, slope)}
for stats in (True, False):
    for name, op in ops.items():
        ms = bench._time_launches(lambda: E.conv_forward(p, op, bias=bias, stats=stats, out=out), 60)
        print('%-11s stats=%d  %.2f us' % (name, stats, ms * 1e3))
