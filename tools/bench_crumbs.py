"""developer tool: time the small per-layer kernels (BatchNorm finalize / backward finalize / slab reduce) at the
trunk's sizes"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch, bench
E = bench.sub('engine')
dev = torch.device('cuda', 0)
C, NT, P = 64, 1152, 16 * 96 * 96
sp = torch.rand(NT, 2, C, device=dev); cp = torch.full((NT,), 128.0, device=dev)
bn = torch.nn.BatchNorm2d(C).to(dev)
part = torch.rand(NT, 2 * C + 1, device=dev)
x = torch.rand(16, 96, 96, C, device=dev)
k = torch.rand(4, C, device=dev) + 0.5
def t(fn, n=300):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print('bn_finalize %.2f us   bn_bwd_finalize %.2f us' % (t(lambda: E.bn_finalize(sp, cp, bn)), t(lambda: E.bn_backward(x, x, k, bn.weight, part=part))))
