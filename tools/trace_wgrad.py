"""Phase timeline of the bf16 weight-gradient kernel for the trunk layer (developer tool; see trace_conv.py).
x operand: BatchNorm-apply + PReLU prologue; dy operand: BatchNorm-backward prologue (the in-step configuration)."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import bench
dev = torch.device('cuda', 0)
E = bench.sub('engine'); Lm = bench.sub('_lib')
E.set_precision('bf16')
B, LR = 16, 96
class Ref: pass
ref = Ref()
ref.weight = (torch.rand(64, 64, 3, 3, device=dev) - 0.5) * 0.1
ref.bias = torch.zeros(64, device=dev); ref.u = ref.v = None; ref.geom = E.ConvGeom(64, 64, 3, 1, 1)
preps, keep = E.prepare_weights([(ref, B, LR, LR)], training=True)
x = torch.rand(B, LR, LR, 64, device=dev) * 2 - 1
dy = torch.rand(B, LR, LR, 64, device=dev) * 2 - 1
sc = torch.rand(64, device=dev) + 0.5; sh = torch.rand(64, device=dev) - 0.5
slope = torch.full((1,), 0.25, device=dev)
xop = E.Operand.affine_act(x, sc, sh, slope)
gop = E.Operand(dy, tuple(dy.shape), pro=Lm.PRO_BNBWD, x2=x, pa=sc, pb=sh, pd=sh)
for _ in range(3):
    E.conv_wgrad(preps[0], xop, gop)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    E.conv_wgrad(preps[0], xop, gop)
e1.record(); torch.cuda.synchronize()
print('wgrad + slab_reduce: %.1f us per call' % (e0.elapsed_time(e1) * 100))
g = preps[0].plans[2]
print('plan: TH %d TW %d TN %d tiles %d grid_x %d n_chunk %d NJ %d lds %d' % (g.TH, g.TW, g.TN, g.n_tiles, g.grid_x, g.n_chunk, g.NJ, g.lds_bytes))
L = C.CDLL(os.environ['SISR_LIB'])
n_wg, slots = g.grid_x * g.n_chunk, 32
buf = np.zeros(n_wg * slots, dtype=np.uint64)
assert L.sisr_wtrace_read(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
t = buf.reshape(n_wg, slots).astype(np.int64) * 10e-3
t0 = t[:, 0].min()
print('kernel span %.2f us; workgroup lifetime avg %.2f' % (t[:, 30].max() - t0, (t[:, 30] - t[:, 0]).mean()))
names = ['sync', 'x staged', 'dy staged', 'barrier', 'MFMA done']
for it in range(3):
    for k in range(5):
        a, b = (0 if (it == 0 and k == 0) else 5 * it + k), 5 * it + k + 1
        ok = t[:, b] > 0
        if ok.any():
            print('tile %d %-10s +%.2f us (n=%d)' % (it, names[k], (t[ok, b] - t[ok, a]).mean(), ok.sum()))
qidx = np.arange(n_wg) // g.grid_x
for qq in range(g.n_chunk):
    m = qidx == qq
    print('chunk %d workgroups: tile0 MFMA phase %.2f us, x staged %.2f, dy staged %.2f' % (qq, (t[m, 5] - t[m, 4]).mean(), (t[m, 2] - t[m, 1]).mean(), (t[m, 3] - t[m, 2]).mean()))
print('tiles->end loop  (28-last)   ; part reduce+slab store +%.2f ; bias +%.2f' % ((t[:, 29] - t[:, 28]).mean(), (t[:, 30] - t[:, 29]).mean()))
print('start p0 %.2f p50 %.2f p100 %.2f ; end-of-tiles p50 %.2f' % (*np.percentile(t[:, 0] - t0, [0, 50, 100]), np.percentile(t[:, 28] - t0, 50)))
