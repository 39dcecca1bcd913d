"""Phase timeline of the persistent trunk kernel (developer tool): SISR_LIB=.../libsisr_hip_trace.so python tools/trace_trunk.py"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import bench
dev = torch.device('cuda', 0)
prec = os.environ.get('SISR_PRECISION', 'bf16')
print(bench.kernel_rooflines(dev, prec, iters=3, only=(os.environ.get('ROLE', 'fwd'),))[0]['launch_ms'])
torch.cuda.synchronize()
L = C.CDLL(os.environ['SISR_LIB'])
slots = 128
buf = np.zeros(512 * slots, dtype=np.uint64)
f32t = prec in ('fp32', 'bf16x3')
reader = L.sisr_cftrace_read if f32t else (L.sisr_wttrace_read if os.environ.get('ROLE', 'fwd') == 'wgrad' else L.sisr_ttrace_read)
assert reader(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
t = buf.reshape(512, slots).astype(np.int64) * 10e-3
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
print('%d workgroups; kernel span %.2f us; lifetime avg %.2f; start spread %.2f' % (len(t), t[:, 63].max() - t0, (t[:, 63] - t[:, 0]).mean(), (t[:, 0] - t0).max()))
if f32t:
    cb = np.zeros(512 * 2, dtype=np.uint64)
    assert L.sisr_cfclk_read(cb.ctypes.data_as(C.c_void_p), C.c_int(cb.size)) == 0
    cb = cb.reshape(512, 2).astype(np.int64)[:len(t)]
    ck = (cb[:, 1] - cb[:, 0]) / ((t[:, 63] - t[:, 0]) * 100.0)
else:
    ck = (t[:, 61] - t[:, 60]) / 10e-3 / ((t[:, 63] - t[:, 0]) * 100.0)      # raw ticks / (us x 100 ticks per us)
print('in-kernel shader clock (s_memtime / s_memrealtime): median %.3f GHz' % (float(np.median(ck)) / 10.0))
print('prologue (weights to registers / first tile staged, barrier) +%.2f' % (t[:, 4] - t[:, 0]).mean())
if not f32t and os.environ.get('ROLE', 'fwd') == 'fwd':
    rel = lambda k: (t[:, k] - t[:, 0]).mean()
    print('  consumer: weight loads issued +%.2f, landed +%.2f; producer (wave 4): set up +%.2f, two tiles issued +%.2f, first tile in LDS +%.2f' % (
        rel(2), rel(58), rel(64), rel(65), rel(66)))
for it in range(8):
    b = 4 + 6 * it
    ok = t[:, b + 5] > 0
    if not ok.any():
        break
    d = lambda i, j: (t[ok, j] - t[ok, i]).mean()
    print('tile %d (n=%d): consumer MFMA %.2f  epilogue %.2f  wait at barrier %.2f' % (
        it, ok.sum(), d(b, b + 2), d(b + 2, b + 4), d(b + 4, b + 5)))
    pb = 64 + b
    okp = ok & (t[:, pb + 5] > 0) & (t[:, pb + 1] > 0)
    if okp.any():
        dp = lambda i, j: (t[okp, j] - t[okp, i]).mean()
        print('          producer (wave 4): issue %.2f  wait + commit %.2f  wait at barrier %.2f' % (
            dp(pb, pb + 1), dp(pb + 1, pb + 4), dp(pb + 4, pb + 5)))
if (t[:, 1] > 0).all():
    print('tail epilogue (after the last barrier) ends at +%.2f, role branches join at +%.2f of the lifetime' % ((t[:, 1] - t[:, 0]).mean(), (t[:, 3] - t[:, 0]).mean()))
print('stats tail +%.2f' % (t[:, 63] - t[:, 3]).mean())
