"""Barrier-wait accounting of conv_trunk_fwd_kernel (developer tool): which role waits for the other in the PRODUCTION
schedule (no per-phase stamps, so nothing drains the LDS queue inside the loop).
build:  make -C single-image-super-resolution_amd/csrc acct
run:    SISR_LIB=.../libsisr_hip_acct.so python tools/barrier_acct.py"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import bench
dev = torch.device('cuda', 0)
if os.environ.get('ROLE') == 'wgrad_f32':
    r = bench.kernel_rooflines(dev, os.environ.get('SISR_PRECISION', 'fp32'), iters=20, only=('wgrad',))[0]
    print('fp32 wgrad + slab reduce launch %.2f us (HIP events)' % (r['launch_ms'] * 1e3))
    torch.cuda.synchronize()
    L = C.CDLL(os.environ['SISR_LIB'])
    buf = np.zeros(512 * 8, dtype=np.uint64)
    assert L.sisr_wfacct_read(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
    b = buf.reshape(512, 8).astype(np.float64)
    b = b[b[:, 0] > 0]
    md = lambda v: float(np.median(v))
    print('%d workgroups; cycles (median): consumer tile loop %.0f (at barriers %.0f); producer loop %.0f (at barriers %.0f)' % (
        len(b), md(b[:, 0]), md(b[:, 1]), md(b[:, 2]), md(b[:, 3])))
    print('  prologue barrier wait %.0f; slab store + bias tail after the loop %.0f' % (md(b[:, 5] - b[:, 4]), md(b[:, 7] - b[:, 6])))
    sys.exit(0)
r = bench.kernel_rooflines(dev, 'bf16', iters=20, only=('fwd',))[0]
print('fwd launch %.2f us (HIP events)' % (r['launch_ms'] * 1e3))
torch.cuda.synchronize()
L = C.CDLL(os.environ['SISR_LIB'])
buf = np.zeros(512 * 4, dtype=np.uint64)
assert L.sisr_bacct_read(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
b = buf.reshape(512, 4).astype(np.float64)
b = b[b[:, 0] > 0]
print('%d workgroups; cycles (median): consumer loop %.0f, of which at barriers %.0f (%.0f %%); producer loop %.0f, of which at barriers %.0f (%.0f %%)' % (
    len(b), np.median(b[:, 0]), np.median(b[:, 1]), 100 * np.median(b[:, 1] / b[:, 0]),
    np.median(b[:, 2]), np.median(b[:, 3]), 100 * np.median(b[:, 3] / b[:, 2])))
