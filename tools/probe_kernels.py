"""live HIP-event probes of the three trunk kernels (bench.kernel_rooflines), one line per role; SISR_PRECISION selects the build"""
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device('cuda', 0)
prec = os.environ.get('SISR_PRECISION', os.environ.get('PRECISION', 'bf16'))
for role in ('fwd', 'dgrad', 'wgrad'):
    r, _ = bench.kernel_rooflines(dev, prec, iters=40, only=(role,))
    print(role, r['launch_ms'], r['achieved'], r['unit'])
