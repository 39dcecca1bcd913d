import sys, os, json
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device('cuda', 0)
for role in ('fwd', 'dgrad', 'wgrad'):
    r, _ = bench.kernel_rooflines(dev, 'bf16', iters=40, only=(role,))
    print(role, r['launch_ms'], r['achieved'], r['unit'])
