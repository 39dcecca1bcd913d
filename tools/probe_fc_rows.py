"""launch time of sisr_fc_wgrad_rows (the locally formed mean of the classifier head's weight gradient from the gathered factors of N ranks)
beside the single-rank sisr_fc_wgrad it replaces per pass: python tools/probe_fc_rows.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
import torch
from gpu_helpers import pkg
E = pkg('engine')


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for k in (18432, 73728):
    w = torch.empty(1024, k, device='cuda')
    d16, x16 = torch.rand(16, 1024, device='cuda'), torch.rand(16, k, device='cuda')
    t1 = timeit(lambda: E.fc_wgrad_only(d16, x16, w))
    line = 'fc_in %6d (|W| %5.1f MB): one rank alone %7.1f us' % (k, w.numel() * 4 / 1e6, t1)
    for ranks in (2, 4, 8):
        d, x = torch.rand(16 * ranks, 1024, device='cuda'), torch.rand(16 * ranks, k, device='cuda')
        line += '   %d ranks (%3d rows) %7.1f us' % (ranks, 16 * ranks, timeit(lambda: E.fc_wgrad_rows(d, x, w, 1.0 / ranks)))
    print(line)
