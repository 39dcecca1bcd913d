"""Launch only the dominant trunk conv (as bench.py's roofline probe does), for rocprofv3 --pmc passes."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import bench
dev = torch.device('cuda', 0)
prec = os.environ.get('SISR_PRECISION', 'bf16')
bench.sub('engine').set_precision(prec)
print(bench.kernel_rooflines(dev, prec, iters=int(os.environ.get('ITERS', '20')), only=(os.environ.get('ROLE', 'fwd'),)))
