import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
from gpu_helpers import pkg
from helpers import load_case
from test_gpu_generator import build
E = pkg('engine'); GE = pkg('generator_engine')
E.set_precision('bf16')
z, cfg, state, grads, after = load_case('gen_x2_sn_w64')
res = {}
for persist in (True, False):
    if persist: os.environ.pop('SISR_BF16_NO_PERSIST', None)
    else: os.environ['SISR_BF16_NO_PERSIST'] = '1'
    net = build(cfg); net.load_state_dict(state, strict=True); net = net.cuda().train()
    x = torch.from_numpy(z['x']).cuda()
    out, sv = GE.run_forward(net._topology(), x, True)
    torch.cuda.synchronize()
    b = sv.blocks[0]
    res[persist] = dict(t0=sv.t0_pre, c1=b.c1, k1=b.k1, c2=b.c2, k2=b.k2, ce=sv.ce, ke=sv.ke, t=sv.t, pre=sv.stage_pre[0], out=out,
                        var=[p.plans[0].plan.variant for p in sv.P.values()])
print(res[True]['var'], res[False]['var'])
for k in ['t0', 'c1', 'k1', 'c2', 'k2', 'ce', 'ke', 't', 'pre', 'out']:
    a, b = res[True][k].float().cpu(), res[False][k].float().cpu()
    print('%-4s shape %-22s maxdiff %.3e  (max %.3f)' % (k, tuple(a.shape), float((a - b).abs().max()), float(b.abs().max())))
