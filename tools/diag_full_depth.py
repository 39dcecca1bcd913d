"""the 16-block golden case (B2, LR 16) in both fp32-tensor builds against the reference's vectors AND against the oracle in fp64:
where a 1e-3 miss on a gradient comes from (a PReLU mask flip moves the gradients in its receptive field)"""
import json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
from gpu_helpers import pkg
from helpers import GOLDEN, load_sampled_case, rel_err, oracle_fwd_bwd
from test_oracle_golden import generator_shapes
from test_gpu_generator import build
E = pkg('engine')
cfg0 = json.loads(str(np.load(os.path.join(GOLDEN, 'gen_x2_sn_16blocks.npz'))['cfg']))
z, cfg, state, after, sample = load_sampled_case('gen_x2_sn_16blocks', generator_shapes(cfg0))
x0, r0 = torch.from_numpy(z['x']), torch.from_numpy(z['r'])
st64 = {k: (v.double() if v.is_floating_point() else v) for k, v in state.items()}
o64 = oracle_fwd_bwd(cfg, st64, x0.double(), r0.double())
o32 = oracle_fwd_bwd(cfg, state, x0, r0)
print('oracle fp32 vs fp64: out %.2e grad_x %.2e' % (rel_err(o32[0], o64[0]), rel_err(o32[1], o64[1])))
print('reference golden vs oracle fp64: out %.2e grad_x %.2e' % (rel_err(torch.from_numpy(z['out']), o64[0]), rel_err(torch.from_numpy(z['grad_x']), o64[1])))
for prec in ('fp32', 'bf16x3'):
    E.set_precision(prec)
    net = build(cfg); net.load_state_dict(state, strict=True); net = net.cuda().train()
    x = x0.cuda().requires_grad_(True)
    out = net(x)
    (out * r0.cuda()).sum().backward()
    gx = x.grad.cpu()
    d = (gx.double() - o64[1]).abs()
    print('%-7s vs golden: out %.2e grad_x %.2e | vs fp64: out %.2e grad_x %.2e (elements of grad_x off by > 1e-4 of max: %d of %d)' % (
        prec, rel_err(out.detach().cpu(), z['out']), rel_err(gx, z['grad_x']), rel_err(out.detach().cpu(), o64[0]), rel_err(gx, o64[1]),
        int((d > 1e-4 * o64[1].abs().max()).sum()), d.numel()))
