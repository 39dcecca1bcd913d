"""GPU: the fused multi-tensor Adam against torch.optim.Adam -- the optimizer the reference
instantiates (config.py:292-294) -- and against the oracle restatement, over several steps with the reference's
LambdaLR decay (config.py:170-180).  Tolerance 1e-6 relative per tensor (same fp32 formula, different fma grouping)."""
import copy

import numpy as np
import pytest
import torch

from gpu_helpers import maxrel, pkg

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 3, 9, 9), (64,), (64, 64, 3, 3), (1,), (256, 64, 3, 3), (3, 64, 3, 3), (1027,), (5, 7)]
    return [torch.nn.Parameter((torch.rand(s, generator=g) - 0.5).cuda()) for s in shapes]


@pytest.mark.parametrize('wd', [0.0, 0.01])
def test_fused_adam_matches_torch_adam_and_the_oracle(wd):
    from oracle import optim as ooptim
    opt_mod = pkg('optim')
    pa, pb = _params(1), _params(1)
    base_lr, ratio, total = 1e-3, 0.1, 50
    oa = opt_mod.Adam(pa, lr=base_lr, betas=(.9, 0.999), weight_decay=wd)
    ob = torch.optim.Adam(pb, lr=base_lr, betas=(.9, 0.999), weight_decay=wd)
    f = ratio ** (1 / total)
    sa = torch.optim.lr_scheduler.LambdaLR(oa, lr_lambda=lambda it: f ** it)
    sb = torch.optim.lr_scheduler.LambdaLR(ob, lr_lambda=lambda it: f ** it)
    po = [p.detach().cpu().numpy().copy() for p in pa]
    mo = [np.zeros_like(x) for x in po]
    vo = [np.zeros_like(x) for x in po]
    gen = torch.Generator().manual_seed(7)
    for it in range(6):
        grads = [(torch.rand(p.shape, generator=gen) - 0.5) * (10.0 ** ((it % 3) - 1)) for p in pa]
        for p, q, g in zip(pa, pb, grads):
            p.grad, q.grad = g.cuda(), g.cuda()
        if it == 3:                     # a parameter without gradient is skipped (its step count lags behind)
            pa[2].grad = pb[2].grad = None
        lr_now = ooptim.lambda_lr(base_lr, ratio, total, it)
        assert abs(oa.param_groups[0]['lr'] - lr_now) < 1e-12
        oa.step(); ob.step(); sa.step(); sb.step()
        for k, g in enumerate(grads):
            if it == 3 and k == 2:
                continue
            t = int(oa.state[pa[k]]['step'].item())
            po[k], mo[k], vo[k] = ooptim.adam_step(po[k], g.numpy(), mo[k], vo[k], t, lr_now, weight_decay=wd)
    for k, (p, q) in enumerate(zip(pa, pb)):
        assert maxrel(p, q) < TOL, k
        assert maxrel(p, torch.from_numpy(po[k])) < 5e-6, k
        assert maxrel(oa.state[p]['exp_avg'], ob.state[q]['exp_avg']) < TOL
        assert maxrel(oa.state[p]['exp_avg_sq'], ob.state[q]['exp_avg_sq']) < TOL


def test_fused_adam_state_dict_is_interchangeable_with_torch_adam():
    """the reference checkpoints optimizer state (utils.py:108-115) and reloads it (config.py:296-301)"""
    opt_mod = pkg('optim')
    pa, pb = _params(2), _params(2)
    oa, ob = opt_mod.Adam(pa, lr=1e-4), torch.optim.Adam(pb, lr=1e-4)
    for p, q in zip(pa, pb):
        p.grad = torch.ones_like(p)
        q.grad = torch.ones_like(q)
    oa.step(); ob.step()
    sd_a, sd_b = oa.state_dict(), ob.state_dict()
    assert sd_a['state'].keys() == sd_b['state'].keys()
    assert set(sd_a['state'][0].keys()) == set(sd_b['state'][0].keys()) == {'step', 'exp_avg', 'exp_avg_sq'}
    ob.load_state_dict(sd_a)            # ours -> torch
    oa.load_state_dict(sd_b)            # torch -> ours
    for p, q in zip(pa, pb):
        p.grad = torch.full_like(p, 0.5)
        q.grad = torch.full_like(q, 0.5)
    oa.step(); ob.step()
    for p, q in zip(pa, pb):
        assert maxrel(p, q) < TOL


def test_fused_adam_load_state_dict_with_static_gradient_tensors():
    """step, load_state_dict, step again with the SAME .grad tensors (the HIP-graph replay situation: parameter and
    gradient addresses never change): the loaded moments must be the ones the next step reads and writes"""
    opt_mod = pkg('optim')
    pa, pb = _params(3), _params(3)
    oa, ob = opt_mod.Adam(pa, lr=1e-3), torch.optim.Adam(pb, lr=1e-3)
    ga = [torch.full_like(p, 0.25) for p in pa]            # static gradient tensors, rewritten in place
    for p, q, g in zip(pa, pb, ga):
        p.grad, q.grad = g, g.clone()
    oa.step(); ob.step()
    donor = torch.optim.Adam(_params(4), lr=1e-3)          # a different optimizer history to load
    for p in donor.param_groups[0]['params']:
        p.grad = torch.rand(p.shape, generator=torch.Generator().manual_seed(5)).cuda()
    donor.step(); donor.step()
    sd = donor.state_dict()
    # (torch hands the SAME 'step' tensor object to every optimizer that loads one state_dict: load copies)
    oa.load_state_dict(copy.deepcopy(sd)); ob.load_state_dict(copy.deepcopy(sd))
    for g, q in zip(ga, pb):
        g.fill_(-0.5)                                      # same tensors, new values
        q.grad.fill_(-0.5)
    oa.step(); ob.step()
    for p, q in zip(pa, pb):
        assert maxrel(p, q) < TOL
        assert maxrel(oa.state[p]['exp_avg'], ob.state[q]['exp_avg']) < TOL
        assert maxrel(oa.state[p]['exp_avg_sq'], ob.state[q]['exp_avg_sq']) < TOL


def test_fused_adam_refuses_cpu_parameters():
    opt_mod = pkg('optim')
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        opt_mod.Adam([p]).step()
