"""Shared helpers for the parity tests (oracle = checker only; see oracle/__init__.py)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

GEN_CASES = ['gen_x2_sn_w64', 'gen_x2_nosn_w16', 'gen_x4_scales22_w32', 'gen_x4_suffix_w32',
             'gen_x8_suffix2_w16']
PROG_CASES = ['prog_x2_w16', 'prog_x8_w64']
DIS_CASES = ['dis_16px_w16']


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    cfg = json.loads(str(z['cfg']))
    state = {k[len('state/'):]: torch.from_numpy(z[k]) for k in z.files if k.startswith('state/')}
    grads = {k[len('grad/'):]: torch.from_numpy(z[k]) for k in z.files if k.startswith('grad/')}
    after = {k[len('after/'):]: torch.from_numpy(z[k]) for k in z.files if k.startswith('after/')}
    return z, cfg, state, grads, after


def rel_err(a, b):
    """max |a-b| / max(|b|_inf, tiny): the '1e-3 relative fp32' metric of BASELINE.json."""
    a = torch.as_tensor(a, dtype=torch.float64).reshape(-1)
    b = torch.as_tensor(b, dtype=torch.float64).reshape(-1)
    if b.numel() == 0:
        return 0.0
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def oracle_forward(cfg, state, x, training=True):
    from oracle import models as om
    if cfg['kind'] == 'generator':
        return om.generator_forward(state, x, cfg['list_scales'], training, cfg['n_suffix'])
    if cfg['kind'] == 'progressive':
        return om.progressive_forward(state, x, cfg['n_suffix'], training)
    if cfg['kind'] == 'discriminator':
        return om.discriminator_forward(state, x, cfg['list_stride'], training)
    raise ValueError(cfg['kind'])


def oracle_fwd_bwd(cfg, state, x, r, training=True):
    """Returns out, grad_x, {param grads}, new_buffers from the CPU oracle."""
    from oracle import models as om
    st = {k: v.clone() for k, v in state.items()}
    pk = om.param_keys(st)
    for k in pk:
        st[k].requires_grad_(True)
    x = x.clone().requires_grad_(True)
    out, new = oracle_forward(cfg, st, x, training)
    (out * r).sum().backward()
    grads = {k: (st[k].grad if st[k].grad is not None else torch.zeros_like(st[k])) for k in pk}
    return out.detach(), x.grad, grads, new


def grads_close(got, ref, tol, floor=0.02):
    """Per-tensor max|got-ref| <= tol * max(|ref|_inf, floor*G), G = largest grad magnitude in
    the model.  The floor exists for gradients that are analytically ZERO (a conv bias feeding a
    training-mode BatchNorm): there both sides hold only rounding noise of size ~eps*sum|dy|."""
    big = max(float(v.abs().max()) for v in ref.values())
    bad = []
    for k, b in ref.items():
        a = got[k]
        scale = max(float(b.abs().max()), floor * big)
        err = float((a.double() - b.double()).abs().max()) / scale
        if not err < tol:
            bad.append((k, err))
    return bad


def load_sampled_case(name, shapes):
    """fixtures of full-depth networks: the state is regenerated from oracle/init.py (`state_seed`), big gradients
    are stored as a strided sample of 4096 entries.  shapes: {state_dict key: shape} of the architecture.
    -> z, cfg, state, sample(fn) where sample(grads) -> (got, ref) dicts restricted / strided like the fixture."""
    from oracle import init as oinit
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    cfg = json.loads(str(z['cfg']))
    state = oinit.synth_state(shapes, cfg['state_seed'])
    after = {k[len('after/'):]: torch.from_numpy(z[k]) for k in z.files if k.startswith('after/')}

    def sample(pg):
        ref, got = {}, {}
        for k in z.files:
            if k.startswith('grad/'):
                ref[k[5:]], got[k[5:]] = torch.from_numpy(z[k]), pg[k[5:]]
            if k.startswith('gradsample/'):
                flat = pg[k[11:]].reshape(-1)
                ref[k[11:]] = torch.from_numpy(z[k])
                got[k[11:]] = flat[:: max(1, flat.numel() // 4096)][:4096]
        return got, ref
    return z, cfg, state, after, sample
