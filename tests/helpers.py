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


def analytically_zero(key, ref):
    """parameter gradients that are exactly zero in exact arithmetic: the bias of a conv whose output goes straight
    into a training-mode BatchNorm (generator: block_list.*.layers.{0,3}.bias, block_list_end.0.bias; discriminator:
    conv.2.*.layers.0.bias) -- the mean subtraction removes any constant, so both sides hold rounding noise only.
    Told from the key layout of the gradient dict: `<p>.<i>.bias` is such a bias when `<p>.<i+1>` is a BatchNorm
    (a 1-D `.weight` and a `.bias` of the same length > 1; a PReLU has a 1-element weight and no bias)."""
    if not key.endswith('.bias'):
        return False
    head, _, idx = key[:-len('.bias')].rpartition('.')
    if not idx.isdigit():
        return False
    nxt = '%s.%d' % (head, int(idx) + 1)
    w, b = ref.get(nxt + '.weight'), ref.get(nxt + '.bias')
    return w is not None and b is not None and w.dim() == 1 and w.numel() > 1 and w.numel() == ref[key].numel()


def grads_close(got, ref, tol, floor=0.02):
    """Per-tensor max|got-ref| <= tol * |ref|_inf  -- BASELINE.json's '1e-3 relative fp32', tensor by tensor.
    Only the gradients that are analytically ZERO (analytically_zero()) are held to an absolute bound instead,
    tol * floor * G with G the largest gradient magnitude of the model: there both sides hold only rounding noise of
    size ~eps * sum|dy|."""
    big = max(float(v.abs().max()) for v in ref.values())
    bad = []
    for k, b in ref.items():
        a = got[k]
        if analytically_zero(k, ref):
            scale = max(float(b.abs().max()), floor * big)
        else:
            scale = max(float(b.abs().max()), 1e-30)
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
