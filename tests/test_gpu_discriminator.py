"""GPU: golden parity of the Discriminator drop-in (row a5) incl. stride-2 convs, NCHW flatten and
the FC layers, against the vectors captured from the reference module (1e-3 relative fp32)."""
import json
import os

import numpy as np
import pytest
import torch

from gpu_helpers import pkg
from helpers import GOLDEN, DIS_CASES, grads_close, load_case, rel_err
from test_oracle_golden import discriminator_shapes

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.mark.parametrize('name', DIS_CASES)
def test_discriminator_matches_reference_golden(name):
    z, cfg, state, grads, after = load_case(name)
    md = pkg('model_discriminator')
    net = md.Discriminator(tuple(cfg['input_shape']), cfg['list_n_features'], cfg['list_stride'])
    net.load_state_dict(state, strict=True)
    net = net.cuda().train()
    x = torch.from_numpy(z['x']).cuda().requires_grad_(True)
    r = torch.from_numpy(z['r']).cuda()
    out = net(x)
    assert tuple(out.shape) == (x.shape[0], 1)
    assert rel_err(out.detach().cpu(), z['out']) < TOL
    (out * r).sum().backward()
    assert rel_err(x.grad.cpu(), z['grad_x']) < TOL
    got = {k: p.grad.detach().cpu() for k, p in net.named_parameters()}
    assert set(got) == set(grads)
    assert grads_close(got, grads, TOL) == []
    sd = net.state_dict()
    for k, v in after.items():
        assert rel_err(sd[k].cpu().double(), v.double()) < TOL, k
    with torch.no_grad():
        assert rel_err(net(x).cpu(), z['out2']) < TOL
        net.eval()
        assert rel_err(net(x).cpu(), z['out_eval']) < TOL


def test_discriminator_srgan_lists_32px():
    """the reference's own feature/stride lists (config.py:81-82) at 32x32, batch 4"""
    from oracle import init as oi
    z = np.load(os.path.join(GOLDEN, 'dis_32px_srgan.npz'))
    cfg = json.loads(str(z['cfg']))
    state = oi.synth_state(discriminator_shapes(cfg['input_shape'], cfg['list_n_features']), cfg['state_seed'])
    md = pkg('model_discriminator')
    net = md.Discriminator(tuple(cfg['input_shape']), cfg['list_n_features'], cfg['list_stride'])
    net.load_state_dict(state, strict=True)
    net = net.cuda().train()
    x = torch.from_numpy(z['x']).cuda().requires_grad_(True)
    out = net(x)
    assert rel_err(out.detach().cpu(), z['out']) < TOL
    (out * torch.from_numpy(z['r']).cuda()).sum().backward()
    assert rel_err(x.grad.cpu(), z['grad_x']) < TOL
    pg = {k: p.grad.detach().cpu() for k, p in net.named_parameters()}
    ref, got = {}, {}
    for k in z.files:
        if k.startswith('grad/'):
            ref[k[5:]], got[k[5:]] = torch.from_numpy(z[k]), pg[k[5:]]
        if k.startswith('gradsample/'):
            flat = pg[k[11:]].reshape(-1)
            ref[k[11:]] = torch.from_numpy(z[k])
            got[k[11:]] = flat[:: max(1, flat.numel() // 4096)][:4096]
    assert grads_close(got, ref, TOL) == []


def test_discriminator_batch_larger_than_the_fc_register_tile():
    """batch 20 > 16: the FC kernels run in batch slices whose weight gradients are summed; checked against the
    oracle on the same seeded input (1e-3)"""
    from helpers import oracle_fwd_bwd
    z, cfg, state, grads, after = load_case('dis_16px_w16')
    g = torch.Generator().manual_seed(5)
    shape = (20,) + tuple(z['x'].shape[1:])
    x = torch.rand(shape, generator=g) * 2 - 1
    r = torch.rand((20, 1), generator=g) * 2 - 1
    out_ref, gx_ref, grads_ref, _ = oracle_fwd_bwd(cfg, state, x, r)
    md = pkg('model_discriminator')
    net = md.Discriminator(tuple(cfg['input_shape']), cfg['list_n_features'], cfg['list_stride'])
    net.load_state_dict(state, strict=True)
    net = net.cuda().train()
    xd = x.cuda().requires_grad_(True)
    out = net(xd)
    assert rel_err(out.detach().cpu(), out_ref) < TOL
    (out * r.cuda()).sum().backward()
    assert rel_err(xd.grad.cpu(), gx_ref) < TOL
    got = {k: p.grad.detach().cpu() for k, p in net.named_parameters()}
    assert grads_close(got, grads_ref, TOL) == []
