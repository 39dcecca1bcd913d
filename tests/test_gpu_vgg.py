"""GPU: MaskedVGG drop-in (row a6) against the torch-primitive VGG19-features stand-in fixtures
(synthetic weights, widths / 8): tap semantics (in-place-ReLU aliasing), NCHW flatten order, pooling
and the data gradient, plus the reference's get_size known-answers for all 31 masks."""
import numpy as np
import pytest
import torch

from gpu_helpers import pkg
from helpers import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-3


def test_masked_vgg_matches_standin_and_get_size(golden_dir):
    z = np.load(golden_dir + '/vgg_standin.npz')
    mce = pkg('model_content_extractor')
    div = int(z['width_div'])
    state = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('state/')}
    x0 = torch.from_numpy(z['x']).cuda()
    for mask in range(1, 32):
        net = mce.MaskedVGG(mask, width_div=div, pretrained=False)
        sd = {k: v for k, v in state.items() if k in net.state_dict()}
        net.load_state_dict(sd, strict=True)
        net = net.cuda()
        key = 'out_%d' % mask
        x = x0.clone().requires_grad_(key in z.files)
        f = net(x)
        assert tuple(f.shape) == (2, mce.get_size(x0, mask) // div)          # model_content_extractor.py:101
        if key in z.files:
            assert rel_err(f.detach().cpu(), z[key]) < TOL, mask
            (f * torch.from_numpy(z['r_%d' % mask]).cuda()).sum().backward()
            assert rel_err(x.grad.cpu(), z['grad_x_%d' % mask]) < TOL, mask
    assert all(not p.requires_grad for p in net.parameters())


def test_masked_vgg_full_width_vs_oracle():
    """full-width VGG22 / VGG54 / default masks on a 32x32 batch against the CPU oracle (seeded weights)"""
    from oracle import models as om
    mce = pkg('model_content_extractor')
    x0 = (torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(5)) * 2 - 1)
    for mask in (0b00010, 0b10000, 0b01111):
        net = mce.MaskedVGG(mask, pretrained=False)
        state = {k: v.detach().clone() for k, v in net.state_dict().items()}
        net = net.cuda()
        x = x0.cuda().requires_grad_(True)
        f = net(x)
        xr = x0.clone().requires_grad_(True)
        fr = om.masked_vgg_forward(state, xr, mask)
        assert rel_err(f.detach().cpu(), fr.detach()) < TOL
        r = torch.rand(fr.shape, generator=torch.Generator().manual_seed(6)) - 0.5
        (fr * r).sum().backward()
        (f * r.cuda()).sum().backward()
        assert rel_err(x.grad.cpu(), xr.grad) < TOL
