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


def test_vgg_4conv_1maxpool_matches_torch_primitives():
    """model_content_extractor.vgg_4conv_1maxPool (model_content_extractor.py:16-31): ``vgg19.features[:9]`` -- conv, ReLU,
    conv, ReLU, MaxPool, conv, ReLU, conv, ReLU -- frozen, output (B, 128, H/2, W/2) post-ReLU; forward and input gradient
    against the same stack of torch primitives with the same (seeded) weights; state_dict keys as torchvision's Sequential"""
    import torch.nn.functional as F
    mce = pkg('model_content_extractor')
    net = mce.vgg_4conv_1maxPool(pretrained=False)
    assert list(net.state_dict()) == ['0.weight', '0.bias', '2.weight', '2.bias', '5.weight', '5.bias', '7.weight', '7.bias']
    assert all(not p.requires_grad for p in net.parameters()) and not net.training
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    x0 = torch.rand(2, 3, 32, 48, generator=torch.Generator().manual_seed(8)) * 2 - 1
    x = x0.cuda().requires_grad_(True)
    y = net(x)
    assert tuple(y.shape) == (2, 128, 16, 24)
    xr = x0.clone().requires_grad_(True)
    t = F.relu(F.conv2d(xr, sd['0.weight'], sd['0.bias'], padding=1))
    t = F.relu(F.conv2d(t, sd['2.weight'], sd['2.bias'], padding=1))
    t = F.max_pool2d(t, 2, 2)
    t = F.relu(F.conv2d(t, sd['5.weight'], sd['5.bias'], padding=1))
    yr = F.relu(F.conv2d(t, sd['7.weight'], sd['7.bias'], padding=1))
    assert rel_err(y.detach().cpu(), yr.detach()) < TOL
    r = torch.rand(yr.shape, generator=torch.Generator().manual_seed(9)) - 0.5
    (yr * r).sum().backward()
    (y * r.cuda()).sum().backward()
    assert rel_err(x.grad.cpu(), xr.grad) < TOL
    with pytest.raises(RuntimeError):
        mce.vgg_4conv_1maxPool()                                  # no torchvision / checkpoint here: refuses, as the reference fails
