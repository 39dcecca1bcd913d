"""GPU: the bf16 matrix-core mode (perf build) on the discriminator, MaskedVGG and a whole training
iteration.  Layers with Cin % 32 == 0 round their MFMA operands to bf16 (fp32 accumulate); stride-2
forward convs and the weight gradients use the bf16 kernels, stride-2 data gradients and 3-channel
edge layers stay on the fp32 kernels.  Tolerances are bf16-sized (stated per assert); the 1e-3 parity
bar belongs to the default fp32 build (tests/test_gpu_*.py)."""
import numpy as np
import pytest
import torch

from gpu_helpers import pkg
from helpers import load_case, rel_err

pytestmark = pytest.mark.gpu


def _cos(a, b):
    return float(torch.nn.functional.cosine_similarity(a.reshape(1, -1).double(), b.reshape(1, -1).double()))


@pytest.fixture()
def bf16_engine():
    E = pkg('engine')
    E.set_precision('bf16')
    yield E
    E.set_precision('fp32')


def test_discriminator_bf16_mode(bf16_engine):
    z, cfg, state, grads, after = load_case('dis_16px_w16')
    md = pkg('model_discriminator')
    net = md.Discriminator(tuple(cfg['input_shape']), cfg['list_n_features'], cfg['list_stride'])
    net.load_state_dict(state, strict=True)
    net = net.cuda().train()
    x = torch.from_numpy(z['x']).cuda().requires_grad_(True)
    out = net(x)
    assert rel_err(out.detach().cpu(), z['out']) < 3e-2
    (out * torch.from_numpy(z['r']).cuda()).sum().backward()
    assert _cos(x.grad.cpu(), torch.from_numpy(z['grad_x'])) > 0.98
    big = max(float(v.abs().max()) for v in grads.values())
    for k, p in net.named_parameters():
        if grads[k].numel() >= 64 and float(grads[k].abs().max()) > 0.05 * big:
            assert _cos(p.grad.cpu(), grads[k]) > 0.95, k


def test_masked_vgg_bf16_mode(bf16_engine, golden_dir):
    z = np.load(golden_dir + '/vgg_standin.npz')
    mce = pkg('model_content_extractor')
    div = int(z['width_div'])
    state = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('state/')}
    for mask in (0b00010, 0b01111, 0b10000):
        net = mce.MaskedVGG(mask, width_div=div, pretrained=False)
        net.load_state_dict({k: v for k, v in state.items() if k in net.state_dict()}, strict=True)
        net = net.cuda()
        x = torch.from_numpy(z['x']).cuda().requires_grad_(True)
        f = net(x)
        assert rel_err(f.detach().cpu(), z['out_%d' % mask]) < 3e-2, mask
        (f * torch.from_numpy(z['r_%d' % mask]).cuda()).sum().backward()
        # 16 bf16 layers deep with ReLU / max-pool masks that flip on rounding: direction, not digits
        assert _cos(x.grad.cpu(), torch.from_numpy(z['grad_x_%d' % mask])) > (0.93 if mask >= 16 else 0.97), mask


def test_training_iteration_runs_in_bf16_mode(bf16_engine):
    """D step + G step with the full-width modules at HR 32 (shapes where every bf16 kernel family is
    exercised: trunk, upscale with PixelShuffle, stride-2 forward, FC, VGG22) -- finite losses and grads."""
    mg, md, mce, ut = pkg('model_generator'), pkg('model_discriminator'), pkg('model_content_extractor'), pkg('utils')
    torch.manual_seed(0)
    dev = torch.device('cuda')
    net_g = mg.Generator(2, 64, 256, [2], use_sn=True).to(dev)
    net_d = md.Discriminator((3, 32, 32), [64, 64, 128, 128], [1, 2, 1, 2]).to(dev)
    ext = mce.MaskedVGG(0b00010, pretrained=False).to(dev)
    crit = torch.nn.BCELoss()
    hr = (torch.rand(8, 3, 32, 32, generator=torch.Generator().manual_seed(3)) * 2 - 1).to(dev)
    fake = net_g(ut.lr_from_hr(hr, (16, 16), device=dev))
    net_d.zero_grad()
    err_d = crit(net_d(hr).view(-1), torch.full((8,), .9, device=dev)) + \
        crit(net_d(fake.detach()).view(-1), torch.zeros(8, device=dev))
    err_d.backward()
    net_g.zero_grad()
    err_g = crit(net_d(fake).view(-1), torch.ones(8, device=dev)) * 5e-2 + torch.mean(torch.pow(ext(hr) - ext(fake), 2))
    err_g.backward()
    assert torch.isfinite(err_d) and torch.isfinite(err_g)
    for net in (net_g, net_d):
        for k, p in net.named_parameters():
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k


def test_bf16_mode_trains_like_the_fp32_build():
    """30 Adam steps of the benchmark objective (10 * MSE to the HR patch, lr 1e-3) on a fixed batch, once per
    precision build from the same initial state: the loss must fall in both and the bf16 trajectory must stay
    within 1 % of the fp32 one at every step (measured: 0.04 %) (bf16 rounds MFMA operands only; accumulation, BatchNorm statistics,
    parameters and the optimizer are fp32)."""
    E, mg, ut, op = pkg('engine'), pkg('model_generator'), pkg('utils'), pkg('optim')
    torch.manual_seed(0)
    ref = mg.Generator(4, 64, 256, [2], use_sn=True)
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    hr = (torch.rand(8, 3, 64, 64, generator=torch.Generator().manual_seed(5)) * 2 - 1).cuda()
    traj = {}
    try:
        for prec in ('fp32', 'bf16'):
            E.set_precision(prec)
            net = mg.Generator(4, 64, 256, [2], use_sn=True)
            net.load_state_dict(state)
            net = net.cuda().train()
            opt = op.Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.999))
            losses = []
            for _ in range(30):
                lr = ut.lr_from_hr(hr, (32, 32), device=hr.device)
                loss = 10.0 * torch.mean(torch.pow(hr - net(lr), 2))
                net.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
                losses.append(float(loss))
            traj[prec] = losses
    finally:
        E.set_precision('fp32')
    a, b = traj['fp32'], traj['bf16']
    assert a[-1] < 0.7 * a[0] and b[-1] < 0.7 * b[0], (a[0], a[-1], b[0], b[-1])
    assert max(abs(x - y) / x for x, y in zip(a, b)) < 0.01, [round(abs(x - y) / x, 4) for x, y in zip(a, b)]


@pytest.mark.parametrize('net_kind', ['discriminator', 'generator_lr48', 'generator_lr96'])
def test_batched_and_per_layer_weight_gradients_agree(bf16_engine, net_kind, monkeypatch):
    """the backward schedules collect the weight gradients of a pass and launch them together (engine.WgradDeepBatch: wgrad_deep.hip's
    flat-grid batch for D and for the LR 48 trunk, the persistent kernel's table launch for the LR 96 trunk); SISR_WGRAD_BATCH=0 launches
    every layer on its own.  Same forward, same operands, same per-tile arithmetic: every parameter gradient agrees to the rounding of
    the bf16 slabs (the partition into slabs is what differs)."""
    E = bf16_engine
    mg, md = pkg('model_generator'), pkg('model_discriminator')
    torch.manual_seed(0)
    gx = torch.Generator().manual_seed(4)
    if net_kind == 'discriminator':
        net = md.Discriminator((3, 96, 96), [64, 64, 128, 128, 256, 256, 512, 512], [1, 2, 1, 2, 1, 2, 1, 2]).cuda().train()
        x = (torch.rand(16, 3, 96, 96, generator=gx) * 2 - 1).cuda()
        key = 'wgrad_deep_batch'
    else:
        lr = 48 if net_kind.endswith('48') else 96
        net = mg.Generator(4, 64, 256, [2], use_sn=True).cuda().train()
        x = (torch.rand(16, 3, lr, lr, generator=gx) * 2 - 1).cuda()
        key = 'wgrad_deep_batch' if lr == 48 else 'wgrad_trunk_batch'
    state = {k: v.detach().clone() for k, v in net.state_dict().items()}
    grads = {}
    for sw in ('1', '0'):
        monkeypatch.setenv('SISR_WGRAD_BATCH', sw)
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        before = E.KERNEL_COUNTS.get(key, 0)
        out = net(x)
        r = torch.rand(out.shape, generator=torch.Generator().manual_seed(5)).cuda() - 0.5
        (out * r).sum().backward()
        assert (E.KERNEL_COUNTS.get(key, 0) > before) == (sw == '1'), (key, sw)
        grads[sw] = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    for k in grads['1']:
        a, b = grads['1'][k], grads['0'][k]
        assert float((a - b).abs().max()) <= 6e-3 * max(float(b.abs().max()), 1e-30), k
