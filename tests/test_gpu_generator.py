"""GPU: golden parity of the generator family drop-ins (rows a2-a4) against the vectors captured
from the reference modules, within BASELINE.json's 1e-3 relative fp32 tolerance."""
import pytest
import torch

from conftest import F32_TENSOR_BUILDS
from gpu_helpers import pkg
from helpers import GEN_CASES, PROG_CASES, grads_close, load_case, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-3


def build(cfg):
    if cfg['kind'] == 'generator':
        mg = pkg('model_generator')
        g = mg.Generator(cfg['n_blocks'], cfg['nf'], cfg['nl'], cfg['list_scales'], use_sn=cfg['use_sn'])
        for _ in range(cfg['n_suffix']):
            g = mg.GeneratorSuffix(g)
        return g
    mp = pkg('model_generator_progressive')
    g = mp.GeneratorProgresiveBase(cfg['n_blocks'], n_features=cfg['nf'])
    nf = cfg['nf']
    for i in range(cfg['n_suffix']):
        g = mp.GeneratorSuffix(g if i == 0 else g.beginning, n_features=nf)
        nf //= 4
    return g


@pytest.mark.parametrize('f32_build', F32_TENSOR_BUILDS, indirect=True)
@pytest.mark.parametrize('name', GEN_CASES + PROG_CASES)
def test_generator_matches_reference_golden(name, f32_build):
    z, cfg, state, grads, after = load_case(name)
    net = build(cfg)
    net.load_state_dict(state, strict=True)
    net = net.cuda().train()
    x = torch.from_numpy(z['x']).cuda().requires_grad_(True)
    r = torch.from_numpy(z['r']).cuda()
    out = net(x)
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
        assert rel_err(net(x).cpu(), z['out2']) < TOL          # advanced SN/BN state
        net.eval()
        assert rel_err(net(x).cpu(), z['out_eval']) < TOL      # running statistics, no power iteration


@pytest.mark.parametrize('f32_build', F32_TENSOR_BUILDS, indirect=True)
def test_full_depth_generator_matches_reference_golden(f32_build):
    """16 residual blocks (34 stacked conv+BatchNorm layers, spectral norm on every conv; the benchmark's own
    architecture, config.py:79-80) at B2, LR 16: out / grad_x / sampled parameter gradients / advanced SN+BN state /
    second training forward against vectors captured from the imported reference module, 1e-3 relative fp32"""
    import json, os
    import numpy as np
    from helpers import GOLDEN, load_sampled_case
    from test_oracle_golden import generator_shapes
    cfg0 = json.loads(str(np.load(os.path.join(GOLDEN, 'gen_x2_sn_16blocks.npz'))['cfg']))
    z, cfg, state, after, sample = load_sampled_case('gen_x2_sn_16blocks', generator_shapes(cfg0))
    net = build(cfg)
    net.load_state_dict(state, strict=True)
    net = net.cuda().train()
    x = torch.from_numpy(z['x']).cuda().requires_grad_(True)
    out = net(x)
    assert rel_err(out.detach().cpu(), z['out']) < TOL
    (out * torch.from_numpy(z['r']).cuda()).sum().backward()
    got, ref = sample({k: p.grad.detach().cpu() for k, p in net.named_parameters()})
    if f32_build == 'fp32':
        assert rel_err(x.grad.cpu(), z['grad_x']) < TOL
        assert grads_close(got, ref, TOL) == []
    else:
        # The split build's contractions carry 2^-17 operands: 3e-5 on the output after these 34 layers, against 5e-6 for the
        # exact-fp32 kernels (tools/diag_full_depth.py).  That is enough for ONE pre-activation within 1e-5 of zero to take the
        # other PReLU branch here, which moves the gradients in its receptive field: 1.49e-3 in max-norm on the input gradient
        # (243 of 1,536 elements by more than 1e-4), i.e. this vector is NOT met at 1e-3 by this build -- it is held to the
        # flip-sized bounds of test_gpu_full_size._flip_aware_compare instead (max-norm 5e-2, RMS 5e-3), and is the reason the
        # exact-fp32 build stays the parity build of record.
        gx, gx_ref = x.grad.cpu().double(), torch.from_numpy(z['grad_x']).double()
        assert float((gx - gx_ref).abs().max() / gx_ref.abs().max()) < 5e-2
        assert float((gx - gx_ref).pow(2).mean().sqrt() / gx_ref.pow(2).mean().sqrt()) < 5e-3
        assert grads_close(got, ref, 5e-2) == []
    sd = net.state_dict()
    for k, v in after.items():
        assert rel_err(sd[k].cpu().double(), v.double()) < TOL, k
    with torch.no_grad():
        assert rel_err(net(x).cpu(), z['out2']) < TOL


def test_no_silent_cpu_path():
    mg = pkg('model_generator')
    g = mg.Generator(1, 16, 64, [2])
    with pytest.raises(RuntimeError):
        g(torch.zeros(1, 3, 8, 8))          # CPU tensor: must refuse, never fall back


def _l2(a, b):
    a, b = a.double().reshape(-1), torch.as_tensor(b).double().reshape(-1)
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize('name', ['gen_x2_sn_w64', 'gen_x4_suffix_w32', 'prog_x8_w64'])
def test_generator_bf16_mode_tracks_reference(name):
    """bf16 matrix-core mode (the perf build; BASELINE.json config 1 names bf16): layers with Cin % 32 == 0
    round their MFMA operands to bf16 (8 significant bits) and accumulate in fp32.  Kernel-level error is
    bounded at 2e-2 (test_gpu_kernels.py); through the whole network the forward stays within 5e-2, while
    gradients additionally see PReLU/BatchNorm mask flips caused by the forward rounding, so they are held
    to an L2 bound and a direction (cosine) bound instead of the fp32 build's 1e-3."""
    E = pkg('engine')
    z, cfg, state, grads, after = load_case(name)
    E.set_precision('bf16')
    try:
        net = build(cfg)
        net.load_state_dict(state, strict=True)
        net = net.cuda().train()
        x = torch.from_numpy(z['x']).cuda().requires_grad_(True)
        out = net(x)
        assert rel_err(out.detach().cpu(), z['out']) < 5e-2
        (out * torch.from_numpy(z['r']).cuda()).sum().backward()
        assert _l2(x.grad.cpu(), z['grad_x']) < 0.25
        big = max(float(v.abs().max()) for v in grads.values())
        for k, p in net.named_parameters():
            ref = grads[k]
            if ref.numel() >= 64 and float(ref.abs().max()) > 0.05 * big:
                cos = float(torch.nn.functional.cosine_similarity(p.grad.cpu().reshape(1, -1).double(),
                                                                  ref.reshape(1, -1).double()))
                assert cos > 0.9, (k, cos)
    finally:
        E.set_precision('fp32')


def test_forward_no_end_matches_oracle():
    """Generator.forward_no_end / GeneratorSuffix.forward_no_end (model_generator.py:86-96,133-136): the
    NCHW activation in front of `end`, with gradients, against the CPU oracle."""
    from oracle import models as om, ops as oo
    z, cfg, state, grads, after = load_case('gen_x4_suffix_w32')
    net = build(cfg)
    net.load_state_dict(state, strict=True)
    net = net.cuda().train()
    x = torch.from_numpy(z['x']).cuda().requires_grad_(True)
    y = net.forward_no_end(x)
    st = {k: v.clone() for k, v in state.items()}
    for k in om.param_keys(st):
        st[k].requires_grad_(True)
    xr = torch.from_numpy(z['x']).clone().requires_grad_(True)
    c = om._Ctx(st, True)
    yr = om._gen_no_end(c, 'base.', xr, (2,))
    yr = oo.prelu(oo.pixel_shuffle(c.conv('upscale.0', yr), 2), st['upscale.2.weight'])
    assert tuple(y.shape) == tuple(yr.shape)
    assert rel_err(y.detach().cpu(), yr.detach()) < TOL
    r = torch.rand(yr.shape, generator=torch.Generator().manual_seed(3)) - 0.5
    (yr * r).sum().backward()
    (y * r.cuda()).sum().backward()
    assert rel_err(x.grad.cpu(), xr.grad) < TOL
    got = {k: p.grad.detach().cpu() for k, p in net.named_parameters() if p.grad is not None}
    ref = {k: st[k].grad for k in got}
    assert grads_close(got, ref, TOL) == []
    assert all(p.grad is None for k, p in net.named_parameters() if k.startswith('base.end'))


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_fused_skip_sums_reproduce_the_unfused_schedule(precision, monkeypatch):
    """Generator forward + backward with the residual blocks' skip sums formed inside the next conv's staging
    (SISR_PRO_RES_AFFINE, the default on trunk-eligible sizes) against the schedule with the separate elementwise pass
    (SISR_FUSE_SKIP=0): the staged values are the same expression, so output and every gradient must be bit-identical"""
    E, mg = pkg('engine'), pkg('model_generator')
    E.set_precision(precision)
    try:
        torch.manual_seed(0)
        net = mg.Generator(4, 64, 256, [2], use_sn=True).cuda().train()
        state = {k: v.clone() for k, v in net.state_dict().items()}
        g = torch.Generator().manual_seed(5)
        x = (torch.rand(2, 3, 16, 32, generator=g) * 2 - 1).cuda()
        r = (torch.rand(2, 3, 32, 64, generator=g) * 2 - 1).cuda()
        res = {}
        for sw in ('1', '0'):
            monkeypatch.setenv('SISR_FUSE_SKIP', sw)
            net.load_state_dict(state)
            net.zero_grad(set_to_none=True)
            xin = x.clone().requires_grad_(True)
            out = net(xin)
            (out * r).sum().backward()
            res[sw] = (out.detach().clone(), xin.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()})
        assert torch.equal(res['1'][0], res['0'][0]) and torch.equal(res['1'][1], res['0'][1])
        assert all(torch.equal(res['1'][2][k], res['0'][2][k]) for k in res['1'][2])
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_batchnorm_finalised_inside_the_consuming_conv(precision, monkeypatch):
    """Training-mode generator forward + backward with the BatchNorm constants finalised by the conv that applies them
    (SisrConvDesc.fin_*, the default on trunk-eligible sizes) against the stand-alone sisr_bn_finalize launches
    (SISR_FUSE_BNFIN=0): same statistics rows, two double-precision reductions of them -- outputs, gradients, saved
    constants and the updated running statistics agree to fp32 rounding (bf16 build: to a few stored roundings)"""
    E, mg = pkg('engine'), pkg('model_generator')
    E.set_precision(precision)
    try:
        torch.manual_seed(0)
        net = mg.Generator(4, 64, 256, [2], use_sn=True).cuda().train()
        state = {k: v.clone() for k, v in net.state_dict().items()}
        g = torch.Generator().manual_seed(5)
        x = (torch.rand(2, 3, 16, 32, generator=g) * 2 - 1).cuda()
        r = (torch.rand(2, 3, 32, 64, generator=g) * 2 - 1).cuda()
        res = {}
        for sw in ('1', '0'):
            monkeypatch.setenv('SISR_FUSE_BNFIN', sw)
            net.load_state_dict(state)
            net.zero_grad(set_to_none=True)
            xin = x.clone().requires_grad_(True)
            out = net(xin)
            (out * r).sum().backward()
            res[sw] = (out.detach().clone(), xin.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()},
                       {k: v.clone() for k, v in net.state_dict().items() if 'running' in k})
        tol = 1e-5 if precision == 'fp32' else 2e-2
        assert rel_err(res['1'][0], res['0'][0]) < tol and rel_err(res['1'][1], res['0'][1]) < 10 * tol
        for k in res['1'][2]:
            assert rel_err(res['1'][2][k], res['0'][2][k]) < 10 * tol, k
        for k in res['1'][3]:                          # running statistics: the same update from the same rows
            assert rel_err(res['1'][3][k], res['0'][3][k]) < 1e-6, k
            assert not torch.equal(res['1'][3][k], state[k]), k
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_deferred_slab_reductions_reproduce_the_separate_launches(precision, monkeypatch):
    """Generator backward with every weight gradient's slab sum carried by the next BatchNorm-backward finishing launch
    (engine.PendingSlabs, the default) against the schedule with one sisr_slab_reduce_f32 launch per layer
    (SISR_FUSE_SLABRED=0): the same sums in the same order -- every gradient bit-identical"""
    E, mg = pkg('engine'), pkg('model_generator')
    E.set_precision(precision)
    try:
        torch.manual_seed(0)
        net = mg.Generator(3, 64, 256, [2], use_sn=True).cuda().train()
        state = {k: v.clone() for k, v in net.state_dict().items()}
        g = torch.Generator().manual_seed(6)
        x = (torch.rand(2, 3, 16, 32, generator=g) * 2 - 1).cuda()
        r = (torch.rand(2, 3, 32, 64, generator=g) * 2 - 1).cuda()
        res = {}
        for sw in ('1', '0'):
            monkeypatch.setenv('SISR_FUSE_SLABRED', sw)
            net.load_state_dict(state)
            net.zero_grad(set_to_none=True)
            xin = x.clone().requires_grad_(True)
            (net(xin) * r).sum().backward()
            res[sw] = (xin.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()})
        assert torch.equal(res['1'][0], res['0'][0])
        assert all(torch.equal(res['1'][1][k], res['0'][1][k]) for k in res['1'][1])
    finally:
        E.set_precision('fp32')


def test_progressive_trunk_and_beginning_are_callable_like_the_references():
    """model_generator_progressive.py:40-44 and :52-56: GeneratorProgresiveBase.forward (the bare trunk: first conv + PReLU,
    blocks, conv + BatchNorm; no long skip, no upscale stage) and the ``beginning`` Sequential a suffix is stacked on
    (prefix, conv, PixelShuffle(2), PReLU) are nn.Modules one can call: outputs and every gradient against the oracle's
    layers"""
    from oracle import models as om, ops as oo
    mp = pkg('model_generator_progressive')
    torch.manual_seed(0)
    g0 = mp.GeneratorProgresiveBase(2, n_features=16)
    g1 = mp.GeneratorSuffix(g0, n_features=16)
    state = {k: v.detach().clone() for k, v in g1.state_dict().items()}
    g1 = g1.cuda().train()
    x0 = torch.rand(2, 3, 16, 16, generator=torch.Generator().manual_seed(4)) * 2 - 1

    def oracle(upto):
        st = {k: v.clone() for k, v in state.items()}
        for k in om.param_keys(st):
            st[k].requires_grad_(True)
        xr = x0.clone().requires_grad_(True)
        c = om._Ctx(st, True)
        base = 'beginning.0.'
        t = c.prelu(base + 'first_layers.1', c.conv(base + 'first_layers.0', xr, padding=4))
        for i in range(2):
            b = base + 'block_list.%d.layers.' % i
            r_ = t
            t = c.bn(b + '4', c.conv(b + '3', c.prelu(b + '2', c.bn(b + '1', c.conv(b + '0', t)))))
            t = r_ + t
        t = c.bn(base + 'block_list_end.1', c.conv(base + 'block_list_end.0', t))
        if upto == 'beginning':
            t = c.prelu('beginning.3', oo.pixel_shuffle(c.conv('beginning.1', t), 2))
        return st, xr, t

    for upto, module, shape in (('trunk', g0, (2, 16, 16, 16)), ('beginning', g1.beginning, (2, 4, 32, 32))):
        g1.load_state_dict(state)
        g1.zero_grad(set_to_none=True)
        x = x0.cuda().requires_grad_(True)
        y = module(x)
        assert tuple(y.shape) == shape
        st, xr, yr = oracle(upto)
        assert rel_err(y.detach().cpu(), yr.detach()) < TOL
        r = torch.rand(yr.shape, generator=torch.Generator().manual_seed(5)) - 0.5
        (yr * r).sum().backward()
        (y * r.cuda()).sum().backward()
        assert rel_err(x.grad.cpu(), xr.grad) < TOL
        got = {k: p.grad.detach().cpu() for k, p in g1.named_parameters() if p.grad is not None}
        ref = {k: st[k].grad for k in got}
        assert grads_close(got, ref, TOL) == [], upto
        assert all((k.startswith('end.') or (upto == 'trunk' and k.startswith(('beginning.1', 'beginning.3')))) == (p.grad is None)
                   for k, p in g1.named_parameters())
