"""GPU: the benchmark configuration at FULL size (BASELINE.json north_star: Generator(16, 64, 256, [2], use_sn=True),
B=16, LR 96x96 -> SR 192x192; cfg2's LR 48; the discriminator at B16 HR 96).

Two kinds of checks:
  * VALUES against the CPU oracle (the fp32 parity build, 1e-3 relative as north_star states): at these sizes every
    persistent kernel walks 5-9 tiles per workgroup (1,152 tiles over 231-256 workgroups), which the small golden
    cases never do -- output, input gradient, every parameter gradient (per tensor), the advanced spectral-norm
    vectors and BatchNorm running statistics.  One oracle step is ~10 s of CPU at LR 96.
  * size-independent PROPERTIES of both builds (the bf16 build has no 1e-3 oracle bound):
      - determinism  -- every reduction on the path has a fixed order, so two runs are BIT-identical;
      - linearity of the backward pass in the incoming gradient -- scaling grad_output by 2 (exact in fp32) must
        scale every parameter gradient and the input gradient by exactly 2;
      - batch-permutation equivariance -- training-mode BatchNorm statistics do not depend on the order of the
        patches, so G(x[perm]) = G(x)[perm] and the parameter gradients agree (up to fp32 re-association of the
        per-tile statistics, 2e-5 relative);
      - per-sample finiteness and the tanh range of the output image."""
import pytest
import torch

from conftest import F32_TENSOR_BUILDS
from gpu_helpers import pkg
from helpers import grads_close, oracle_fwd_bwd, rel_err

pytestmark = pytest.mark.gpu
B, LR = 16, 96


def _run(net, mg, x, r, spectral_state):
    """one training-mode forward+backward from a fixed spectral-norm / BatchNorm state"""
    net.load_state_dict(spectral_state)
    net.zero_grad(set_to_none=True)
    xx = x.clone().requires_grad_(True)
    out = net(xx)
    (out * r).sum().backward()
    return out.detach(), xx.grad.detach(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_full_size_generator_properties(precision):
    E, mg = pkg('engine'), pkg('model_generator')
    E.set_precision(precision)
    try:
        torch.manual_seed(0)
        net = mg.Generator(16, 64, 256, [2], use_sn=True).cuda().train()
        state = {k: v.clone() for k, v in net.state_dict().items()}
        g = torch.Generator().manual_seed(11)
        x = (torch.rand(B, 3, LR, LR, generator=g) * 2 - 1).cuda()
        r = (torch.rand(B, 3, 2 * LR, 2 * LR, generator=g) * 2 - 1).cuda()
        out, gx, grads = _run(net, mg, x, r, state)
        assert tuple(out.shape) == (B, 3, 2 * LR, 2 * LR)
        assert bool(torch.isfinite(out).all()) and float(out.abs().max()) <= 1.0           # tanh range
        assert all(bool(torch.isfinite(v).all()) for v in grads.values()) and bool(torch.isfinite(gx).all())
        # determinism: bit-identical replay
        out2, gx2, grads2 = _run(net, mg, x, r, state)
        assert torch.equal(out, out2) and torch.equal(gx, gx2)
        assert all(torch.equal(grads[k], grads2[k]) for k in grads)
        # linearity of the backward pass: 2 * grad_output -> exactly 2 * every gradient
        _, gx3, grads3 = _run(net, mg, x, 2.0 * r, state)
        assert torch.equal(gx3, 2.0 * gx)
        assert all(torch.equal(grads3[k], 2.0 * grads[k]) for k in grads), \
            [k for k in grads if not torch.equal(grads3[k], 2.0 * grads[k])][:5]
        # batch-permutation equivariance (training-mode BatchNorm statistics are order-free)
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).cuda()
        outp, gxp, gradsp = _run(net, mg, x[perm], r[perm], state)
        # fp32: max-norm 2e-5.  bf16: the permutation re-associates the statistics sums, which flips bf16 roundings of
        # stored activations (2^-8 each) that 33 layers then carry along -- a tail statistic over 1.8 M outputs, so the
        # max-norm bound is loose (6e-2) and the RMS error carries the test (2e-2)
        tol = 2e-5 if precision == 'fp32' else 2e-2
        if precision == 'fp32':
            assert rel_err(outp.cpu(), out[perm].cpu()) < tol
        else:
            d, ref_o = (outp - out[perm]).double(), out[perm].double()
            assert float(d.pow(2).mean().sqrt() / ref_o.pow(2).mean().sqrt()) < 2e-2       # (measured 0.9e-2 .. 1.0e-2 across builds)
            assert rel_err(outp.cpu(), out[perm].cpu()) < 3 * tol
        # input gradient: a re-associated statistics sum moves a BatchNorm output by ~1e-7 relative, which flips the PReLU
        # mask of the few pre-activations that close to zero (expected: a handful per layer at 9.4 M activations) -- each
        # flip changes the gradient in its receptive field by O(1e-2) of the max (the persistent kernels accumulate the
        # statistics of a workgroup's tiles in fp32 running sums, so the association does change with the batch order;
        # measured: fp32 RMS 9e-4, max 7e-3).  So: RMS error bounded at 2e-3, max-norm loose.
        dg, ref_g = (gxp - gx[perm]).double(), gx[perm].double()
        assert float(dg.pow(2).mean().sqrt() / ref_g.pow(2).mean().sqrt()) < (2e-3 if precision == 'fp32' else 0.15)
        assert rel_err(gxp.cpu(), gx[perm].cpu()) < (5e-2 if precision == 'fp32' else 50 * tol)
        big = max(float(v.abs().max()) for v in grads.values())
        for k in grads:
            scale = max(float(grads[k].abs().max()), 0.02 * big)
            assert float((gradsp[k] - grads[k]).abs().max()) / scale < (2e-2 if precision == 'fp32' else 50 * tol), k   # (worst: a PReLU slope, a cancelling scalar sum; measured 5.4e-3)
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_full_size_discriminator_and_vgg_properties(precision):
    """SURVEY 8d cfg2 sizes: Discriminator on 16 x 3 x 96 x 96 with the reference's feature / stride lists
    (config.py:81-82; fc_in = 18,432) and MaskedVGG(0b00010): determinism and backward linearity"""
    E, md, mce = pkg('engine'), pkg('model_discriminator'), pkg('model_content_extractor')
    E.set_precision(precision)
    try:
        torch.manual_seed(0)
        net = md.Discriminator((3, 96, 96), [64, 64, 128, 128, 256, 256, 512, 512], [1, 2, 1, 2, 1, 2, 1, 2]).cuda().train()
        ext = mce.MaskedVGG(0b00010, pretrained=False).cuda()
        state = {k: v.clone() for k, v in net.state_dict().items()}
        g = torch.Generator().manual_seed(12)
        x = (torch.rand(16, 3, 96, 96, generator=g) * 2 - 1).cuda()
        r = (torch.rand(16, 1, generator=g) * 2 - 1).cuda()

        def run(scale):
            net.load_state_dict(state)
            net.zero_grad(set_to_none=True)
            xx = x.clone().requires_grad_(True)
            out = net(xx)
            (out * (scale * r)).sum().backward()
            return out.detach(), xx.grad.detach(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}

        out, gx, grads = run(1.0)
        assert tuple(out.shape) == (16, 1) and bool(((out > 0) & (out < 1)).all())           # sigmoid range
        out2, gx2, grads2 = run(1.0)
        assert torch.equal(out, out2) and torch.equal(gx, gx2) and all(torch.equal(grads[k], grads2[k]) for k in grads)
        _, gx3, grads3 = run(2.0)
        assert torch.equal(gx3, 2.0 * gx) and all(torch.equal(grads3[k], 2.0 * grads[k]) for k in grads)
        # VGG22 features: deterministic, finite, and the input gradient is linear in the incoming gradient
        xv = x.clone().requires_grad_(True)
        f = ext(xv)
        rv = torch.rand(f.shape, generator=torch.Generator().manual_seed(4)).cuda()
        (f * rv).sum().backward()
        g1 = xv.grad.clone()
        xv2 = x.clone().requires_grad_(True)
        f2 = ext(xv2)
        (f2 * (2.0 * rv)).sum().backward()
        assert torch.equal(f, f2) and bool(torch.isfinite(f).all()) and torch.equal(xv2.grad, 2.0 * g1)
    finally:
        E.set_precision('fp32')


TOL = 1e-3      # BASELINE.json north_star: "within 1e-3 relative fp32"


def _gpu_fwd_bwd(net, state, x, r):
    net.load_state_dict(state)
    net.zero_grad(set_to_none=True)
    xx = x.cuda().requires_grad_(True)
    out = net(xx)
    (out * r.cuda()).sum().backward()
    grads = {k: p.grad.detach().cpu() for k, p in net.named_parameters()}
    return out.detach().cpu(), xx.grad.cpu(), grads, {k: v.detach().cpu() for k, v in net.state_dict().items()}


# The split build ('bf16x3': 2^-17 operands in the trunk contractions) is ~6x noisier than fp32 arithmetic in the FORWARD pass
# (3e-5 against 5e-6 on the output of the 34 layers -- both far inside 1e-3), and every bit of forward noise flips proportionally
# more activation masks: its GRADIENTS are held to wider, measured bounds (tools/parity_diag.py with SISR_PRECISION=bf16x3, B16 /
# LR 96: nearly linear activations <= 2e-3 on the worst tensor; the reference's activations: input gradient 2.3e-2 max / 4e-3 RMS
# against fp64 where the fp32 oracle itself is 7e-3 / 8e-4).  That is why it is NOT the parity build of record.
GRAD_TOL = {'fp32': TOL, 'bf16x3': 3e-3}
FLIP_BOUNDS = {'fp32': (5e-2, 5e-3, 1.5e-1, 3.0), 'bf16x3': (1e-1, 2e-2, 6e-1, 10.0)}   # max-norm, RMS, PReLU slope, x the oracle's mean RMS


def _oracle_compare(net, cfg, state, x, r, build='fp32'):
    """one training-mode forward+backward of `net` (GPU) against the CPU oracle on the same state: output, advanced buffers at
    1e-3; input gradient and EVERY parameter gradient tensor by tensor at GRAD_TOL[build] (1e-3 for the fp32 parity build)"""
    out, gx, got, sd = _gpu_fwd_bwd(net, state, x, r)
    o_out, o_gx, o_grads, o_new = oracle_fwd_bwd(cfg, state, x, r)
    assert rel_err(out, o_out) < TOL, 'output'
    assert rel_err(gx, o_gx) < GRAD_TOL[build], 'input gradient'
    assert set(got) == set(o_grads)
    assert grads_close(got, o_grads, GRAD_TOL[build]) == []
    for k, v in o_new.items():                                 # advanced u / v, running statistics, batch counters
        assert rel_err(sd[k].double(), v.double()) < TOL, k
    return out


def _generator_case(lr, init, slopes=None):
    from oracle import init as oinit
    mg = pkg('model_generator')                                 # (the build under test is set by the f32_build fixture)
    torch.manual_seed(0)
    net = mg.Generator(16, 64, 256, [2], use_sn=True).cuda().train()
    if init == 'default':
        state = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    else:
        state = oinit.synth_state({k: tuple(v.shape) for k, v in net.state_dict().items()}, 5)
    if slopes is not None:                                      # PReLU weights: the 1-element `weight` tensors
        lo, hi = slopes
        gs = torch.Generator().manual_seed(3)
        for k in state:
            if k.endswith('.weight') and state[k].numel() == 1:
                state[k] = lo + (hi - lo) * torch.rand(state[k].shape, generator=gs)
    g = torch.Generator().manual_seed(21)
    x = torch.rand(B, 3, lr, lr, generator=g) * 2 - 1
    r = torch.rand(B, 3, 2 * lr, 2 * lr, generator=g) * 2 - 1
    return net, {'kind': 'generator', 'list_scales': [2], 'n_suffix': 0}, state, x, r


def _builds_with_slow_split_at_lr96():
    """(lr, init, build) cases of the full-size generator tests: the frozen split build's LR 96 cases (71 s + 27 s of CPU oracle each) are
    marked gpu_slow -- the default run keeps the fp32 parity build at both sizes and the split build at LR 48"""
    out = []
    for lr, init in ((48, 'default'), (96, 'synthetic')):
        for b in F32_TENSOR_BUILDS:
            marks = [pytest.mark.gpu_slow] if (lr == 96 and b != 'fp32') else []
            out.append(pytest.param(lr, init, b, marks=marks, id='%d-%s-%s' % (lr, init, b)))
    return out


@pytest.mark.parametrize('lr,init,f32_build', _builds_with_slow_split_at_lr96(), indirect=['f32_build'])
def test_full_size_generator_values_at_1e3_with_nearly_linear_activations(lr, init, f32_build):
    """model_generator.py:86-101 at config.py:79-80's sizes -- Generator(16, 64, 256, [2], use_sn=True), B16, LR 48
    (cfg2's generator: 288 tiles) and LR 96 (the headline workload: 1,152 tiles = 5 / 9 per workgroup, the schedule
    no small case reaches) -- against the CPU oracle, EVERY tensor at 1e-3 relative: output, input gradient, all 140
    parameter gradients, advanced spectral-norm vectors and running statistics.

    The PReLU slopes are set to 0.990-0.999 for this test.  Why: with the reference's 0.25 a pre-activation that lands
    within fp32 rounding of zero flips its mask between ANY two fp32 implementations, and each flip moves the
    gradients in its receptive field by (1 - slope) x O(1e-2) of their maximum -- at 77-300 M activations a few dozen
    always do, on the CPU oracle as much as here (measured with tools/parity_diag.py: the oracle in fp32 against
    itself in fp64 is 5e-3 .. 1e-2 off in max-norm on the input gradient, on this box and on the GPU box).  With
    slopes near 1 the same kernels run the same schedule (general-slope code path, slope read from the device tensor)
    on a function that is smooth to within (1 - slope), so the stated tolerance is a meaningful per-tensor bound."""
    net, cfg, state, x, r = _generator_case(lr, init, slopes=(0.990, 0.999))
    out = _oracle_compare(net, cfg, state, x, r, build=f32_build)
    assert tuple(out.shape) == (B, 3, 2 * lr, 2 * lr)


def _errs(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    d = a - b
    return float(d.abs().max() / b.abs().max()), float(d.pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


def _flip_aware_compare(net, cfg, state, x, r, strict_keys=(), build='fp32'):
    """Hard activations (the reference's own PReLU 0.25 / LeakyReLU 0.01) at full size.
    * FORWARD quantities -- output, advanced spectral-norm vectors, running statistics -- at 1e-3 against the fp32
      oracle (in fact ~1e-6).
    * GRADIENTS against the oracle evaluated in FP64 (the exact answer), with the oracle in fp32 -- the reference's own
      arithmetic -- measured beside it.  A pre-activation within fp32 rounding of zero takes the other branch in any
      second fp32 implementation; each such flip moves the gradients in its receptive field by up to O(1e-2) of their
      maximum, and every per-channel / scalar reduction (BatchNorm affine gradients, PReLU slopes) by its share.  The
      reference's fp32 CPU path is itself 3e-3 .. 1.3e-2 (max-norm) / 3e-4 .. 2e-3 (RMS) from the exact input gradient
      at these sizes, differently on different hosts (tools/parity_diag.py).  So, per gradient tensor:
        - max-norm error < 5e-2 and RMS error < 5e-3 -- flip-sized, not kernel-bug-sized (one wrong 8x16 tile of 1,152
          is O(1) on 1e-3 of the elements);
        - tensors in `strict_keys` (no hard activation between them and the loss) at the plain 1e-3;
      and over the model: the mean RMS error of the HIP path is at most 3x that of the fp32 oracle, or below 1e-3."""
    from helpers import analytically_zero
    out, gx, got, sd = _gpu_fwd_bwd(net, state, x, r)
    o_out, o_gx, o_grads, o_new = oracle_fwd_bwd(cfg, state, x, r)
    assert rel_err(out, o_out) < TOL, 'output'
    for k, v in o_new.items():
        assert rel_err(sd[k].double(), v.double()) < TOL, k
    st64 = {k: (v.double() if v.is_floating_point() else v) for k, v in state.items()}
    _, x_gx, x_grads, _ = oracle_fwd_bwd(cfg, st64, x.double(), r.double())
    got['grad_x'], o_grads['grad_x'], x_grads['grad_x'] = gx, o_gx, x_gx
    g_rms, c_rms, bad = [], [], []
    for k in x_grads:
        if analytically_zero(k, x_grads):
            continue
        gm, gr = _errs(got[k], x_grads[k])
        _, cr = _errs(o_grads[k], x_grads[k])
        g_rms.append(gr)
        c_rms.append(cr)
        b_max, b_rms, b_slope, b_mean = FLIP_BOUNDS[build]
        limit_max, limit_rms = (GRAD_TOL[build], GRAD_TOL[build]) if any(k.startswith(p) for p in strict_keys) else (b_max, b_rms)
        if k.endswith('.weight') and x_grads[k].numel() == 1:
            limit_max = limit_rms = b_slope                        # PReLU slope: ONE cancelling sum over 9-38 M products
        if not (gm < limit_max and gr < limit_rms):
            bad.append((k, gm, gr))
    assert bad == []
    mg_, mc_ = sum(g_rms) / len(g_rms), sum(c_rms) / len(c_rms)
    assert mg_ <= max(FLIP_BOUNDS[build][3] * mc_, TOL), (mg_, mc_)
    return out


@pytest.mark.parametrize('lr,init,f32_build', _builds_with_slow_split_at_lr96(), indirect=['f32_build'])
def test_full_size_generator_with_the_references_activations(lr, init, f32_build):
    """the same sizes with the reference's own activations (PReLU 0.25 from the default init / 0.1-0.4 synthetic):
    see _flip_aware_compare.  The last conv and the upscale conv sit behind one PReLU only; everything is held to the
    flip-aware bounds, the forward to 1e-3."""
    net, cfg, state, x, r = _generator_case(lr, init)
    _flip_aware_compare(net, cfg, state, x, r, strict_keys=('end.',), build=f32_build)


def test_full_size_discriminator_matches_the_oracle():
    """model_discriminator.py:55-62 at cfg2's size: B16, HR 96, the reference's feature / stride lists
    (config.py:81-82; fc_in = 18,432, 23.6 M parameters), synthetic state.  LeakyReLU(0.01) is part of the reference's
    module (not a parameter), so the gradients of the three 96^2 / 48^2 layers see mask flips (_flip_aware_compare);
    the five deeper conv layers and both Linear layers are held to the plain 1e-3 (measured: 5e-6)."""
    from oracle import init as oinit
    E, md = pkg('engine'), pkg('model_discriminator')
    E.set_precision('fp32')
    feats, strides = [64, 64, 128, 128, 256, 256, 512, 512], [1, 2, 1, 2, 1, 2, 1, 2]
    torch.manual_seed(0)
    net = md.Discriminator((3, 96, 96), feats, strides).cuda().train()
    state = oinit.synth_state({k: tuple(v.shape) for k, v in net.state_dict().items()}, 6)
    g = torch.Generator().manual_seed(22)
    x = torch.rand(16, 3, 96, 96, generator=g) * 2 - 1
    r = torch.rand(16, 1, generator=g) * 2 - 1
    cfg = {'kind': 'discriminator', 'list_stride': strides}
    out = _flip_aware_compare(net, cfg, state, x, r,
                              strict_keys=('fc.', 'conv.2.2.', 'conv.2.3.', 'conv.2.4.', 'conv.2.5.', 'conv.2.6.'))
    assert tuple(out.shape) == (16, 1)


def test_full_size_discriminator_gradients_with_the_hip_paths_masks_forced():
    """The flip-INSENSITIVE full-size check: the same B16 / 96^2 discriminator case, but the fp32 oracle is made to take the
    HIP path's own LeakyReLU masks (oracle.ops.ACT_MASKS: the sign of every pre-activation as the HIP forward computed it, read from the
    state its backward keeps) -- both sides then differentiate the same piecewise-linear function, so EVERY gradient tensor, the three
    96^2 / 48^2 layers and the input gradient included, is held to the plain 1e-3 instead of the flip-sized bounds.  A kernel that reads
    a wrong halo column on an edge tile only at multi-tile sizes has nowhere to hide here."""
    from oracle import init as oinit
    E, md, de = pkg('engine'), pkg('model_discriminator'), pkg('discriminator_engine')
    E.set_precision('fp32')
    feats, strides = [64, 64, 128, 128, 256, 256, 512, 512], [1, 2, 1, 2, 1, 2, 1, 2]
    torch.manual_seed(0)
    net = md.Discriminator((3, 96, 96), feats, strides).cuda().train()
    state = oinit.synth_state({k: tuple(v.shape) for k, v in net.state_dict().items()}, 6)
    g = torch.Generator().manual_seed(22)
    x = torch.rand(16, 3, 96, 96, generator=g) * 2 - 1
    r = torch.rand(16, 1, generator=g) * 2 - 1
    cfg = {'kind': 'discriminator', 'list_stride': strides}
    nchw_ = lambda t: t.permute(0, 3, 1, 2).contiguous().cpu()

    def masks_of(sv):
        m = [nchw_(sv.c0.float() > 0)]
        for c, k in zip(sv.cs, sv.ks):
            m.append(nchw_(k[0].double() * c.double() + k[1].double() > 0))          # BatchNorm output = scale * conv + shift
        return m + [(sv.h1 > 0).cpu()]
    flips = _forced_mask_compare(de, net, cfg, state, x, r, masks_of)
    assert len(flips) == 9


def _forced_mask_compare(engine_mod, net, cfg, state, x, r, masks_of):
    """HIP forward+backward, then the fp32 oracle with the HIP path's activation masks (masks_of(saved state) in the oracle's forward
    order) forced: output and EVERY gradient tensor at the plain 1e-3.  -> number of mask elements the oracle alone decides otherwise"""
    from oracle import ops as oops
    from helpers import analytically_zero
    engine_mod.KEEP_SAVED = []
    try:
        out, gx, got, sd = _gpu_fwd_bwd(net, state, x, r)
        sv = engine_mod.KEEP_SAVED[-1]
    finally:
        engine_mod.KEEP_SAVED = None
    masks = masks_of(sv)
    own, orig = [], oops._positive
    oops._positive = lambda t: (own.append((t > 0).detach()), own[-1])[1]
    try:
        oracle_fwd_bwd(cfg, state, x, r)
    finally:
        oops._positive = orig
    assert len(own) == len(masks)
    flips = [int((a != b).sum()) for a, b in zip(own, masks)]
    oops.ACT_MASKS = list(masks)
    try:
        o_out, o_gx, o_grads, _ = oracle_fwd_bwd(cfg, state, x, r)
        assert oops.ACT_MASKS == []
    finally:
        oops.ACT_MASKS = None
    assert rel_err(out, o_out) < TOL
    got['grad_x'], o_grads['grad_x'] = gx, o_gx
    bad = [(k, rel_err(got[k], v)) for k, v in o_grads.items() if not analytically_zero(k, o_grads) and not rel_err(got[k], v) < TOL]
    assert bad == [], (bad, flips)
    return flips


@pytest.mark.parametrize('lr,init', [(48, 'default'), (96, 'synthetic')])
def test_full_size_generator_gradients_with_the_hip_paths_masks_forced(lr, init):
    """the flip-insensitive check for the generator: Generator(16, 64, 256, [2], use_sn=True), B16, LR 48 / LR 96, the reference's own
    PReLU slopes (0.25), fp32 parity build; the oracle takes the 18 PReLU masks of the HIP forward (first conv, 16 blocks, upscale), and
    every one of the ~150 gradient tensors and the input gradient is held to 1e-3 -- where test_full_size_generator_with_the_
    references_activations has to allow flip-sized 5e-2 / 5e-3"""
    E, ge = pkg('engine'), pkg('generator_engine')
    E.set_precision('fp32')
    net, cfg, state, x, r = _generator_case(lr, init)
    nchw_ = lambda t: t.permute(0, 3, 1, 2).contiguous().cpu()

    def masks_of(sv):
        m = [nchw_(sv.t0_pre.float() > 0)]
        for rec in sv.blocks:
            m.append(nchw_(rec.k1[0].double() * rec.c1.double() + rec.k1[1].double() > 0))
        return m + [nchw_(pre.float() > 0) for pre in sv.stage_pre]
    flips = _forced_mask_compare(ge, net, cfg, state, x, r, masks_of)
    assert len(flips) == 18
