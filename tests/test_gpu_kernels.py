"""GPU: kernel-level parity of the C ABI entry points against torch-CPU primitives (the same
primitives the oracle restates).  Tolerance 1e-4 relative (fp32 MFMA is an exact fmaf chain; only
summation order differs)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_helpers import FakeConv, maxrel, nchw, nhwc, pkg

pytestmark = pytest.mark.gpu
TOL = 2e-4


@pytest.fixture(scope='module')
def E():
    return pkg('engine')


@pytest.fixture(scope='module')
def L():
    return pkg('_lib')


def test_mfma_lane_layout(L):
    out = torch.zeros(32 * 32, device='cuda')
    L.check(L.lib().sisr_mfma_selftest(out.data_ptr(), torch.cuda.current_stream().cuda_stream), 'selftest')
    i = torch.arange(32, dtype=torch.float64)[:, None]
    j = torch.arange(32, dtype=torch.float64)[None, :]
    ref = sum((3 * i + 7 * k + 1) * (5 * k - 2 * j + 11) for k in range(8))
    assert torch.equal(out.cpu().double().reshape(32, 32), ref)


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


CONV_CASES = [
    # n, cin, cout, k, stride, h, w
    (2, 64, 64, 3, 1, 12, 12),
    (1, 64, 64, 3, 1, 37, 29),
    (2, 3, 64, 9, 1, 12, 12),
    (2, 64, 3, 3, 1, 10, 14),
    (2, 16, 16, 3, 1, 9, 8),
    (1, 4, 4, 3, 1, 16, 16),
    (1, 1, 3, 3, 1, 16, 16),
    (2, 32, 128, 3, 1, 8, 8),
    (2, 64, 128, 3, 2, 16, 16),
    (2, 128, 64, 3, 1, 6, 6),
    (3, 3, 16, 3, 1, 16, 16),
    (2, 48, 40, 3, 1, 7, 9),
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_forward_dgrad_wgrad(E, L, case):
    n, cin, cout, k, stride, h, w = case
    x = _rand((n, cin, h, w), 1)
    wt = _rand((cout, cin, k, k), 2, (1.0 / (cin * k * k)) ** 0.5 * 1.7)
    b = _rand((cout,), 3, 0.1)
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, br, stride=stride, padding=k // 2)
    r = _rand(tuple(y_ref.shape), 4)
    (y_ref * r).sum().backward()

    geom = E.ConvGeom(cin, cout, k, stride, k // 2)
    ref = FakeConv(wt.cuda(), b.cuda(), geom)
    preps, keep = E.prepare_weights([(ref, n, h, w)], training=True, need_dgrad=(stride == 1))
    p = preps[0]
    xd = nhwc(x).cuda()
    y, _, _ = E.conv_forward(p, E.Operand.plain(xd), bias=ref.bias)
    assert maxrel(nchw(y), y_ref) < TOL, 'forward'
    rd = nhwc(r).cuda()
    red = E.conv_wgrad(p, E.Operand.plain(xd), E.Operand.plain(rd))
    wg = E.WeightGradBatch()
    wg.add(p, red)
    gw, gb = wg.run()[id(ref)]
    assert maxrel(gw, wr.grad) < TOL, 'wgrad'
    assert maxrel(gb, br.grad) < TOL, 'bias grad'
    if stride == 1:
        dx = E.conv_dgrad(p, E.Operand.plain(rd))
        assert maxrel(nchw(dx), xr.grad) < TOL, 'dgrad'


def test_conv_nchw_in_shuffle_out_tanh_and_stats(E, L):
    n, h, w = 2, 10, 12
    # NCHW 3-channel input, 9x9
    x = _rand((n, 3, h, w), 5)
    w0 = _rand((32, 3, 9, 9), 6, 0.1)
    b0 = _rand((32,), 7, 0.1)
    g0 = E.ConvGeom(3, 32, 9, 1, 4)
    r0 = FakeConv(w0.cuda(), b0.cuda(), g0)
    # 32 -> 64 with PixelShuffle(2) on store
    w1 = _rand((64, 32, 3, 3), 8, 0.08)
    b1 = _rand((64,), 9, 0.1)
    g1 = E.ConvGeom(32, 64, 3, 1, 1, shuffle2=True)
    r1 = FakeConv(w1.cuda(), b1.cuda(), g1)
    # 16 -> 3, NCHW + tanh
    w2 = _rand((3, 16, 3, 3), 10, 0.15)
    b2 = _rand((3,), 11, 0.1)
    g2 = E.ConvGeom(16, 3, 3, 1, 1)
    r2 = FakeConv(w2.cuda(), b2.cuda(), g2)
    preps, keep = E.prepare_weights([(r0, n, h, w), (r1, n, h, w), (r2, n, 2 * h, 2 * w)], training=True)
    slope = torch.tensor([0.3], device='cuda')
    xin = x.cuda()
    y0, sp, cp = E.conv_forward(preps[0], E.Operand.plain(xin, dims=(n, h, w, 3), mode=L.X_NCHW), bias=r0.bias,
                                stats=True)
    ref0 = F.conv2d(x, w0, b0, padding=4)
    assert maxrel(nchw(y0), ref0) < TOL
    # statistics of y0
    bn = torch.nn.BatchNorm2d(32).cuda()
    k = E.bn_finalize(sp, cp, bn)
    mean = ref0.mean(dim=(0, 2, 3))
    var = ref0.var(dim=(0, 2, 3), unbiased=False)
    assert maxrel(k[2], mean) < 1e-4 and maxrel(k[3], torch.rsqrt(var + 1e-5)) < 1e-4
    assert maxrel(bn.running_var, 0.9 + 0.1 * ref0.var(dim=(0, 2, 3), unbiased=True)) < 1e-4
    # BN apply + PReLU prologue, PixelShuffle store
    y1, _, _ = E.conv_forward(preps[1], E.Operand.affine_act(y0, k[0], k[1], slope), bias=r1.bias)
    a0 = F.batch_norm(ref0, None, None, bn.weight.cpu(), bn.bias.cpu(), True, 0.1, 1e-5)
    a0 = torch.where(a0 > 0, a0, 0.3 * a0)
    ref1 = F.pixel_shuffle(F.conv2d(a0, w1, b1, padding=1), 2)
    assert tuple(y1.shape) == (n, 2 * h, 2 * w, 16)
    assert maxrel(nchw(y1), ref1) < TOL
    y2, _, _ = E.conv_forward(preps[2], E.Operand.act(y1, slope), bias=r2.bias, y_mode=L.Y_NCHW, epi=L.EPI_TANH)
    ref2 = torch.tanh(F.conv2d(torch.where(ref1 > 0, ref1, 0.3 * ref1), w2, b2, padding=1))
    assert maxrel(y2, ref2) < TOL


def test_spectral_norm_prepare_and_grad(E, L):
    cout, cin, k, n, h, w = 64, 32, 3, 2, 8, 8
    wt = _rand((cout, cin, k, k), 20, 0.1)
    b = _rand((cout,), 21, 0.1)
    u = F.normalize(_rand((cout,), 22), dim=0)
    v = F.normalize(_rand((cin * k * k,), 23), dim=0)
    x = _rand((n, cin, h, w), 24)
    # CPU: legacy spectral norm algorithm
    wr = wt.clone().requires_grad_(True)
    wm = wr.reshape(cout, -1)
    with torch.no_grad():
        v1 = F.normalize(torch.mv(wm.t(), u), dim=0, eps=1e-12)
        u1 = F.normalize(torch.mv(wm, v1), dim=0, eps=1e-12)
    sigma = torch.dot(u1, torch.mv(wm, v1))
    y_ref = F.conv2d(x, wr / sigma, b, padding=1)
    r = _rand(tuple(y_ref.shape), 25)
    (y_ref * r).sum().backward()
    geom = E.ConvGeom(cin, cout, k, 1, 1)
    ref = FakeConv(wt.cuda(), b.cuda(), geom, u.cuda(), v.cuda())
    preps, keep = E.prepare_weights([(ref, n, h, w)], training=True)
    p = preps[0]
    assert maxrel(ref.u, u1) < 1e-5 and maxrel(ref.v, v1) < 1e-5 and maxrel(p.sigma, sigma.reshape(1)) < 1e-5
    y, _, _ = E.conv_forward(p, E.Operand.plain(nhwc(x).cuda()), bias=ref.bias)
    assert maxrel(nchw(y), y_ref) < TOL
    red = E.conv_wgrad(p, E.Operand.plain(nhwc(x).cuda()), E.Operand.plain(nhwc(r).cuda()))
    wg = E.WeightGradBatch()
    wg.add(p, red)
    gw, gb = wg.run()[id(ref)]
    assert maxrel(gw, wr.grad) < TOL
    # eval mode: no power iteration
    ref2 = FakeConv(wt.cuda(), b.cuda(), geom, u.cuda(), v.cuda())
    preps2, _ = E.prepare_weights([(ref2, n, h, w)], training=False)
    assert torch.equal(ref2.u.cpu(), u) and maxrel(preps2[0].sigma, torch.dot(u, torch.mv(wt.reshape(cout, -1), v)).reshape(1)) < 1e-5


def test_bn_backward_and_eltwise(E, L):
    n, h, w, c = 2, 9, 7, 64
    x = _rand((n, c, h, w), 30, 2.0)
    dy = _rand((n, c, h, w), 31)
    gamma = _rand((c,), 32) + 1.5
    beta = _rand((c,), 33)
    slope = 0.2
    xr = x.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    sl = torch.tensor([slope], requires_grad=True)
    z = F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    a = torch.where(z > 0, z, sl * z)
    (a * dy).sum().backward()
    mean = x.mean(dim=(0, 2, 3))
    invstd = torch.rsqrt(x.var(dim=(0, 2, 3), unbiased=False) + 1e-5)
    scale = gamma * invstd
    shift = beta - mean * scale
    k = torch.stack([scale, shift, mean, invstd]).cuda()
    xd, dyd = nhwc(x).cuda(), nhwc(dy).cuda()
    q, dgam, dbet, dsl = E.bn_backward(dyd, xd, k, gamma.cuda(), slope=torch.tensor([slope], device='cuda'))
    assert maxrel(dgam, gr.grad) < TOL and maxrel(dbet, br.grad) < TOL and maxrel(dsl, sl.grad) < TOL
    # dx through the BNACT_BWD prologue of a 1x1 identity conv
    eye = torch.eye(c).reshape(c, c, 1, 1)
    geom = E.ConvGeom(c, c, 1, 1, 0)
    ref = FakeConv(eye.cuda(), None, geom)
    preps, _ = E.prepare_weights([(ref, n, h, w)], training=True)
    op = E.Operand(dyd, (n, h, w, c), pro=L.PRO_BNACT_BWD, x2=xd, pa=q[0], pb=q[1], pd=q[2], ps=k[0], pt=k[1],
                   slope=torch.tensor([slope], device='cuda'))
    dx, _, _ = E.conv_forward(preps[0], op)
    assert maxrel(nchw(dx), xr.grad) < TOL
    y = E.eltwise_res_affine(xd, 0.25, dyd, k[0], k[1])
    ref_y = torch.where(x > 0, x, 0.25 * x) + dy * scale[None, :, None, None] + shift[None, :, None, None]
    assert maxrel(nchw(y), ref_y) < 1e-6
    out = E.prelu_slope_grad(dyd, xd)
    assert maxrel(out, (dy * x)[x <= 0].sum().reshape(1)) < TOL


def test_bicubic_against_golden(L, golden_dir):
    z = np.load(golden_dir + '/bicubic.npz')
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    for i in range(int(z['n'])):
        x = torch.from_numpy(z['x%d' % i]).cuda()
        oh, ow = (int(v) for v in z['size%d' % i])
        n, c, h, w = x.shape
        y = torch.empty((n, c, oh, ow), device='cuda')
        L.check(lib.sisr_bicubic_fwd(x.data_ptr(), y.data_ptr(), n * c, h, w, oh, ow, 0, st), 'bicubic')
        assert float((y.cpu() - torch.from_numpy(z['interp%d' % i])).abs().max()) < 5e-6
        L.check(lib.sisr_bicubic_fwd(x.data_ptr(), y.data_ptr(), n * c, h, w, oh, ow, 1, st), 'bicubic')
        assert float((y.cpu() - torch.from_numpy(z['lr%d' % i])).abs().max()) < 5e-6
        dy = torch.from_numpy(z['r%d' % i]).cuda()
        dx = torch.empty_like(x)
        L.check(lib.sisr_bicubic_bwd(dy.data_ptr(), None, dx.data_ptr(), n * c, h, w, oh, ow, st), 'bicubic_bwd')
        assert float((dx.cpu() - torch.from_numpy(z['grad_x%d' % i])).abs().max()) < 2e-5


# ---- bf16 matrix-core kernel family (fp32 tensors in HBM, bf16 MFMA, fp32 accumulate) -------------------
def _walk(shape, monkeypatch):
    """shape = (n, h, w) or (n, h, w, cap).  cap: SISR_PERSIST_MAX_WG, the number of workgroup slots the persistent
    kernels may fill -- with a small cap a small input walks SEVERAL tiles per workgroup (double-buffer swap, T+2
    prefetch, statistics / reductions / weight gradient carried across tiles), the schedule the B16 96x96 launches of
    the benchmark run; the (16, 96, 96) / (16, 192, 192) entries are those launches themselves (1,152 tiles)."""
    if len(shape) == 4:
        monkeypatch.setenv('SISR_PERSIST_MAX_WG', str(shape[3]))
    return shape[:3]


TRUNK_SHAPES = [(2, 16, 32), (3, 24, 48), (1, 96, 96), (3, 24, 48, 5), (16, 96, 96)]
TRUNK_SHAPES_SHORT = [(2, 16, 32), (1, 96, 96), (3, 24, 48, 5), (16, 96, 96)]

BF16_TOL = 2e-2      # bf16 has 8 significant bits: operands rounded to 2^-9 relative, fp32 accumulation


def test_tr16_lane_roles(L):
    """ds_read_b64_tr_b16 hands lane (group g, i) channel 16*(g&1)+i of pixels 8*(g>>1)+0..7"""
    out = torch.zeros(64 * 8, dtype=torch.int16, device='cuda')
    L.check(L.lib().sisr_tr16_selftest(out.data_ptr(), torch.cuda.current_stream().cuda_stream), 'tr16')
    got = out.cpu().reshape(64, 8)
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        exp = [(8 * (g >> 1) + j) * 64 + 16 * (g & 1) + i for j in range(8)]
        assert got[lane].tolist() == exp, (lane, got[lane].tolist(), exp)


BF16_CASES = [(2, 64, 64, 3, 1, 12, 12), (1, 64, 64, 3, 1, 37, 29), (2, 32, 128, 3, 1, 8, 8), (2, 128, 64, 3, 1, 6, 6),
              (2, 64, 256, 3, 1, 16, 16), (2, 64, 128, 3, 2, 16, 16), (1, 64, 64, 3, 1, 96, 96),
              # the discriminator's deep layers (small maps, many channels: the planner's 32-cout tiles)
              (4, 256, 512, 3, 1, 12, 12), (4, 512, 512, 3, 2, 12, 12), (3, 128, 256, 3, 2, 24, 24)]


def _storage(monkeypatch, storage):
    """bf16 build with fp32 tensors (round-1 layout, SISR_STORAGE=f32) or with bf16 tensors in HBM (the default)"""
    monkeypatch.setenv('SISR_STORAGE', storage)
    return torch.bfloat16 if storage == 'bf16' else torch.float32


@pytest.mark.parametrize('storage', ['f32', 'bf16'])
@pytest.mark.parametrize('case', BF16_CASES)
def test_conv_bf16_forward_dgrad_wgrad(E, L, case, storage, monkeypatch):
    dt = _storage(monkeypatch, storage)
    n, cin, cout, k, stride, h, w = case
    x = _rand((n, cin, h, w), 1).to(dt).float()          # values representable in the storage type
    wt = _rand((cout, cin, k, k), 2, (1.0 / (cin * k * k)) ** 0.5 * 1.7)
    b = _rand((cout,), 3, 0.1)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, br, stride=stride, padding=k // 2)
    r = _rand(tuple(y_ref.shape), 4).to(dt).float()
    (y_ref * r).sum().backward()
    E.set_precision('bf16')
    try:
        geom = E.ConvGeom(cin, cout, k, stride, k // 2)
        ref = FakeConv(wt.cuda(), b.cuda(), geom)
        preps, keep = E.prepare_weights([(ref, n, h, w)], training=True)
        p = preps[0]
        assert p.kinds[0] and p.kinds[2] and (p.kinds[1] or stride == 2)
        xd, rd = nhwc(x).cuda().to(dt), nhwc(r).cuda().to(dt)
        y, sp, cp = E.conv_forward(p, E.Operand.plain(xd), bias=ref.bias, stats=True)
        assert y.dtype == dt
        assert maxrel(nchw(y.float()), y_ref) < BF16_TOL, 'forward'
        # BatchNorm statistics of the epilogue (per-tile count / mean / M2 merged here in float64)
        cnt, mean_t, m2_t = cp.double().cpu(), sp[:, 0].double().cpu(), sp[:, 1].double().cpu()
        tot = cnt.sum()
        mean = (cnt[:, None] * mean_t).sum(0) / tot
        var = (m2_t + cnt[:, None] * (mean_t - mean) ** 2).sum(0) / tot
        yr = y_ref.detach().double()
        assert maxrel(mean, yr.mean(dim=(0, 2, 3))) < BF16_TOL and maxrel(var, yr.var(dim=(0, 2, 3), unbiased=False)) < BF16_TOL
        red = E.conv_wgrad(p, E.Operand.plain(xd), E.Operand.plain(rd))
        wg = E.WeightGradBatch()
        wg.add(p, red)
        gw, gb = wg.run()[id(ref)]
        assert maxrel(gw, wr.grad) < BF16_TOL, 'wgrad'
        assert maxrel(gb, br.grad) < BF16_TOL, 'bias grad'
        dx = E.conv_dgrad(p, E.Operand.plain(rd))
        assert dx.dtype == dt
        assert maxrel(nchw(dx.float()), xr.grad) < BF16_TOL, 'dgrad'
    finally:
        E.set_precision('fp32')


def test_wgrad_bf16_few_channel_output_padded(E, L):
    """the generator's last conv (64 -> 3 + tanh): its weight gradient runs on the bf16 kernel over a
    4-channel NHWC copy of the NCHW gradient with the tanh backward fused into the copy"""
    n, cin, cout, k, h, w = 2, 64, 3, 3, 20, 24
    x = _rand((n, cin, h, w), 11)
    wt = _rand((cout, cin, k, k), 12, 0.05)
    b = _rand((cout,), 13, 0.1)
    xr, wr, br = x.clone(), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = torch.tanh(F.conv2d(xr, wr, br, padding=1))
    r = _rand(tuple(y_ref.shape), 14)
    (y_ref * r).sum().backward()
    E.set_precision('bf16')
    try:
        ref = FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(cin, cout, k, 1, 1))
        p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
        assert p.kinds[2] and p.plans[2].Cout == 4
        dy = E.Operand(r.cuda(), (n, h, w, cout), pro=L.PRO_TANH_BWD, mode=L.X_NCHW, x2=y_ref.detach().cuda())
        red = E.conv_wgrad(p, E.Operand.plain(nhwc(x).cuda()), dy)
        wg = E.WeightGradBatch()
        wg.add(p, red)
        gw, gb = wg.run()[id(ref)]
        assert maxrel(gw, wr.grad) < BF16_TOL, 'wgrad'
        assert maxrel(gb, br.grad) < BF16_TOL, 'bias grad'
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('storage', ['f32', 'bf16'])
@pytest.mark.parametrize('shape,act', [((2, 64, 64, 12, 12), True), ((1, 64, 64, 37, 29), False), ((3, 32, 64, 6, 6), True)])
def test_bn_backward_reductions_from_the_conv_epilogue(E, L, shape, act, storage, monkeypatch):
    """conv_dgrad(..., bnb=...): the data-gradient conv (bf16 kernel) also emits the backward reductions of the
    BatchNorm its output arrives at; they must equal the stand-alone reduction kernel run on the same gradient"""
    dt = _storage(monkeypatch, storage)
    n, cin, cout, h, w = shape
    wt = _rand((cout, cin, 3, 3), 51, (1.0 / (cin * 9)) ** 0.5 * 1.7)
    dy, skip, x = _rand((n, cout, h, w), 52), _rand((n, cin, h, w), 53), _rand((n, cin, h, w), 54, 2.0)
    x = x.to(dt).float()
    gamma, beta = _rand((cin,), 55) + 1.5, _rand((cin,), 56)
    mean = x.mean(dim=(0, 2, 3))
    invstd = torch.rsqrt(x.var(dim=(0, 2, 3), unbiased=False) + 1e-5)
    k = torch.stack([gamma * invstd, beta - mean * gamma * invstd, mean, invstd]).cuda()
    slope = torch.tensor([0.2], device='cuda') if act else None
    E.set_precision('bf16')
    try:
        ref = FakeConv(wt.cuda(), None, E.ConvGeom(cin, cout, 3, 1, 1))
        p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
        assert E.can_fuse_bn_backward(p)
        xd = nhwc(x).cuda().to(dt)
        g, part = E.conv_dgrad(p, E.Operand.plain(nhwc(dy).cuda().to(dt)), res=nhwc(skip).cuda().to(dt), bnb=(xd, k, slope))
        assert g.dtype == dt
        # the gradient itself: conv-transpose of dy plus the skip gradient
        g_ref = F.conv_transpose2d(dy.to(dt).float(), wt, padding=1) + skip.to(dt).float()
        assert maxrel(nchw(g.float()), g_ref) < BF16_TOL
        fused = E.bn_backward(g, xd, k, gamma.cuda(), slope=slope, part=part)
        plain = E.bn_backward(g, xd, k, gamma.cuda(), slope=slope)
        # fp32 tensors: both read the same gradient.  bf16 tensors: the epilogue sums the fp32 accumulators, the
        # stand-alone kernel the gradient after its rounding to bf16 (2^-9 relative per element, zero mean)
        tol = 1e-4 if storage == 'f32' else 5e-3
        for a, b in zip(fused, plain):
            if a is not None:
                assert maxrel(a, b) < tol
    finally:
        E.set_precision('fp32')


def test_bn_backward_finish_carries_a_slab_reduction(E, L):
    """sisr_bn_bwd_finalize_slab: ONE launch finishes a BatchNorm backward from partial rows and sums the slabs of an
    unrelated weight gradient in additional workgroups -- both results bit-identical to the two separate launches
    (sisr_bn_bwd_finalize, sisr_slab_reduce_f32), for slab widths that do and do not fill the last workgroup"""
    import ctypes as C
    lib = L.lib()
    cch, rows = 64, 231
    g = torch.Generator().manual_seed(5)
    part = (torch.rand(rows, 2 * cch + 1, generator=g) - 0.5).cuda()
    x = torch.rand(4, 8, 8, cch, generator=g).cuda()
    consts = (torch.rand(4, cch, generator=g) + 0.5).cuda()
    gamma = (torch.rand(cch, generator=g) + 0.5).cuda()
    slope = torch.tensor([0.25], device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    for n_slabs, stride in ((231, 36928), (7, 20), (40, 16 * 4 * 5 + 4)):
        slab = (torch.rand(n_slabs, stride, generator=g) - 0.5).cuda()
        want_red = torch.empty(stride, device='cuda')
        L.check(lib.sisr_slab_reduce_f32(slab.data_ptr(), want_red.data_ptr(), n_slabs, stride, 0, st), 'slab_reduce')
        want = E.bn_backward(x, x, consts, gamma, slope=slope, part=part)
        pend = E.PendingSlabs()
        red = torch.full((stride,), float('nan'), device='cuda')
        pend.jobs.append((slab, red, n_slabs, stride, 0))
        got = E.bn_backward(x, x, consts, gamma, slope=slope, part=part, slabs=pend)
        assert pend.jobs == []
        assert torch.equal(red, want_red)
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    # rows whose first `lead` elements are bf16 at the row's start (the persistent bf16 weight-gradient kernel's slabs), the rest
    # fp32 at their float offset: exactly the sum of the values as stored, through both entry points
    for n_slabs, stride, lead in ((231, 36928, 36864), (9, 48, 32), (40, 16 * 4 * 5 + 4, 16 * 4 * 4)):
        vals = torch.rand(n_slabs, stride, generator=g) - 0.5
        lead_vals = vals[:, :lead].bfloat16()
        slab = (torch.rand(n_slabs, stride, generator=g) * 1e6).cuda()                 # (the hole behind the bf16 part holds junk)
        slab[:, lead:] = vals[:, lead:].cuda()
        slab.view(torch.bfloat16)[:, :lead] = lead_vals.cuda()
        want_red = torch.cat([lead_vals.double().sum(0), vals[:, lead:].double().sum(0)])
        red = torch.full((stride,), float('nan'), device='cuda')
        L.check(lib.sisr_slab_reduce_f32(slab.data_ptr(), red.data_ptr(), n_slabs, stride, lead, st), 'slab_reduce (bf16 lead)')
        assert maxrel(red, want_red) < 1e-5
        pend = E.PendingSlabs()
        red2 = torch.full((stride,), float('nan'), device='cuda')
        pend.jobs.append((slab, red2, n_slabs, stride, lead))
        E.bn_backward(x, x, consts, gamma, slope=slope, part=part, slabs=pend)
        assert torch.equal(red2, red)


def test_several_slab_reductions_in_one_launch(E, L):
    """sisr_slab_reduce_multi (PendingSlabs.flush): up to eight jobs per launch, each bit-identical to sisr_slab_reduce_f32 on the same
    slabs -- many slabs (16 columns x 16 splits per workgroup) and few (one column per thread) mixed in one launch, fp32 and
    bf16-lead rows, eleven jobs = two launches"""
    lib = L.lib()
    g = torch.Generator().manual_seed(9)
    st = torch.cuda.current_stream().cuda_stream
    shapes = [(231, 36928, 36864), (1, 4096 + 64, 4096), (3, 1028, 0), (16, 260, 256), (17, 260, 256), (64, 36928, 36864), (2, 64, 64),
              (5, 2304 + 64, 0), (12, 148, 128), (40, 16 * 4 * 5 + 4, 0), (4, 589824 + 512, 589824)]
    pend, want = E.PendingSlabs(), []
    for n_slabs, stride, lead in shapes:
        slab = (torch.rand(n_slabs, stride, generator=g) - 0.5).cuda()
        if lead:
            slab.view(torch.bfloat16)[:, :lead] = (torch.rand(n_slabs, lead, generator=g) - 0.5).bfloat16().cuda()
        ref = torch.empty(stride, device='cuda')
        L.check(lib.sisr_slab_reduce_f32(slab.data_ptr(), ref.data_ptr(), n_slabs, stride, lead, st), 'slab_reduce')
        red = torch.full((stride,), float('nan'), device='cuda')
        pend.jobs.append((slab, red, n_slabs, stride, lead))
        want.append((red, ref))
    pend.flush()
    assert pend.jobs == []
    for i, (red, ref) in enumerate(want):
        assert torch.equal(red, ref), shapes[i]


def _merged_stats(sp, cp):
    cnt, mean_t, m2_t = cp.double().cpu(), sp[:, 0].double().cpu(), sp[:, 1].double().cpu()
    tot = cnt.sum()
    mean = (cnt[:, None] * mean_t).sum(0) / tot
    var = (m2_t + cnt[:, None] * (mean_t - mean) ** 2).sum(0) / tot
    return float(tot), mean, var


@pytest.mark.parametrize('pro', ['none', 'act', 'affine_act'])
@pytest.mark.parametrize('shape', TRUNK_SHAPES)
def test_trunk_kernel_matches_the_generic_kernel_and_the_reference(E, L, shape, pro, monkeypatch):
    """conv_trunk.hip (persistent, weights in registers; 3x3 64->64 on bf16 tensors, H % 8 == 0, W % 16 == 0) against
    the generic bf16 kernel on the same operands and against F.conv2d: output, and the BatchNorm statistics merged from
    its per-workgroup partials"""
    n, h, w = _walk(shape, monkeypatch)
    monkeypatch.setenv('SISR_STORAGE', 'bf16')
    x = (_rand((n, 64, h, w), 61) * 2.0).bfloat16().float()
    wt = _rand((64, 64, 3, 3), 62, (1.0 / 576) ** 0.5 * 1.7)
    b = _rand((64,), 63, 0.1)
    sc, sh = _rand((64,), 64) * 0.5 + 1.0, _rand((64,), 65) * 0.3
    slope = torch.tensor([0.2])
    xin = x
    if pro == 'act':
        xin = F.leaky_relu(x, 0.2)
    elif pro == 'affine_act':
        xin = F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.2)
    y_ref = F.conv2d(xin, wt, b, padding=1)
    E.set_precision('bf16')
    try:
        ref = FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(64, 64, 3, 1, 1))
        p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
        xd = nhwc(x).cuda().bfloat16()
        if pro == 'none':
            op = E.Operand.plain(xd)
        elif pro == 'act':
            op = E.Operand.act(xd, slope.cuda())
        else:
            op = E.Operand.affine_act(xd, sc.cuda(), sh.cuda(), slope.cuda())
        res = {}
        for sw in ('1', '0'):
            monkeypatch.setenv('SISR_TRUNK', sw)
            y, sp, cp = E.conv_forward(p, op, bias=ref.bias, stats=True)
            res[sw] = (y.float(), sp, cp)
        assert res['1'][1].shape[0] <= min(res['0'][1].shape[0], 512)  # one partial per workgroup, not per tile
        assert maxrel(nchw(res['1'][0]), y_ref) < BF16_TOL
        assert maxrel(res['1'][0], res['0'][0]) < 6e-3                 # same arithmetic, different fp32 summation order
        t1, m1, v1 = _merged_stats(res['1'][1], res['1'][2])
        t0, m0, v0 = _merged_stats(res['0'][1], res['0'][2])
        assert t1 == t0 == n * h * w
        assert maxrel(m1, m0) < 1e-4 and maxrel(v1, v0) < 1e-4
        yr = y_ref.double()
        assert maxrel(m1, yr.mean(dim=(0, 2, 3))) < BF16_TOL and maxrel(v1, yr.var(dim=(0, 2, 3), unbiased=False)) < BF16_TOL
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('pro,res,bnb', [('bnbwd', False, None), ('bnbwd', True, 'plain'), ('bnact_bwd', True, 'act'),
                                         ('bnact_bwd', False, 'plain'), ('bnact_bwd', True, None), ('bnbwd', False, 'act')])
@pytest.mark.parametrize('shape', TRUNK_SHAPES)
def test_trunk_kernel_data_gradient_role(E, L, shape, pro, res, bnb, monkeypatch):
    """conv_trunk.hip, data-gradient role: two-tensor BatchNorm-backward prologue (with / without the activation),
    skip gradient added in the epilogue, backward reductions of the next BatchNorm from the epilogue -- against the
    generic bf16 kernel on the same operands and against conv_transpose2d of the prologue written out in fp32"""
    n, h, w = _walk(shape, monkeypatch)
    monkeypatch.setenv('SISR_STORAGE', 'bf16')
    bf = lambda t: t.bfloat16().float()
    g_in, c = bf(_rand((n, 64, h, w), 71)), bf(_rand((n, 64, h, w), 72) * 2.0)
    wt = _rand((64, 64, 3, 3), 73, (1.0 / 576) ** 0.5 * 1.7)
    qa, qb, qd = _rand((64,), 74) * 0.3 + 1.0, _rand((64,), 75) * 0.2, _rand((64,), 76) * 0.1
    ks, kt = _rand((64,), 77) * 0.5 + 1.0, _rand((64,), 78) * 0.3
    slope = torch.tensor([0.2])
    skip = bf(_rand((n, 64, h, w), 79))
    xb = bf(_rand((n, 64, h, w), 80) * 2.0)
    bc = lambda v: v[None, :, None, None]
    gg = g_in
    if pro == 'bnact_bwd':
        z = bc(ks) * c + bc(kt)
        gg = torch.where(z > 0, g_in, 0.2 * g_in)
    dy_ref = bc(qa) * gg + bc(qb) * c + bc(qd)
    out_ref = F.conv_transpose2d(dy_ref, wt, padding=1) + (skip if res else 0.0)
    gamma, beta = _rand((64,), 81) + 1.5, _rand((64,), 82)
    mean = xb.mean(dim=(0, 2, 3))
    invstd = torch.rsqrt(xb.var(dim=(0, 2, 3), unbiased=False) + 1e-5)
    kb = torch.stack([gamma * invstd, beta - mean * gamma * invstd, mean, invstd]).cuda()
    E.set_precision('bf16')
    try:
        ref = FakeConv(wt.cuda(), None, E.ConvGeom(64, 64, 3, 1, 1))
        p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
        gd, cd = nhwc(g_in).cuda().bfloat16(), nhwc(c).cuda().bfloat16()
        kw = dict(pa=qa.cuda(), pb=qb.cuda(), pd=qd.cuda())
        if pro == 'bnact_bwd':
            kw.update(ps=ks.cuda(), pt=kt.cuda(), slope=slope.cuda())
        op = E.Operand(gd, tuple(cd.shape), pro=L.PRO_BNACT_BWD if pro == 'bnact_bwd' else L.PRO_BNBWD, x2=cd, **kw)
        rd = nhwc(skip).cuda().bfloat16() if res else None
        xd = nhwc(xb).cuda().bfloat16()
        b_slope = slope.cuda() if bnb == 'act' else None
        out = {}
        for sw in ('1', '0'):
            monkeypatch.setenv('SISR_TRUNK', sw)
            r = E.conv_dgrad(p, op, res=rd, bnb=None if bnb is None else (xd, kb, b_slope))
            out[sw] = r if bnb is not None else (r, None)
        assert maxrel(nchw(out['1'][0].float()), out_ref) < BF16_TOL
        assert maxrel(out['1'][0].float(), out['0'][0].float()) < 6e-3
        if bnb is not None:
            assert out['1'][1].shape[0] <= min(out['0'][1].shape[0], 512)
            s1, s0 = out['1'][1].double().sum(0), out['0'][1].double().sum(0)
            assert maxrel(s1[:128], s0[:128]) < 2e-3                       # sum(g), sum(g * xhat) per channel
            if bnb == 'act':
                assert abs(float(s1[128] - s0[128])) <= 2e-3 * max(1.0, abs(float(s0[128])))
            fused = E.bn_backward(out['1'][0], xd, kb, gamma.cuda(), slope=b_slope, part=out['1'][1])
            plain = E.bn_backward(out['1'][0], xd, kb, gamma.cuda(), slope=b_slope)
            for a, b in zip(fused, plain):
                if a is not None:
                    assert maxrel(a, b) < 5e-3
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('xpro,gpro', [('none', 'bnbwd'), ('act', 'bnact_bwd'), ('affine_act', 'bnbwd'), ('affine_act', 'bnact_bwd')])
@pytest.mark.parametrize('shape', TRUNK_SHAPES)
def test_trunk_kernel_weight_gradient_role(E, L, shape, xpro, gpro, monkeypatch):
    """wgrad_trunk.hip (persistent, the whole 64 x 576 gradient in accumulators, one slab per workgroup) against the
    generic bf16 weight-gradient kernel on the same lazy operands and against autograd on the prologues written out in
    fp32: packed gradient, un-packed weight gradient, bias gradient"""
    n, h, w = _walk(shape, monkeypatch)
    monkeypatch.setenv('SISR_STORAGE', 'bf16')
    bf = lambda t: t.bfloat16().float()
    bc = lambda v: v[None, :, None, None]
    x = bf(_rand((n, 64, h, w), 91) * 2.0)
    g_in, c = bf(_rand((n, 64, h, w), 92)), bf(_rand((n, 64, h, w), 93) * 2.0)
    wt = _rand((64, 64, 3, 3), 94, (1.0 / 576) ** 0.5)
    b = _rand((64,), 95, 0.1)
    sc, sh = _rand((64,), 96) * 0.5 + 1.0, _rand((64,), 97) * 0.3
    qa, qb, qd = _rand((64,), 98) * 0.3 + 1.0, _rand((64,), 99) * 0.2, _rand((64,), 100) * 0.1
    ks, kt = _rand((64,), 101) * 0.5 + 1.0, _rand((64,), 102) * 0.3
    slope = torch.tensor([0.2])
    xin = x
    if xpro == 'act':
        xin = F.leaky_relu(x, 0.2)
    elif xpro == 'affine_act':
        xin = F.leaky_relu(x * bc(sc) + bc(sh), 0.2)
    gg = g_in
    if gpro == 'bnact_bwd':
        gg = torch.where(bc(ks) * c + bc(kt) > 0, g_in, 0.2 * g_in)
    dy_ref = bc(qa) * gg + bc(qb) * c + bc(qd)
    wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    (F.conv2d(xin, wr, br, padding=1) * dy_ref).sum().backward()
    E.set_precision('bf16')
    try:
        ref = FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(64, 64, 3, 1, 1))
        p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
        xd = nhwc(x).cuda().bfloat16()
        if xpro == 'none':
            x_op = E.Operand.plain(xd)
        elif xpro == 'act':
            x_op = E.Operand.act(xd, slope.cuda())
        else:
            x_op = E.Operand.affine_act(xd, sc.cuda(), sh.cuda(), slope.cuda())
        gd, cd = nhwc(g_in).cuda().bfloat16(), nhwc(c).cuda().bfloat16()
        kw = dict(pa=qa.cuda(), pb=qb.cuda(), pd=qd.cuda())
        if gpro == 'bnact_bwd':
            kw.update(ps=ks.cuda(), pt=kt.cuda(), slope=slope.cuda())
        dy_op = E.Operand(gd, tuple(cd.shape), pro=L.PRO_BNACT_BWD if gpro == 'bnact_bwd' else L.PRO_BNBWD, x2=cd, **kw)
        red = {}
        for sw in ('1', '0'):
            monkeypatch.setenv('SISR_TRUNK_WGRAD', sw)
            red[sw] = E.conv_wgrad(p, x_op, dy_op)
        assert maxrel(red['1'], red['0']) < 5e-3                     # same products; the persistent kernel's partial sums are stored as bf16 (2^-9 each)
        wg = E.WeightGradBatch()
        wg.add(p, red['1'])
        gw, gb = wg.run()[id(ref)]
        assert maxrel(gw, wr.grad) < BF16_TOL, 'wgrad'
        assert maxrel(gb, br.grad) < BF16_TOL, 'bias grad'
        # deterministic: a second launch reproduces the first bit for bit
        monkeypatch.setenv('SISR_TRUNK_WGRAD', '1')
        assert torch.equal(E.conv_wgrad(p, x_op, dy_op), red['1'])
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('precision', ['bf16', 'fp32', 'bf16x3'])
def test_trunk_weight_gradients_of_several_layers_in_one_launch(E, L, precision, monkeypatch):
    """engine.WgradDeepBatch._run_trunk -> sisr_wgrad_trunk_batch / sisr_wgrad_trunk_f32_batch: five trunk layers (three with the
    BatchNorm-backward prologue, two with the activation form: two launches, workgroups [z * wpl, (z + 1) * wpl) serve layer z) against
    the same layers launched alone -- same per-tile arithmetic, only the partition into slabs differs -- at (16, 96, 96), the size the
    benchmark runs, and at a small size walked by few workgroups"""
    bc = lambda v: v[None, :, None, None]
    monkeypatch.setenv('SISR_STORAGE', 'bf16' if precision == 'bf16' else 'f32')
    monkeypatch.setenv('SISR_WGRAD_BATCH_TRUNK_PIXELS', '0')      # (the bf16 build hands small trunk layers to wgrad_deep's batch: not here)
    E.set_precision(precision)
    try:
        for n, h, w in ((16, 96, 96), (2, 16, 32)):
            members = []
            for i, gpro in enumerate(('bnbwd', 'bnact_bwd', 'bnbwd', 'bnbwd', 'bnact_bwd')):
                s0 = 400 + 20 * i
                wt = _rand((64, 64, 3, 3), s0, (1.0 / 576) ** 0.5)
                ref = FakeConv(wt.cuda(), _rand((64,), s0 + 1, 0.1).cuda(), E.ConvGeom(64, 64, 3, 1, 1))
                p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
                dt = torch.bfloat16 if p.kinds[2] else torch.float32
                xd = nhwc(_rand((n, 64, h, w), s0 + 2) * 2.0).cuda().to(dt)
                gd, cd = nhwc(_rand((n, 64, h, w), s0 + 3)).cuda().to(dt), nhwc(_rand((n, 64, h, w), s0 + 4) * 2.0).cuda().to(dt)
                x_op = E.Operand.affine_act(xd, (_rand((64,), s0 + 5) * 0.5 + 1.0).cuda(), (_rand((64,), s0 + 6) * 0.3).cuda(), torch.tensor([0.2]).cuda())
                kw = dict(pa=(_rand((64,), s0 + 7) * 0.3 + 1.0).cuda(), pb=(_rand((64,), s0 + 8) * 0.2).cuda(), pd=(_rand((64,), s0 + 9) * 0.1).cuda())
                if gpro == 'bnact_bwd':
                    kw.update(ps=(_rand((64,), s0 + 10) * 0.5 + 1.0).cuda(), pt=(_rand((64,), s0 + 11) * 0.3).cuda(), slope=torch.tensor([0.2]).cuda())
                dy_op = E.Operand(gd, tuple(cd.shape), pro=L.PRO_BNACT_BWD if gpro == 'bnact_bwd' else L.PRO_BNBWD, x2=cd, **kw)
                members.append((p, ref, x_op, dy_op))
            alone = [E.conv_wgrad(p, x_op, dy_op) for p, ref, x_op, dy_op in members]
            wb, pending = E.WgradDeepBatch(), E.PendingSlabs()
            before = E.KERNEL_COUNTS.get('wgrad_trunk_batch', 0)
            reds = [wb.add(p, x_op, dy_op) for p, ref, x_op, dy_op in members]
            assert all(r is not None for r in reds) and len(wb.trunk) == 5 and wb.items == []
            wb.run(pending)
            assert E.KERNEL_COUNTS.get('wgrad_trunk_batch', 0) == before + 2 and len(pending.jobs) == 5
            pending.flush()
            tol = 5e-3 if precision == 'bf16' else (1e-5 if precision == 'fp32' else 1e-4)       # (bf16: slabs rounded to bf16, 2^-9 each)
            # compared un-packed: the fp32 slab layout has padding rows that no kernel writes (whatever the allocation held)
            for reds_ in (reds, alone):
                wg = E.WeightGradBatch()
                for (p, ref, _, _), r in zip(members, reds_):
                    wg.add(p, r)
                reds_[:] = [wg.run()[id(ref)] for p, ref, _, _ in members]
            for (gw, gb), (gw1, gb1) in zip(reds, alone):
                assert maxrel(gw, gw1) < tol and maxrel(gb, gb1) < tol, (precision, n, h, w, maxrel(gw, gw1), maxrel(gb, gb1))
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('xpro,gpro', [('none', 'bnbwd'), ('act', 'bnact_bwd'), ('affine_act', 'bnbwd'), ('affine_act', 'bnact_bwd')])
@pytest.mark.parametrize('shape', [(2, 12, 16)] + TRUNK_SHAPES[1:])
def test_trunk_kernel_weight_gradient_role_fp32(E, L, shape, xpro, gpro, precision, monkeypatch):
    """wgrad_trunk_f32.hip (fp32 tensors, persistent accumulators, 4 x 16 tiles; 'fp32': exact fp32 MFMA, 'bf16x3': bf16 MFMA
    over hi / lo pairs of both fp32 operands) against the generic fp32 weight-gradient kernel on the same lazy operands and
    against autograd: packed gradient, un-packed weight gradient, bias gradient, bit-identical replay"""
    n, h, w = _walk(shape, monkeypatch)
    bc = lambda v: v[None, :, None, None]
    x = _rand((n, 64, h, w), 111) * 2.0
    g_in, c = _rand((n, 64, h, w), 112), _rand((n, 64, h, w), 113) * 2.0
    wt = _rand((64, 64, 3, 3), 114, (1.0 / 576) ** 0.5)
    b = _rand((64,), 115, 0.1)
    sc, sh = _rand((64,), 116) * 0.5 + 1.0, _rand((64,), 117) * 0.3
    qa, qb, qd = _rand((64,), 118) * 0.3 + 1.0, _rand((64,), 119) * 0.2, _rand((64,), 120) * 0.1
    ks, kt = _rand((64,), 121) * 0.5 + 1.0, _rand((64,), 122) * 0.3
    slope = torch.tensor([0.2])
    xin = x
    if xpro == 'act':
        xin = F.leaky_relu(x, 0.2)
    elif xpro == 'affine_act':
        xin = F.leaky_relu(x * bc(sc) + bc(sh), 0.2)
    gg = g_in
    if gpro == 'bnact_bwd':
        gg = torch.where(bc(ks) * c + bc(kt) > 0, g_in, 0.2 * g_in)
    dy_ref = bc(qa) * gg + bc(qb) * c + bc(qd)
    wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    (F.conv2d(xin.double(), wr.double(), br.double(), padding=1) * dy_ref.double()).sum().backward()
    E.set_precision(precision)
    try:
        ref = FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(64, 64, 3, 1, 1))
        p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
        assert not p.kinds[2]
        xd = nhwc(x).cuda()
        if xpro == 'none':
            x_op = E.Operand.plain(xd)
        elif xpro == 'act':
            x_op = E.Operand.act(xd, slope.cuda())
        else:
            x_op = E.Operand.affine_act(xd, sc.cuda(), sh.cuda(), slope.cuda())
        gd, cd = nhwc(g_in).cuda(), nhwc(c).cuda()
        kw = dict(pa=qa.cuda(), pb=qb.cuda(), pd=qd.cuda())
        if gpro == 'bnact_bwd':
            kw.update(ps=ks.cuda(), pt=kt.cuda(), slope=slope.cuda())
        dy_op = E.Operand(gd, tuple(cd.shape), pro=L.PRO_BNACT_BWD if gpro == 'bnact_bwd' else L.PRO_BNBWD, x2=cd, **kw)
        grads = {}
        for sw in ('1', '0'):
            monkeypatch.setenv('SISR_TRUNK_WGRAD', sw)
            red = E.conv_wgrad(p, x_op, dy_op)
            wg = E.WeightGradBatch()
            wg.add(p, red)
            grads[sw] = wg.run()[id(ref)] + (red,)
        for k in (0, 1):
            assert maxrel(grads['1'][k], grads['0'][k]) < 2 * SPLIT_TOL[precision]   # fp32: same products, different summation order
        if precision == 'bf16x3':
            assert not torch.equal(grads['1'][0], grads['0'][0])                      # (the split contraction did run)
        assert maxrel(grads['1'][0], wr.grad) < 1e-4, 'wgrad'
        assert maxrel(grads['1'][1], br.grad) < 1e-4, 'bias grad'
        monkeypatch.setenv('SISR_TRUNK_WGRAD', '1')
        red2 = E.conv_wgrad(p, x_op, dy_op)
        wg = E.WeightGradBatch()
        wg.add(p, red2)
        gw2, gb2 = wg.run()[id(ref)]
        assert torch.equal(gw2, grads['1'][0]) and torch.equal(gb2, grads['1'][1])
    finally:
        E.set_precision('fp32')


# the split build's contraction (SisrConvDesc.mfma_split): every fp32 operand as hi + lo bf16, good to 2^-17 = 7.6e-6 relative
SPLIT_TOL = {'fp32': 1e-5, 'bf16x3': 4e-5}


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('pro', ['none', 'act', 'affine_act'])
@pytest.mark.parametrize('shape', TRUNK_SHAPES)
def test_trunk_kernel_fp32_forward_role(E, L, shape, pro, precision, monkeypatch):
    """conv_trunk_f32.hip (fp32 tensors, weights of one cout half resident in LDS, producer / consumer waves; 'fp32': exact fp32
    MFMA, 'bf16x3': the bf16 MFMA over hi / lo pairs of the fp32 operands), forward role, against the generic fp32 kernel on
    the same operands and against F.conv2d in double: output and the BatchNorm statistics merged from its per-stream partial rows"""
    tol = SPLIT_TOL[precision]
    n, h, w = _walk(shape, monkeypatch)
    x = _rand((n, 64, h, w), 131) * 2.0
    wt = _rand((64, 64, 3, 3), 132, (1.0 / 576) ** 0.5 * 1.7)
    b = _rand((64,), 133, 0.1)
    sc, sh = _rand((64,), 134) * 0.5 + 1.0, _rand((64,), 135) * 0.3
    slope = torch.tensor([0.2])
    xin = x
    if pro == 'act':
        xin = F.leaky_relu(x, 0.2)
    elif pro == 'affine_act':
        xin = F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.2)
    y_ref = F.conv2d(xin.double(), wt.double(), b.double(), padding=1)
    E.set_precision(precision)
    try:
        ref = FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(64, 64, 3, 1, 1))
        p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
        assert not p.kinds[0]
        xd = nhwc(x).cuda()
        if pro == 'none':
            op = E.Operand.plain(xd)
        elif pro == 'act':
            op = E.Operand.act(xd, slope.cuda())
        else:
            op = E.Operand.affine_act(xd, sc.cuda(), sh.cuda(), slope.cuda())
        res = {}
        for sw in ('1', '0'):
            monkeypatch.setenv('SISR_TRUNK_F32CONV', sw)
            y, sp, cp = E.conv_forward(p, op, bias=ref.bias, stats=True)
            res[sw] = (y, sp, cp)
        assert res['1'][1].shape[0] <= min(res['0'][1].shape[0], 128)            # one row per pair of workgroups, not per tile
        assert maxrel(nchw(res['1'][0]), y_ref) < tol
        assert maxrel(res['1'][0], res['0'][0]) < tol
        if precision == 'bf16x3':
            assert not torch.equal(res['1'][0], res['0'][0])                      # (the split contraction did run)
        t1, m1, v1 = _merged_stats(res['1'][1], res['1'][2])
        t0, m0, v0 = _merged_stats(res['0'][1], res['0'][2])
        assert t1 == t0 == n * h * w
        assert maxrel(m1, m0) < tol and maxrel(v1, v0) < tol
        assert maxrel(m1, y_ref.mean(dim=(0, 2, 3))) < tol and maxrel(v1, y_ref.var(dim=(0, 2, 3), unbiased=False)) < tol
        monkeypatch.setenv('SISR_TRUNK_F32CONV', '1')
        y2, _, _ = E.conv_forward(p, op, bias=ref.bias, stats=True)
        assert torch.equal(y2, res['1'][0])
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('pro,res', [('bnbwd', False), ('bnbwd', True), ('bnact_bwd', True), ('bnact_bwd', False)])
@pytest.mark.parametrize('shape', TRUNK_SHAPES)
def test_trunk_kernel_fp32_data_gradient_role(E, L, shape, pro, res, precision, monkeypatch):
    """conv_trunk_f32.hip, data-gradient role: two-tensor BatchNorm-backward prologue (with / without the activation),
    skip gradient added in the epilogue -- against the generic fp32 kernel and against conv_transpose2d in double"""
    n, h, w = _walk(shape, monkeypatch)
    g_in, c = _rand((n, 64, h, w), 141), _rand((n, 64, h, w), 142) * 2.0
    wt = _rand((64, 64, 3, 3), 143, (1.0 / 576) ** 0.5 * 1.7)
    qa, qb, qd = _rand((64,), 144) * 0.3 + 1.0, _rand((64,), 145) * 0.2, _rand((64,), 146) * 0.1
    ks, kt = _rand((64,), 147) * 0.5 + 1.0, _rand((64,), 148) * 0.3
    slope = torch.tensor([0.2])
    skip = _rand((n, 64, h, w), 149)
    bc = lambda v: v[None, :, None, None]
    gg = g_in
    if pro == 'bnact_bwd':
        gg = torch.where(bc(ks) * c + bc(kt) > 0, g_in, 0.2 * g_in)
    dy_ref = bc(qa) * gg + bc(qb) * c + bc(qd)
    out_ref = F.conv_transpose2d(dy_ref.double(), wt.double(), padding=1) + (skip.double() if res else 0.0)
    E.set_precision(precision)
    try:
        ref = FakeConv(wt.cuda(), None, E.ConvGeom(64, 64, 3, 1, 1))
        p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
        gd, cd = nhwc(g_in).cuda(), nhwc(c).cuda()
        kw = dict(pa=qa.cuda(), pb=qb.cuda(), pd=qd.cuda())
        if pro == 'bnact_bwd':
            kw.update(ps=ks.cuda(), pt=kt.cuda(), slope=slope.cuda())
        op = E.Operand(gd, tuple(cd.shape), pro=L.PRO_BNACT_BWD if pro == 'bnact_bwd' else L.PRO_BNBWD, x2=cd, **kw)
        rd = nhwc(skip).cuda() if res else None
        out = {}
        for sw in ('1', '0'):
            monkeypatch.setenv('SISR_TRUNK_F32CONV', sw)
            out[sw] = E.conv_dgrad(p, op, res=rd)
        assert maxrel(nchw(out['1']), out_ref) < SPLIT_TOL[precision]
        assert maxrel(out['1'], out['0']) < SPLIT_TOL[precision]
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('pro,res,act', [('bnbwd', True, False), ('bnact_bwd', True, True), ('bnbwd', False, True)])
@pytest.mark.parametrize('shape', TRUNK_SHAPES_SHORT)
def test_trunk_kernel_fp32_fused_bn_backward_reductions(E, L, shape, pro, res, act, monkeypatch):
    """conv_trunk_f32.hip, data-gradient role with bnb: the epilogue's per-workgroup rows of the next BatchNorm's backward
    reductions, finalized, must equal the stand-alone reduction over the same gradient (fp32: 1e-5)"""
    n, h, w = _walk(shape, monkeypatch)
    g_in, c = _rand((n, 64, h, w), 151), _rand((n, 64, h, w), 152) * 2.0
    wt = _rand((64, 64, 3, 3), 153, (1.0 / 576) ** 0.5 * 1.7)
    qa, qb, qd = _rand((64,), 154) * 0.3 + 1.0, _rand((64,), 155) * 0.2, _rand((64,), 156) * 0.1
    ks, kt = _rand((64,), 157) * 0.5 + 1.0, _rand((64,), 158) * 0.3
    slope = torch.tensor([0.2])
    skip = _rand((n, 64, h, w), 159)
    xb = _rand((n, 64, h, w), 160) * 2.0
    gamma, beta = _rand((64,), 161) + 1.5, _rand((64,), 162)
    mean = xb.mean(dim=(0, 2, 3))
    invstd = torch.rsqrt(xb.var(dim=(0, 2, 3), unbiased=False) + 1e-5)
    kb = torch.stack([gamma * invstd, beta - mean * gamma * invstd, mean, invstd]).cuda()
    E.set_precision('fp32')
    ref = FakeConv(wt.cuda(), None, E.ConvGeom(64, 64, 3, 1, 1))
    p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
    assert E.can_fuse_bn_backward(p)
    gd, cd = nhwc(g_in).cuda(), nhwc(c).cuda()
    kw = dict(pa=qa.cuda(), pb=qb.cuda(), pd=qd.cuda())
    if pro == 'bnact_bwd':
        kw.update(ps=ks.cuda(), pt=kt.cuda(), slope=slope.cuda())
    op = E.Operand(gd, tuple(cd.shape), pro=L.PRO_BNACT_BWD if pro == 'bnact_bwd' else L.PRO_BNBWD, x2=cd, **kw)
    rd = nhwc(skip).cuda() if res else None
    xd = nhwc(xb).cuda()
    b_slope = slope.cuda() if act else None
    g, part = E.conv_dgrad(p, op, res=rd, bnb=(xd, kb, b_slope))
    assert part is not None and part.shape[0] <= 256
    plain_g = E.conv_dgrad(p, op, res=rd)
    assert torch.equal(g, plain_g)                                       # the fusion does not touch the gradient itself
    fused = E.bn_backward(g, xd, kb, gamma.cuda(), slope=b_slope, part=part)
    plain = E.bn_backward(g, xd, kb, gamma.cuda(), slope=b_slope)
    for a_, b_ in zip(fused, plain):
        if a_ is not None:
            assert maxrel(a_, b_) < 2e-5
    monkeypatch.setenv('SISR_TRUNK_F32CONV', '0')                        # generic kernel: no fusion, no rows
    assert not E.can_fuse_bn_backward(p)


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
@pytest.mark.parametrize('res_slope', [None, 0.25])
@pytest.mark.parametrize('shape', TRUNK_SHAPES_SHORT)
def test_trunk_kernels_form_the_skip_sum_in_their_staging(E, L, shape, res_slope, precision, monkeypatch):
    """SISR_PRO_RES_AFFINE (both persistent forward kernels): conv(lrelu(res) + (scale * t + shift)) with the sum stored
    once as a side effect -- against the separate elementwise pass followed by the plain conv: the materialised sum must be
    bit-identical (same fp32 expression, same rounding), the conv output equal up to summation order, statistics equal"""
    n, h, w = _walk(shape, monkeypatch)
    if precision == 'bf16':
        monkeypatch.setenv('SISR_STORAGE', 'bf16')
    E.set_precision(precision)
    try:
        dt = E.act_dtype(64)
        resid = nhwc(_rand((n, 64, h, w), 171) * 2.0).cuda().to(dt)
        t = nhwc(_rand((n, 64, h, w), 172) * 2.0).cuda().to(dt)
        sc, sh = (_rand((64,), 173) * 0.5 + 1.0).cuda(), (_rand((64,), 174) * 0.3).cuda()
        wt = _rand((64, 64, 3, 3), 175, (1.0 / 576) ** 0.5 * 1.7)
        b = _rand((64,), 176, 0.1)
        slope = None if res_slope is None else torch.tensor([res_slope], device='cuda')
        ref = FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(64, 64, 3, 1, 1))
        p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
        assert E.trunk_takes_skip_sum(p, resid, t)
        out = torch.empty_like(resid)
        y1, sp1, cp1 = E.conv_forward(p, E.Operand.res_affine(resid, slope, t, sc, sh, out), bias=ref.bias, stats=True)
        summed = E.eltwise_res_affine(resid, slope, t, sc, sh)
        y0, sp0, cp0 = E.conv_forward(p, E.Operand.plain(summed), bias=ref.bias, stats=True)
        assert torch.equal(out, summed)
        assert torch.equal(y1, y0)                                  # same kernel, same staged values
        assert torch.equal(sp1, sp0) and torch.equal(cp1, cp0)
        monkeypatch.setenv('SISR_FUSE_SKIP', '0')
        assert not E.trunk_takes_skip_sum(p, resid, t)
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('pro', ['none', 'act'])
@pytest.mark.parametrize('shape', TRUNK_SHAPES_SHORT)
def test_trunk_kernel_upscale_conv_with_pixel_shuffle_store(E, L, shape, pro, monkeypatch):
    """conv_trunk.hip forward role with Cout = 256 stored through PixelShuffle(2) (the generator's upscale conv,
    model_generator.py:43-48: four cout groups = the four shuffle phases) against the generic bf16 kernel and against
    F.pixel_shuffle(F.conv2d(...)) -- bias in original channel order included"""
    n, h, w = _walk(shape, monkeypatch)
    monkeypatch.setenv('SISR_STORAGE', 'bf16')
    x = (_rand((n, 64, h, w), 181) * 2.0).bfloat16().float()
    wt = _rand((256, 64, 3, 3), 182, (1.0 / 576) ** 0.5 * 1.7)
    b = _rand((256,), 183, 0.1)
    slope = torch.tensor([0.25])
    xin = F.leaky_relu(x, 0.25) if pro == 'act' else x
    y_ref = F.pixel_shuffle(F.conv2d(xin, wt, b, padding=1), 2)
    E.set_precision('bf16')
    try:
        ref = FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(64, 256, 3, 1, 1, shuffle2=True))
        p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
        xd = nhwc(x).cuda().bfloat16()
        op = E.Operand.plain(xd) if pro == 'none' else E.Operand.act(xd, slope.cuda())
        out = {}
        for sw in ('1', '0'):
            monkeypatch.setenv('SISR_TRUNK', sw)
            out[sw] = E.conv_forward(p, op, bias=ref.bias)[0].float()
        assert tuple(out['1'].shape) == (n, 2 * h, 2 * w, 64)
        assert maxrel(nchw(out['1']), y_ref) < BF16_TOL
        assert maxrel(out['1'], out['0']) < 6e-3
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('shape', TRUNK_SHAPES_SHORT)
def test_trunk_kernel_upscale_conv_weight_gradient(E, L, shape, monkeypatch):
    """wgrad_trunk.hip with Cout = 256 and the gradient stored shuffled (the upscale conv: four cout groups = the four
    PixelShuffle phases, activation-backward prologue on the strided view of each phase) against the generic bf16 kernel
    and against autograd through conv -> pixel_shuffle -> PReLU"""
    n, h, w = _walk(shape, monkeypatch)
    monkeypatch.setenv('SISR_STORAGE', 'bf16')
    bf = lambda t: t.bfloat16().float()
    x = bf(_rand((n, 64, h, w), 191) * 2.0)
    wt = _rand((256, 64, 3, 3), 192, (1.0 / 576) ** 0.5 * 1.7)
    b = _rand((256,), 193, 0.1)
    slope = torch.tensor([0.25])
    wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    pre_ref = F.pixel_shuffle(F.conv2d(x, wr, br, padding=1), 2)
    pre = bf(pre_ref.detach())                                   # the stored pre-activation (bf16 NHWC in the engine)
    g = bf(_rand((n, 64, 2 * h, 2 * w), 194))                    # gradient arriving at the PReLU output
    gpre = torch.where(pre > 0, g, 0.25 * g)
    pre_ref.backward(gpre)
    E.set_precision('bf16')
    try:
        ref = FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(64, 256, 3, 1, 1, shuffle2=True))
        p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
        x_op = E.Operand.plain(nhwc(x).cuda().bfloat16())
        gd, pd_ = nhwc(g).cuda().bfloat16(), nhwc(pre).cuda().bfloat16()
        dy_op = E.Operand(gd, (n, h, w, 256), pro=L.PRO_ACT_BWD, mode=L.X_UNSHUFFLE2, x2=pd_, slope=slope.cuda())
        red = {}
        for sw in ('1', '0'):
            monkeypatch.setenv('SISR_TRUNK_UP', sw)
            red[sw] = E.conv_wgrad(p, x_op, dy_op)
        assert maxrel(red['1'], red['0']) < 5e-3     # (the persistent kernel's per-workgroup partial sums are stored as bf16: 2^-9 each)
        wg = E.WeightGradBatch()
        wg.add(p, red['1'])
        gw, gb = wg.run()[id(ref)]
        assert maxrel(gw, wr.grad) < BF16_TOL and maxrel(gb, br.grad) < BF16_TOL
        monkeypatch.setenv('SISR_TRUNK_UP', '1')
        assert torch.equal(E.conv_wgrad(p, x_op, dy_op), red['1'])
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('pro', ['none', 'act', 'affine_act'])
@pytest.mark.parametrize('shape', TRUNK_SHAPES_SHORT)
def test_trunk_kernel_fp32_upscale_conv_with_pixel_shuffle_store(E, L, shape, pro, precision, monkeypatch):
    """conv_trunk_f32.hip forward role with Cout = 256 stored through PixelShuffle(2) (the generator's upscale conv,
    model_generator.py:43-48: eight blocks of 32 packed couts per pixel-tile stream) with fp32 tensors, exact and split
    contraction, against the generic fp32 kernel and against F.pixel_shuffle(F.conv2d(...)) in double -- bias in original
    channel order included"""
    n, h, w = _walk(shape, monkeypatch)
    x = _rand((n, 64, h, w), 281) * 2.0
    wt = _rand((256, 64, 3, 3), 282, (1.0 / 576) ** 0.5 * 1.7)
    b = _rand((256,), 283, 0.1)
    sc, sh = _rand((64,), 284) * 0.5 + 1.0, _rand((64,), 285) * 0.3
    slope = torch.tensor([0.25])
    xin = x
    if pro == 'act':
        xin = F.leaky_relu(x, 0.25)
    elif pro == 'affine_act':
        xin = F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.25)
    y_ref = F.pixel_shuffle(F.conv2d(xin.double(), wt.double(), b.double(), padding=1), 2)
    E.set_precision(precision)
    try:
        ref = FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(64, 256, 3, 1, 1, shuffle2=True))
        p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
        assert not p.kinds[0]
        xd = nhwc(x).cuda()
        op = (E.Operand.plain(xd) if pro == 'none' else E.Operand.act(xd, slope.cuda()) if pro == 'act'
              else E.Operand.affine_act(xd, sc.cuda(), sh.cuda(), slope.cuda()))
        out = {}
        for sw in ('1', '0'):
            monkeypatch.setenv('SISR_TRUNK_UP', sw)
            out[sw] = E.conv_forward(p, op, bias=ref.bias)[0]
        assert tuple(out['1'].shape) == (n, 2 * h, 2 * w, 64)
        assert maxrel(nchw(out['1']), y_ref) < SPLIT_TOL[precision]
        assert maxrel(out['1'], out['0']) < SPLIT_TOL[precision]
        assert not torch.equal(out['1'], out['0'])                     # (two kernels: different summation order at least)
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('shape', TRUNK_SHAPES_SHORT)
def test_trunk_kernel_fp32_upscale_conv_weight_gradient(E, L, shape, precision, monkeypatch):
    """wgrad_trunk_f32.hip with Cout = 256 and the gradient stored shuffled (the upscale conv: four cout groups = the four
    PixelShuffle phases, activation-backward prologue on the strided view of each phase), fp32 tensors, exact and split
    contraction, against the generic fp32 kernel and against autograd through conv -> pixel_shuffle -> PReLU"""
    n, h, w = _walk(shape, monkeypatch)
    x = _rand((n, 64, h, w), 291) * 2.0
    wt = _rand((256, 64, 3, 3), 292, (1.0 / 576) ** 0.5 * 1.7)
    b = _rand((256,), 293, 0.1)
    slope = torch.tensor([0.25])
    wr, br = wt.clone().double().requires_grad_(True), b.clone().double().requires_grad_(True)
    pre_ref = F.pixel_shuffle(F.conv2d(x.double(), wr, br, padding=1), 2)
    pre = pre_ref.detach().float()                               # the stored pre-activation (fp32 NHWC in the engine)
    g = _rand((n, 64, 2 * h, 2 * w), 294)                        # gradient arriving at the PReLU output
    gpre = torch.where(pre > 0, g, 0.25 * g)
    pre_ref.backward(gpre.double())
    E.set_precision(precision)
    try:
        ref = FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(64, 256, 3, 1, 1, shuffle2=True))
        p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
        assert not p.kinds[2]
        x_op = E.Operand.plain(nhwc(x).cuda())
        gd, pd_ = nhwc(g).cuda(), nhwc(pre).cuda()
        dy_op = E.Operand(gd, (n, h, w, 256), pro=L.PRO_ACT_BWD, mode=L.X_UNSHUFFLE2, x2=pd_, slope=slope.cuda())
        red = {}
        for sw in ('1', '0'):
            monkeypatch.setenv('SISR_TRUNK_UP', sw)
            red[sw] = E.conv_wgrad(p, x_op, dy_op)
        def unpacked(r):
            wg = E.WeightGradBatch()
            wg.add(p, r)
            return wg.run()[id(ref)]
        (gw, gb), (gw0, gb0) = unpacked(red['1']), unpacked(red['0'])      # (the packed rows have padding the trunk kernel never writes)
        assert maxrel(gw, gw0) < 2 * SPLIT_TOL[precision] and maxrel(gb, gb0) < 2 * SPLIT_TOL[precision]
        assert not torch.equal(gw, gw0)
        assert maxrel(gw, wr.grad) < 1e-4 and maxrel(gb, br.grad) < 1e-4
        monkeypatch.setenv('SISR_TRUNK_UP', '1')
        gw2, gb2 = unpacked(E.conv_wgrad(p, x_op, dy_op))
        assert torch.equal(gw2, gw) and torch.equal(gb2, gb)
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('res', [False, True])
@pytest.mark.parametrize('shape', TRUNK_SHAPES_SHORT)
def test_trunk_kernel_fp32_upscale_conv_data_gradient(E, L, shape, res, precision, monkeypatch):
    """conv_trunk_f32.hip, data gradient of the upscale conv (256 -> 64 over the un-shuffling view of the [N][2H][2W][64]
    gradient, activation-backward prologue): four launches of the data-gradient role, one per PixelShuffle phase, each adding
    onto the one before -- against the generic fp32 kernel and against autograd through conv -> pixel_shuffle -> PReLU"""
    n, h, w = _walk(shape, monkeypatch)
    x = (_rand((n, 64, h, w), 301) * 2.0).double().requires_grad_(True)
    wt = _rand((256, 64, 3, 3), 302, (1.0 / 576) ** 0.5 * 1.7)
    skip = _rand((n, 64, h, w), 303)
    slope = torch.tensor([0.25])
    pre_ref = F.pixel_shuffle(F.conv2d(x, wt.double(), None, padding=1), 2)
    pre = pre_ref.detach().float()
    g = _rand((n, 64, 2 * h, 2 * w), 304)
    pre_ref.backward(torch.where(pre > 0, g, 0.25 * g).double())
    want = x.grad + (skip.double() if res else 0.0)
    E.set_precision(precision)
    try:
        ref = FakeConv(wt.cuda(), None, E.ConvGeom(64, 256, 3, 1, 1, shuffle2=True))
        p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
        gd, pd_ = nhwc(g).cuda(), nhwc(pre).cuda()
        dy_op = E.Operand(gd, (n, h, w, 256), pro=L.PRO_ACT_BWD, mode=L.X_UNSHUFFLE2, x2=pd_, slope=slope.cuda())
        rd = nhwc(skip).cuda() if res else None
        out = {}
        for sw in ('1', '0'):
            monkeypatch.setenv('SISR_TRUNK_UP', sw)
            out[sw] = E.conv_dgrad(p, dy_op, res=rd)
        assert tuple(out['1'].shape) == (n, h, w, 64)
        assert maxrel(nchw(out['1']), want) < 2 * SPLIT_TOL[precision]
        assert maxrel(out['1'], out['0']) < 2 * SPLIT_TOL[precision]
        assert not torch.equal(out['1'], out['0'])
        if res:
            assert torch.equal(rd, nhwc(skip).cuda())                      # the residual itself is read, never written
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('role', ['first_conv', 'end_dgrad'])
@pytest.mark.parametrize('shape', [(2, 16, 32), (3, 48, 48), (1, 96, 96), (3, 48, 48, 4), (16, 96, 96), (4, 192, 192)])
def test_thin_kernel_over_a_3_channel_image(E, L, shape, role, monkeypatch):
    """conv_thin.hip -- the generator's first conv (model_generator.py:32: 9x9, NCHW fp32 image -> 64 bf16 NHWC channels)
    and the data gradient of its last conv (model_generator.py:52-53: 3x3 over the 3-channel image gradient with tanh'
    as prologue and the flipped weights) -- against the generic kernel the same descriptor runs on with SISR_THIN=0 and
    against F.conv2d / autograd on the bf16-rounded operands"""
    n, h, w = _walk(shape, monkeypatch)
    monkeypatch.setenv('SISR_STORAGE', 'bf16')
    bf = lambda t: t.bfloat16().float()
    E.set_precision('bf16')
    try:
        out = {}
        if role == 'first_conv':
            x = _rand((n, 3, h, w), 201)
            wt = _rand((64, 3, 9, 9), 202, (1.0 / 243) ** 0.5 * 1.7)
            b = _rand((64,), 203, 0.1)
            y_ref = F.conv2d(bf(x), bf(wt), b, padding=4)
            ref = FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(3, 64, 9, 1, 4))
            p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
            op = E.Operand.plain(x.cuda(), dims=(n, h, w, 3), mode=L.X_NCHW)
            for sw in ('1', '0'):
                monkeypatch.setenv('SISR_THIN', sw)
                out[sw] = E.conv_forward(p, op, bias=ref.bias)[0]
            assert out['1'].dtype == torch.bfloat16 and tuple(out['1'].shape) == (n, h, w, 64)
            assert maxrel(nchw(out['1'].float()), y_ref) < 8e-3       # (bf16 rounding of the stored result)
        else:
            xin = _rand((n, 64, h, w), 211).requires_grad_(True)
            wt = _rand((3, 64, 3, 3), 212, (1.0 / 576) ** 0.5 * 1.7)
            yt = torch.tanh(F.conv2d(xin, bf(wt), None, padding=1))
            g = _rand((n, 3, h, w), 213)
            gpre = bf(g * (1 - yt.detach() ** 2))                     # what the kernel's staging hands the matrix cores
            gx_ref, = torch.autograd.grad(F.conv2d(xin, bf(wt), None, padding=1), xin, gpre)
            ref = FakeConv(wt.cuda(), None, E.ConvGeom(64, 3, 3, 1, 1))
            p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
            dy = E.Operand(g.cuda(), (n, h, w, 3), pro=L.PRO_TANH_BWD, mode=L.X_NCHW, x2=yt.detach().cuda())
            for sw in ('1', '0'):
                monkeypatch.setenv('SISR_THIN', sw)
                out[sw] = E.conv_dgrad(p, dy)
            assert out['1'].dtype == torch.bfloat16 and tuple(out['1'].shape) == (n, h, w, 64)
            assert maxrel(nchw(out['1'].float()), gx_ref) < 8e-3
        # same descriptor on the generic kernel (bf16 multiplicands there as well; fp32-accumulation order differs)
        assert maxrel(out['1'].float(), out['0'].float()) < 1.2e-2
        monkeypatch.setenv('SISR_THIN', '1')
        again = E.conv_forward(p, op, bias=ref.bias)[0] if role == 'first_conv' else E.conv_dgrad(p, dy)
        assert torch.equal(again, out['1'])
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('act', [True, False])
@pytest.mark.parametrize('shape', [(2, 16, 32), (3, 24, 64), (1, 96, 96), (3, 24, 64, 3), (16, 96, 96)])
def test_thin_kernel_first_conv_weight_gradient(E, L, shape, act, monkeypatch):
    """wgrad_thin.hip -- weight / bias gradient of the generator's first conv (model_generator.py:32-33: 9x9 over the
    NCHW fp32 image; the gradient arrives through the PReLU, activation-backward prologue) -- against the generic
    exact-fp32 kernel the same descriptor runs on with SISR_THIN=0 and against autograd"""
    n, h, w = _walk(shape, monkeypatch)
    monkeypatch.setenv('SISR_STORAGE', 'bf16')
    bf = lambda t: t.bfloat16().float()
    x = _rand((n, 3, h, w), 221)
    wt = _rand((64, 3, 9, 9), 222, (1.0 / 243) ** 0.5 * 1.7)
    b = _rand((64,), 223, 0.1)
    wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    pre_ref = F.conv2d(bf(x), wr, br, padding=4)               # (the kernel multiplies bf16-rounded image values)
    pre = bf(pre_ref.detach())                                  # the stored pre-activation (bf16 NHWC in the engine)
    g = bf(_rand((n, 64, h, w), 224))
    gpre = bf(torch.where(pre > 0, g, 0.25 * g)) if act else g
    pre_ref.backward(gpre)
    E.set_precision('bf16')
    try:
        ref = FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(3, 64, 9, 1, 4))
        p = E.prepare_weights([(ref, n, h, w)], training=True, need_dgrad=False)[0][0]
        x_op = E.Operand.plain(x.cuda(), dims=(n, h, w, 3), mode=L.X_NCHW)
        gd, pd_ = nhwc(g).cuda().bfloat16(), nhwc(pre).cuda().bfloat16()
        slope = torch.tensor([0.25], device='cuda')
        dy_op = E.Operand(gd, (n, h, w, 64), pro=L.PRO_ACT_BWD, x2=pd_, slope=slope) if act else E.Operand.plain(gd)
        red = {}
        for sw in ('1', '0'):
            monkeypatch.setenv('SISR_THIN', sw)
            red[sw] = E.conv_wgrad(p, x_op, dy_op)
        assert red['1'].shape == red['0'].shape == (9 * 28 * 64 + 64,)
        # packed layout [ky][kx * 3 + ci, padded 27 -> 28][co] + bias row; the padding row is not part of the gradient
        # (this kernel writes zeros there, the generic one leaves it as allocated)
        body = lambda r: torch.cat([r[:9 * 28 * 64].view(9, 28, 64)[:, :27].reshape(-1), r[9 * 28 * 64:]])
        assert maxrel(body(red['1']), body(red['0'])) < 8e-3     # (fp32 image values / unrounded act' there)
        assert float(red['1'][:9 * 28 * 64].view(9, 28, 64)[:, 27].abs().max()) == 0.0
        wg = E.WeightGradBatch()
        wg.add(p, red['1'])
        gw, gb = wg.run()[id(ref)]
        assert maxrel(gw, wr.grad) < 2e-3 and maxrel(gb, br.grad) < 2e-3
        monkeypatch.setenv('SISR_THIN', '1')
        assert torch.equal(E.conv_wgrad(p, x_op, dy_op), red['1'])
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('act', [True, False])
@pytest.mark.parametrize('shape', [(2, 16, 32), (3, 24, 64), (3, 24, 64, 3), (16, 96, 96), (2, 192, 192)])
def test_thin_kernel_discriminator_first_conv_weight_gradient(E, L, shape, act, monkeypatch):
    """wgrad_thin.hip, KS = 3 -- weight / bias gradient of the discriminator's first conv (model_discriminator.py:36: 3x3, 3 -> 64 over
    the NCHW fp32 image; the gradient arrives through the LeakyReLU) -- against the generic exact-fp32 kernel the same descriptor runs
    on with SISR_THIN3=0 and against autograd"""
    n, h, w = _walk(shape, monkeypatch)
    monkeypatch.setenv('SISR_STORAGE', 'bf16')
    bf = lambda t: t.bfloat16().float()
    x = _rand((n, 3, h, w), 321)
    wt = _rand((64, 3, 3, 3), 322, (1.0 / 27) ** 0.5 * 1.7)
    b = _rand((64,), 323, 0.1)
    wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    pre_ref = F.conv2d(bf(x), wr, br, padding=1)               # (the kernel multiplies bf16-rounded image values)
    pre = bf(pre_ref.detach())
    g = bf(_rand((n, 64, h, w), 324))
    gpre = bf(torch.where(pre > 0, g, 0.2 * g)) if act else g
    pre_ref.backward(gpre)
    E.set_precision('bf16')
    try:
        ref = FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(3, 64, 3, 1, 1))
        p = E.prepare_weights([(ref, n, h, w)], training=True, need_dgrad=False)[0][0]
        x_op = E.Operand.plain(x.cuda(), dims=(n, h, w, 3), mode=L.X_NCHW)
        gd, pd_ = nhwc(g).cuda().bfloat16(), nhwc(pre).cuda().bfloat16()
        dy_op = E.Operand(gd, (n, h, w, 64), pro=L.PRO_ACT_BWD, x2=pd_, slope=0.2) if act else E.Operand.plain(gd)
        red = {}
        for sw in ('1', '0'):
            monkeypatch.setenv('SISR_THIN3', sw)
            gdesc = E._copy_struct(p.plans[2])
            x_op.fill(gdesc)
            dy_op.fill(gdesc, g=True)
            assert bool(L.lib().sisr_wgrad_thin_eligible(C.byref(gdesc))) == (sw == '1')
            red[sw] = E.conv_wgrad(p, x_op, dy_op)
        assert red['1'].shape == red['0'].shape == (3 * 12 * 64 + 64,)
        # packed layout [ky][kx * 3 + ci, padded 9 -> 12][co] + bias row; this kernel writes zeros into the padding rows
        body = lambda r: torch.cat([r[:3 * 12 * 64].view(3, 12, 64)[:, :9].reshape(-1), r[3 * 12 * 64:]])
        assert maxrel(body(red['1']), body(red['0'])) < 8e-3     # (fp32 image values / unrounded act' there)
        assert float(red['1'][:3 * 12 * 64].view(3, 12, 64)[:, 9:].abs().max()) == 0.0
        wg = E.WeightGradBatch()
        wg.add(p, red['1'])
        gw, gb = wg.run()[id(ref)]
        assert maxrel(gw, wr.grad) < 2e-3 and maxrel(gb, br.grad) < 2e-3
        monkeypatch.setenv('SISR_THIN3', '1')
        assert torch.equal(E.conv_wgrad(p, x_op, dy_op), red['1'])
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('pro,tanh', [('act', True), ('none', True), ('act', False)])
@pytest.mark.parametrize('shape', [(2, 16, 32), (3, 13, 31), (1, 96, 96), (2, 192, 192), (16, 192, 192)])
def test_last_conv_as_gemm_plus_col2im(E, L, shape, pro, tanh, monkeypatch):
    """conv_toimage.hip -- the generator's last conv (model_generator.py:52-53: 3x3, 64 -> 3, + Tanh; bf16 NHWC
    activations in, NCHW fp32 image out; PReLU of the upscale stage as prologue) -- against the generic bf16 kernel the
    same descriptor runs on with SISR_THIN=0 and against F.conv2d on the bf16-rounded operands; ragged sizes included"""
    n, h, w = _walk(shape, monkeypatch)
    monkeypatch.setenv('SISR_STORAGE', 'bf16')
    bf = lambda t: t.bfloat16().float()
    x = bf(_rand((n, 64, h, w), 231) * 2.0)
    wt = _rand((3, 64, 3, 3), 232, (1.0 / 576) ** 0.5 * 1.7)
    b = _rand((3,), 233, 0.1)
    xin = bf(F.leaky_relu(x, 0.25)) if pro == 'act' else x
    y_ref = F.conv2d(xin, bf(wt), b, padding=1)
    y_ref = torch.tanh(y_ref) if tanh else y_ref
    E.set_precision('bf16')
    try:
        ref = FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(64, 3, 3, 1, 1))
        p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
        xd = nhwc(x).cuda().bfloat16()
        op = E.Operand.plain(xd) if pro == 'none' else E.Operand.act(xd, torch.tensor([0.25], device='cuda'))
        out = {}
        for sw in ('1', '0'):
            monkeypatch.setenv('SISR_THIN', sw)
            out[sw] = E.conv_forward(p, op, bias=ref.bias, y_mode=L.Y_NCHW, epi=L.EPI_TANH if tanh else L.EPI_NONE)[0]
        assert out['1'].dtype == torch.float32 and tuple(out['1'].shape) == (n, 3, h, w)
        assert maxrel(out['1'], y_ref) < 2e-3
        assert maxrel(out['1'], out['0']) < 2e-3
        monkeypatch.setenv('SISR_THIN', '1')
        again = E.conv_forward(p, op, bias=ref.bias, y_mode=L.Y_NCHW, epi=L.EPI_TANH if tanh else L.EPI_NONE)[0]
        assert torch.equal(again, out['1'])
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('pro,tanh', [('act', True), ('none', False)])
@pytest.mark.parametrize('shape', [(2, 16, 32), (3, 24, 64), (1, 96, 96), (3, 24, 64, 3), (16, 192, 192)])
def test_last_conv_weight_gradient_reads_the_image_gradient_directly(E, L, shape, pro, tanh, monkeypatch):
    """wgrad_toimage.hip -- weight / bias gradient of the generator's last conv (model_generator.py:52-53: 3x3, 64 -> 3,
    + Tanh; bf16 NHWC activations through the PReLU prologue, NCHW fp32 image gradient through tanh') -- against the
    generic bf16 kernel (which runs on a 4-channel NHWC copy of the gradient, SISR_THIN=0) and against autograd"""
    n, h, w = _walk(shape, monkeypatch)
    monkeypatch.setenv('SISR_STORAGE', 'bf16')
    bf = lambda t: t.bfloat16().float()
    x = bf(_rand((n, 64, h, w), 241) * 2.0)
    wt = _rand((3, 64, 3, 3), 242, (1.0 / 576) ** 0.5 * 1.7)
    b = _rand((3,), 243, 0.1)
    wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    xin = bf(F.leaky_relu(x, 0.25)) if pro == 'act' else x
    pre = F.conv2d(xin, wr, br, padding=1)
    y = torch.tanh(pre) if tanh else pre
    g = _rand((n, 3, h, w), 244)
    gpre = bf(g * (1 - y.detach() ** 2)) if tanh else bf(g)       # what the staging hands the matrix cores
    pre.backward(gpre)
    E.set_precision('bf16')
    try:
        ref = FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(64, 3, 3, 1, 1))
        p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
        xd = nhwc(x).cuda().bfloat16()
        x_op = E.Operand.plain(xd) if pro == 'none' else E.Operand.act(xd, torch.tensor([0.25], device='cuda'))
        if tanh:
            dy_op = E.Operand(g.cuda(), (n, h, w, 3), pro=L.PRO_TANH_BWD, mode=L.X_NCHW, x2=y.detach().cuda())
        else:
            dy_op = E.Operand.plain(g.cuda(), dims=(n, h, w, 3), mode=L.X_NCHW)
        red = {}
        for sw in ('1', '0'):
            monkeypatch.setenv('SISR_THIN', sw)
            red[sw] = E.conv_wgrad(p, x_op, dy_op)
        assert red['1'].shape == red['0'].shape
        assert maxrel(red['1'], red['0']) < 2e-3
        wg = E.WeightGradBatch()
        wg.add(p, red['1'])
        gw, gb = wg.run()[id(ref)]
        # (the bias gradient is summed in fp32 from the unrounded tanh' products, the matrix operand is their bf16 rounding)
        gb_ref = (g * (1 - y.detach() ** 2) if tanh else g).sum(dim=(0, 2, 3))
        assert maxrel(gw, wr.grad) < 2e-3 and maxrel(gb, gb_ref) < 1e-4
        monkeypatch.setenv('SISR_THIN', '1')
        assert torch.equal(E.conv_wgrad(p, x_op, dy_op), red['1'])
    finally:
        E.set_precision('fp32')


@pytest.mark.parametrize('pro,tanh', [('act', True), ('none', False)])
@pytest.mark.parametrize('shape', [(2, 16, 32), (3, 13, 31), (1, 96, 96), (16, 192, 192)])
def test_last_conv_as_gemm_plus_col2im_fp32(E, L, shape, pro, tanh, monkeypatch):
    """conv_toimage.hip's exact-fp32 variant (fp32 parity build: fp32 NHWC activations in, NCHW fp32 image out) against
    the generic fp32 kernel (SISR_THIN=0) and F.conv2d at the parity build's tolerance"""
    n, h, w = _walk(shape, monkeypatch)
    x = _rand((n, 64, h, w), 251) * 2.0
    wt = _rand((3, 64, 3, 3), 252, (1.0 / 576) ** 0.5 * 1.7)
    b = _rand((3,), 253, 0.1)
    y_ref = F.conv2d(F.leaky_relu(x, 0.25) if pro == 'act' else x, wt, b, padding=1)
    y_ref = torch.tanh(y_ref) if tanh else y_ref
    ref = FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(64, 3, 3, 1, 1))
    p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
    xd = nhwc(x).cuda()
    op = E.Operand.plain(xd) if pro == 'none' else E.Operand.act(xd, torch.tensor([0.25], device='cuda'))
    out = {}
    for sw in ('1', '0'):
        monkeypatch.setenv('SISR_THIN', sw)
        out[sw] = E.conv_forward(p, op, bias=ref.bias, y_mode=L.Y_NCHW, epi=L.EPI_TANH if tanh else L.EPI_NONE)[0]
    assert maxrel(out['1'], y_ref) < 1e-5 and maxrel(out['1'], out['0']) < 1e-5
    d = L.ConvDesc.from_buffer_copy(p.plans[0])
    d.x_mode, d.pro_mode, d.y_mode, d.epi_act = L.X_NHWC, L.PRO_ACT if pro == 'act' else L.PRO_NONE, L.Y_NCHW, 0
    monkeypatch.setenv('SISR_THIN', '1')
    assert L.lib().sisr_conv2d_toimage_f32_eligible(d) == 1      # (the kernel under test did run)


@pytest.mark.parametrize('pro,tanh', [('act', True), ('none', False)])
@pytest.mark.parametrize('shape', [(2, 16, 32), (3, 24, 64), (1, 96, 96), (3, 24, 64, 3), (16, 192, 192)])
def test_last_conv_weight_gradient_fp32(E, L, shape, pro, tanh, monkeypatch):
    """wgrad_toimage.hip's exact-fp32 variant (fp32 parity build) against the generic fp32 kernel (SISR_THIN=0) and
    autograd at the parity build's tolerance; padding entries of the packed slab are zero"""
    n, h, w = _walk(shape, monkeypatch)
    x = _rand((n, 64, h, w), 261) * 2.0
    wt = _rand((3, 64, 3, 3), 262, (1.0 / 576) ** 0.5 * 1.7)
    b = _rand((3,), 263, 0.1)
    wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    pre = F.conv2d(F.leaky_relu(x, 0.25) if pro == 'act' else x, wr, br, padding=1)
    y = torch.tanh(pre) if tanh else pre
    g = _rand((n, 3, h, w), 264)
    y.backward(g)
    ref = FakeConv(wt.cuda(), b.cuda(), E.ConvGeom(64, 3, 3, 1, 1))
    p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
    xd = nhwc(x).cuda()
    x_op = E.Operand.plain(xd) if pro == 'none' else E.Operand.act(xd, torch.tensor([0.25], device='cuda'))
    if tanh:
        dy_op = E.Operand(g.cuda(), (n, h, w, 3), pro=L.PRO_TANH_BWD, mode=L.X_NCHW, x2=y.detach().cuda())
    else:
        dy_op = E.Operand.plain(g.cuda(), dims=(n, h, w, 3), mode=L.X_NCHW)
    wgd = L.WgradDesc.from_buffer_copy(p.plans[2])
    x_op.fill(wgd)
    dy_op.fill(wgd, g=True)
    monkeypatch.setenv('SISR_THIN', '1')
    assert L.lib().sisr_wgrad_toimage_f32_eligible(wgd) == 1      # (the kernel under test does run)
    grads = {}
    for sw in ('1', '0'):
        monkeypatch.setenv('SISR_THIN', sw)
        red = E.conv_wgrad(p, x_op, dy_op)
        wg = E.WeightGradBatch()
        wg.add(p, red)
        grads[sw] = wg.run()[id(ref)]
    for sw in ('1', '0'):
        assert maxrel(grads[sw][0], wr.grad) < 2e-5 and maxrel(grads[sw][1], br.grad) < 2e-5, sw
    monkeypatch.setenv('SISR_THIN', '1')
    assert torch.equal(E.conv_wgrad(p, x_op, dy_op), E.conv_wgrad(p, x_op, dy_op))
