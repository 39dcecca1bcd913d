"""GPU: the split-K implicit-GEMM convolution family (csrc/conv_deep.hip) through the C ABI, at kernel level.

Shapes: the discriminator's eight conv layers (model_discriminator.py:10,39-44 with config.py:81-82's features / strides) and the
VGG19 layers behind the VGG22 / VGG54 taps (model_content_extractor.py:43) at B16 full size, plus small shapes that force every
special path (row bands straddling images, stride-2 halo, the parity classes of a stride-2 data gradient, K split, several cout
tiles, the fused BatchNorm-backward reductions).  Reference: torch-CPU conv2d / conv_transpose2d in DOUBLE on exactly the operands
the kernel multiplies (inputs and weights rounded to bf16 after the lazy operand's prologue), so the comparison only leaves the
fp32 accumulation order and the bf16 rounding of the stored result (2^-9 of a value): tolerance 6e-3 of the tensor's max-norm."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_helpers import FakeConv, maxrel, nchw, nhwc, pkg

pytestmark = pytest.mark.gpu
TOL = 6e-3


@pytest.fixture(scope='module')
def E():
    e = pkg('engine')
    e.set_precision('bf16')
    yield e
    e.set_precision('fp32')


@pytest.fixture(scope='module')
def L():
    return pkg('_lib')


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _bf(t):
    return t.to(torch.bfloat16).float()


def _lrelu(t, s):
    return torch.where(t > 0, t, s * t)


def _prep(E, n, cin, cout, stride, h, w, seed=0, deep_dgrad=False):
    wt = _rand((cout, cin, 3, 3), seed + 2, (1.0 / (cin * 9)) ** 0.5 * 1.7)
    b = _rand((cout,), seed + 3, 0.1)
    geom = E.ConvGeom(cin, cout, 3, stride, 1, deep_dgrad=deep_dgrad)
    ref = FakeConv(wt.cuda(), b.cuda(), geom)
    preps, keep = E.prepare_weights([(ref, n, h, w)], training=True)
    return preps[0], ref, wt, b, keep


def _merge_stats(sp, cp):
    cnt, mean_t, m2_t = cp.double().cpu(), sp[:, 0].double().cpu(), sp[:, 1].double().cpu()
    tot = cnt.sum()
    mean = (cnt[:, None] * mean_t).sum(0) / tot
    var = (m2_t + cnt[:, None] * (mean_t - mean) ** 2).sum(0) / tot
    return tot, mean, var


# n, cin, cout, stride, h, w
SMALL = [
    (3, 64, 64, 1, 12, 12),        # row bands of 10 rows straddling images, BN = 64
    (5, 32, 128, 1, 6, 6),         # 21-row bands over 3.5 images, BN = 128
    (2, 64, 128, 1, 24, 24),       # 5-row bands
    (2, 128, 64, 1, 16, 32),       # 8 x 16 tiles, two column tiles
    (2, 64, 64, 1, 37, 29),        # ragged: 4-row bands of 29 columns
    (3, 64, 128, 2, 24, 24),       # stride 2 -> 12 x 12 bands
    (2, 64, 64, 2, 32, 32),        # stride 2 -> 16 x 16: 8 x 16 tiles
    (4, 256, 512, 1, 12, 12),      # 4 cout tiles, K split
    (4, 512, 512, 2, 12, 12),      # D's last layer: 6 x 6 maps, deep K split
]


@pytest.mark.parametrize('pro', ['none', 'act', 'affine_act'])
@pytest.mark.parametrize('case', SMALL)
def test_forward_prologues_statistics(E, L, case, pro):
    n, cin, cout, stride, h, w = case
    p, ref, wt, b, keep = _prep(E, n, cin, cout, stride, h, w)
    assert p.kinds[0] == 2, 'forward role not planned for conv_deep.hip'
    x = _bf(_rand((n, cin, h, w), 1))
    sc, sh = _rand((cin,), 5) * 0.5 + 1.0, _rand((cin,), 6, 0.3)
    slope = 0.2
    xd = nhwc(x).cuda().to(torch.bfloat16)
    if pro == 'none':
        op, a = E.Operand.plain(xd), x
    elif pro == 'act':
        op, a = E.Operand.act(xd, slope), _lrelu(x, slope)
    else:
        op = E.Operand.affine_act(xd, sc.cuda(), sh.cuda(), slope)
        a = _lrelu(x * sc[None, :, None, None] + sh[None, :, None, None], slope)
    y, sp, cp = E.conv_forward(p, op, bias=ref.bias, stats=True)
    y_ref = F.conv2d(_bf(a).double(), _bf(wt).double(), b.double(), stride=stride, padding=1)
    assert y.dtype == torch.bfloat16 and tuple(y.shape) == (n, y_ref.shape[2], y_ref.shape[3], cout)
    assert maxrel(nchw(y.float()), y_ref) < TOL, 'forward'
    tot, mean, var = _merge_stats(sp, cp)
    assert int(tot) == n * y_ref.shape[2] * y_ref.shape[3]
    assert maxrel(mean, y_ref.mean(dim=(0, 2, 3))) < 1e-3 and maxrel(var, y_ref.var(dim=(0, 2, 3), unbiased=False)) < 1e-3


@pytest.mark.parametrize('pro', ['none', 'act_bwd', 'bnact_bwd'])
@pytest.mark.parametrize('case', SMALL)
def test_data_gradient_prologues_and_fused_reductions(E, L, case, pro):
    """dx = conv_transpose(dy'), dy' the lazy gradient operand; with the BatchNorm-backward reductions of the BatchNorm whose
    input has dx's shape fused into the epilogue (the discriminator's backward chain, model_discriminator.py:5-15 differentiated)"""
    n, cin, cout, stride, h, w = case
    p, ref, wt, b, keep = _prep(E, n, cin, cout, stride, h, w)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    dy = _bf(_rand((n, cout, ho, wo), 11))
    c = _bf(_rand((n, cout, ho, wo), 12))                    # the forward conv output the activation / BatchNorm saw
    slope = 0.01
    qa, qb, qd = _rand((cout,), 13) * 0.5 + 1.0, _rand((cout,), 14, 0.2), _rand((cout,), 15, 0.1)
    ks, kt = _rand((cout,), 16) * 0.5 + 1.0, _rand((cout,), 17, 0.3)
    dyd, cd = nhwc(dy).cuda().to(torch.bfloat16), nhwc(c).cuda().to(torch.bfloat16)
    bc = lambda v: v[None, :, None, None]
    if pro == 'none':
        op, g = E.Operand.plain(dyd), dy
    elif pro == 'act_bwd':
        op = E.Operand(dyd, tuple(dyd.shape), pro=L.PRO_ACT_BWD, x2=cd, slope=slope)
        g = torch.where(c > 0, dy, slope * dy)
    else:
        op = E.Operand(dyd, tuple(dyd.shape), pro=L.PRO_BNACT_BWD, x2=cd, pa=qa.cuda(), pb=qb.cuda(), pd=qd.cuda(),
                       ps=ks.cuda(), pt=kt.cuda(), slope=slope)
        z = bc(ks) * c + bc(kt)
        g = bc(qa) * torch.where(z > 0, dy, slope * dy) + bc(qb) * c + bc(qd)
    dx_ref = F.conv_transpose2d(_bf(g).double(), _bf(wt).double(), stride=stride, padding=1, output_padding=stride - 1)
    assert tuple(dx_ref.shape) == (n, cin, h, w)
    # the BatchNorm in front of this layer: input xb (dx's shape), constants, LeakyReLU after it
    xb = _bf(_rand((n, cin, h, w), 18))
    k4 = torch.stack([_rand((cin,), 19) * 0.5 + 1.0, _rand((cin,), 20, 0.3), _rand((cin,), 21, 0.2), _rand((cin,), 22) * 0.5 + 1.0])
    assert E.can_fuse_bn_backward(p)
    dx, part = E.conv_dgrad(p, op, bnb=(nhwc(xb).cuda().to(torch.bfloat16), k4.cuda(), slope))
    assert dx.dtype == torch.bfloat16
    assert maxrel(nchw(dx.float()), dx_ref) < TOL, 'data gradient'
    assert part is not None and part.shape[1] == 2 * cin + 1
    # reductions over the UNROUNDED gradient the epilogue holds; compare with the double reference
    gfull = dx_ref
    zb = bc(k4[0]).double() * xb.double() + bc(k4[1]).double()
    gg = torch.where(zb > 0, gfull, slope * gfull)
    xhat = (xb.double() - bc(k4[2]).double()) * bc(k4[3]).double()
    s = part.double().cpu().sum(0)

    def close(got, terms, dims):
        """a sum of N terms carried in fp32 from bf16-rounded operands: within 5e-3 of the terms' L1 norm"""
        ref, l1 = terms.sum(dim=dims), terms.abs().sum(dim=dims)
        return float(((got - ref).abs() / l1.clamp_min(1e-30)).max()) < 5e-3
    assert close(s[:cin], gg, (0, 2, 3)), 'sum of gradients'
    assert close(s[cin:2 * cin], gg * xhat, (0, 2, 3)), 'sum of gradient * xhat'
    assert close(s[2 * cin:], torch.where(zb > 0, torch.zeros_like(gfull), gfull * zb).reshape(1, -1), (1,)), 'slope term'


def test_data_gradient_with_residual(E, L):
    """the generator's trunk at a size the persistent kernels do not take (LR 24): BatchNorm-backward prologue + skip gradient"""
    n, cin, cout, h, w = 2, 64, 64, 24, 24
    p, ref, wt, b, keep = _prep(E, n, cin, cout, 1, h, w)
    assert p.kinds[1] == 2
    dy, x2 = _bf(_rand((n, cout, h, w), 31)), _bf(_rand((n, cout, h, w), 32))
    res = _bf(_rand((n, cin, h, w), 33))
    qa, qb, qd = _rand((cout,), 34) * 0.5 + 1.0, _rand((cout,), 35, 0.2), _rand((cout,), 36, 0.1)
    bc = lambda v: v[None, :, None, None]
    op = E.Operand(nhwc(dy).cuda().to(torch.bfloat16), (n, h, w, cout), pro=L.PRO_BNBWD, x2=nhwc(x2).cuda().to(torch.bfloat16),
                   pa=qa.cuda(), pb=qb.cuda(), pd=qd.cuda())
    g = bc(qa) * dy + bc(qb) * x2 + bc(qd)
    dx = E.conv_dgrad(p, op, res=nhwc(res).cuda().to(torch.bfloat16))
    dx_ref = F.conv_transpose2d(_bf(g).double(), _bf(wt).double(), padding=1) + res.double()
    assert maxrel(nchw(dx.float()), dx_ref) < TOL


def test_k_split_matches_single_slice(E, L, monkeypatch):
    """the same layer with and without the K split: sums of fp32 partial tiles in slice order against one accumulation chain"""
    n, cin, cout, h, w = 4, 256, 128, 12, 12
    x = _bf(_rand((n, cin, h, w), 41))
    xd = nhwc(x).cuda().to(torch.bfloat16)
    outs = []
    for target in ('1', '4096'):
        monkeypatch.setenv('SISR_DEEP_TARGET', target)
        monkeypatch.setenv('SISR_DEEP_MINCPS', '1')
        p, ref, wt, b, keep = _prep(E, n, cin, cout, 1, h, w)
        assert p.kinds[0] == 2
        assert (p.plans[0].deep.split == 1) == (target == '1')
        y, sp, cp = E.conv_forward(p, E.Operand.plain(xd), bias=ref.bias, stats=True)
        outs.append((y.float().cpu(), sp.cpu(), p.plans[0].deep.split))
    assert outs[1][2] == cin // 32
    assert maxrel(outs[1][0], outs[0][0]) < 4e-3                    # (one bf16 rounding of slightly different fp32 sums)
    y_ref = F.conv2d(x.double(), _bf(wt).double(), b.double(), padding=1)
    assert maxrel(nchw(outs[1][0]), y_ref) < TOL


# the discriminator's conv stack at B16 (config.py:81-82) for HR 96, and the VGG19 layers up to conv5_4 at HR 96
D_FEATS, D_STRIDES = [64, 64, 128, 128, 256, 256, 512, 512], [1, 2, 1, 2, 1, 2, 1, 2]


def _d_layers(hr):
    out, c, r = [], 3, hr
    for f, st in zip(D_FEATS, D_STRIDES):
        if c >= 32:
            out.append((16, c, f, st, r, r))
        c, r = f, r // st
    return out


VGG_LAYERS = [(16, 64, 128, 1, 48, 48), (16, 128, 128, 1, 48, 48), (16, 128, 256, 1, 24, 24), (16, 256, 256, 1, 24, 24),
              (16, 256, 512, 1, 12, 12), (16, 512, 512, 1, 12, 12), (16, 512, 512, 1, 6, 6)]


@pytest.mark.parametrize('case', _d_layers(96) + VGG_LAYERS + [(16, 64, 64, 1, 96, 96)])
def test_full_size_layers_forward_and_data_gradient(E, L, case):
    n, cin, cout, stride, h, w = case
    trunk_shape = cin == 64 and cout == 64 and stride == 1 and h % 8 == 0 and w % 16 == 0
    p, ref, wt, b, keep = _prep(E, n, cin, cout, stride, h, w, deep_dgrad=trunk_shape)
    x = _bf(_rand((n, cin, h, w), 51))
    if not trunk_shape:
        assert p.kinds[0] == 2
        y, sp, cp = E.conv_forward(p, E.Operand.act(nhwc(x).cuda().to(torch.bfloat16), 0.01), bias=ref.bias, stats=True)
        y_ref = F.conv2d(_bf(_lrelu(x, 0.01)), _bf(wt), b, stride=stride, padding=1)
        assert maxrel(nchw(y.float()), y_ref) < TOL, 'forward'
        tot, mean, var = _merge_stats(sp, cp)
        assert maxrel(mean, y_ref.double().mean(dim=(0, 2, 3))) < 2e-3
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    dy = _bf(_rand((n, cout, ho, wo), 52))
    assert p.kinds[1] == (2 if stride == 1 else 3)             # stride 2: ONE launch over the four output-parity classes
    dx = E.conv_dgrad(p, E.Operand.plain(nhwc(dy).cuda().to(torch.bfloat16)))
    dx_ref = F.conv_transpose2d(dy, _bf(wt), stride=stride, padding=1, output_padding=stride - 1)
    assert maxrel(nchw(dx.float()), dx_ref) < TOL, 'data gradient'


# ---- the classifier head (csrc/fc_head.hip; model_discriminator.py:47-53) -----------------------------------------------------
@pytest.mark.parametrize('shape', [(16, 18432, 1024), (5, 2048, 256), (16, 73728, 1024)])
def test_classifier_head_forward_backward(L, shape):
    """Linear(K, N) -> LeakyReLU -> Linear(N, 1) -> Sigmoid through fc_head.hip against torch autograd in double (exact-fp32
    products, fp32 sums in a different order: 1e-5 of the max-norm)"""
    Ef = pkg('engine')
    bsz, k, n = shape
    x = _rand((bsz, k), 61)
    w1, b1 = _rand((n, k), 62, (1.0 / k) ** 0.5), _rand((n,), 63, 0.1)
    w2, b2 = _rand((1, n), 64, (1.0 / n) ** 0.5 * 3), _rand((1,), 65, 0.1)
    g = _rand((bsz, 1), 66)
    xr, w1r, b1r, w2r, b2r = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    h1_ref = xr @ w1r.t() + b1r
    y_ref = torch.sigmoid(F.leaky_relu(h1_ref, 0.01) @ w2r.t() + b2r)
    (y_ref * g.double()).sum().backward()
    assert Ef.fc_head_ok(bsz, k, n)
    xd, w1d = x.cuda(), w1.cuda()
    h1, y = Ef.fc_head_forward(xd, w1d, b1.cuda(), w2.cuda(), b2.cuda(), 0.01)
    assert maxrel(h1, h1_ref) < 1e-5 and maxrel(y, y_ref) < 1e-5
    d1, dw2, db2, db1 = Ef.fc_head_backward(g.cuda(), y, h1, w2.cuda(), 0.01)
    assert maxrel(dw2, w2r.grad) < 1e-4 and maxrel(db2, b2r.grad) < 1e-4 and maxrel(db1, b1r.grad) < 1e-4
    dx = Ef.fc1_dgrad(d1, w1d)
    assert maxrel(dx, xr.grad) < 1e-4
    dw1 = Ef.fc_wgrad_only(d1, xd, w1d)
    assert maxrel(dw1, w1r.grad) < 1e-4


# ---- the weight gradient of the same layers (csrc/wgrad_deep.hip) -----------------------------------------------------------------
WG_SMALL = [
    (3, 64, 64, 1, 12, 12),        # bands straddling images
    (5, 64, 128, 1, 6, 6),         # many images per band
    (2, 64, 128, 1, 24, 24),
    (2, 128, 64, 1, 16, 32),       # two column tiles
    (2, 64, 64, 1, 37, 29),        # ragged
    (3, 64, 128, 2, 24, 24),       # stride 2: parity planes
    (2, 64, 64, 2, 32, 32),
    (3, 64, 64, 2, 13, 11),        # stride 2, odd sizes
    (4, 256, 512, 1, 12, 12),      # 4 x 8 channel blocks
    (4, 512, 512, 2, 12, 12),      # D's last layer
]


def _wgrad_case(E, L, case, xpro, gpro, seed=0):
    n, cin, cout, stride, h, w = case
    p, ref, wt, b, keep = _prep(E, n, cin, cout, stride, h, w, seed)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    bc = lambda v: v[None, :, None, None]
    slope = 0.2
    x = _bf(_rand((n, cin, h, w), seed + 31))
    xd = nhwc(x).cuda().to(torch.bfloat16)
    if xpro == 'none':
        x_op, xt = E.Operand.plain(xd), x
    elif xpro == 'act':
        x_op, xt = E.Operand(xd, tuple(xd.shape), pro=L.PRO_ACT, slope=slope), _lrelu(x, slope)
    else:
        pa, pd = _rand((cin,), seed + 32) * 0.5 + 1.0, _rand((cin,), seed + 33, 0.3)
        x_op = E.Operand(xd, tuple(xd.shape), pro=L.PRO_AFFINE_ACT, pa=pa.cuda(), pd=pd.cuda(), slope=slope)
        xt = _lrelu(bc(pa) * x + bc(pd), slope)
    dy = _bf(_rand((n, cout, ho, wo), seed + 34))
    c = _bf(_rand((n, cout, ho, wo), seed + 35))
    dyd, cd = nhwc(dy).cuda().to(torch.bfloat16), nhwc(c).cuda().to(torch.bfloat16)
    qa, qb, qd = _rand((cout,), seed + 36) * 0.5 + 1.0, _rand((cout,), seed + 37, 0.2), _rand((cout,), seed + 38, 0.1)
    ks, kt = _rand((cout,), seed + 39) * 0.5 + 1.0, _rand((cout,), seed + 40, 0.3)
    if gpro == 'none':
        g_op, g = E.Operand.plain(dyd), dy
    elif gpro == 'act_bwd':
        g_op, g = E.Operand(dyd, tuple(dyd.shape), pro=L.PRO_ACT_BWD, x2=cd, slope=slope), torch.where(c > 0, dy, slope * dy)
    elif gpro == 'bnbwd':
        g_op = E.Operand(dyd, tuple(dyd.shape), pro=L.PRO_BNBWD, x2=cd, pa=qa.cuda(), pb=qb.cuda(), pd=qd.cuda())
        g = bc(qa) * dy + bc(qb) * c + bc(qd)
    else:
        g_op = E.Operand(dyd, tuple(dyd.shape), pro=L.PRO_BNACT_BWD, x2=cd, pa=qa.cuda(), pb=qb.cuda(), pd=qd.cuda(), ps=ks.cuda(),
                         pt=kt.cuda(), slope=slope)
        z = bc(ks) * c + bc(kt)
        g = bc(qa) * torch.where(z > 0, dy, slope * dy) + bc(qb) * c + bc(qd)
    gw_ref = torch.nn.grad.conv2d_weight(_bf(xt).double(), (cout, cin, 3, 3), _bf(g).double(), stride=stride, padding=1)
    return p, ref, x_op, g_op, gw_ref, g.double()


def _run_wgrad(E, p, ref, x_op, g_op):
    before = E.KERNEL_COUNTS.get('wgrad_deep', 0)
    red = E.conv_wgrad(p, x_op, g_op)
    assert E.KERNEL_COUNTS.get('wgrad_deep', 0) == before + 1, 'the descriptor did not run on wgrad_deep.hip'
    wg = E.WeightGradBatch()
    wg.add(p, red)
    return wg.run()[id(ref)]


@pytest.mark.parametrize('pros', [('none', 'none'), ('act', 'act_bwd'), ('affine_act', 'bnact_bwd'), ('affine_act', 'bnbwd')])
@pytest.mark.parametrize('case', WG_SMALL)
def test_weight_gradient_prologues_and_bias(E, L, case, pros):
    """dW = sum_p x'(p * s + tap - 1) (x) dy'(p) with both operands lazy (x: BatchNorm apply + LeakyReLU of the layer in front, dy: the
    activation / BatchNorm backward of the layer itself, model_discriminator.py:5-15 differentiated); bias gradient = sum of dy'"""
    p, ref, x_op, g_op, gw_ref, g = _wgrad_case(E, L, case, *pros)
    assert p.plans[2].deep.enabled == 1
    gw, gb = _run_wgrad(E, p, ref, x_op, g_op)
    assert maxrel(gw, gw_ref) < TOL, 'weight gradient'
    l1 = g.abs().sum(dim=(0, 2, 3))
    assert float(((gb.double().cpu() - g.sum(dim=(0, 2, 3))).abs() / l1.clamp_min(1e-30)).max()) < 2e-3, 'bias gradient'
    gw2, gb2 = _run_wgrad(E, p, ref, x_op, g_op)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2), 'fixed-order slab reduction: bitwise repeatable'


def test_weight_gradient_fp32_slabs_and_pixel_block_counts(E, L, monkeypatch):
    """SISR_SLAB_BF16=0 keeps fp32 slabs; the result may not depend on how the tiles are split into pixel blocks beyond rounding"""
    case = (3, 64, 128, 2, 24, 24)
    outs = []
    for env in ({'SISR_SLAB_BF16': '0'}, {'SISR_SLAB_BF16': '0', 'SISR_WGRAD_DEEP_PB': '1'}, {'SISR_WGRAD_DEEP_PB': '7'}):
        for k in ('SISR_SLAB_BF16', 'SISR_WGRAD_DEEP_PB'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        p, ref, x_op, g_op, gw_ref, g = _wgrad_case(E, L, case, 'affine_act', 'bnact_bwd')
        if 'SISR_WGRAD_DEEP_PB' in env:
            assert p.plans[2].deep.n_pb <= int(env['SISR_WGRAD_DEEP_PB'])           # (rounded to whole tiles per block)
        gw, gb = _run_wgrad(E, p, ref, x_op, g_op)
        assert maxrel(gw, gw_ref) < (1e-3 if 'SISR_SLAB_BF16' in env else TOL)
        outs.append(gw)
    assert maxrel(outs[0], outs[1]) < 1e-5


@pytest.mark.parametrize('case', _d_layers(96) + [(16, 64, 64, 1, 24, 24)])
def test_full_size_layers_weight_gradient(E, L, case):
    """the discriminator's seven 3x3 layers at 96 x 96, B = 16, and the generator's trunk layer at a 24 x 24 input"""
    p, ref, x_op, g_op, gw_ref, g = _wgrad_case(E, L, case, 'affine_act', 'bnact_bwd')
    gw, gb = _run_wgrad(E, p, ref, x_op, g_op)
    assert maxrel(gw, gw_ref) < TOL
    l1 = g.abs().sum(dim=(0, 2, 3))
    assert float(((gb.double().cpu() - g.sum(dim=(0, 2, 3))).abs() / l1.clamp_min(1e-30)).max()) < 2e-3


@pytest.mark.parametrize('stride', [1, 2])
def test_weight_gradients_of_several_layers_in_one_launch(E, L, stride):
    """WgradDeepBatch: layers of one stride collected and launched together (grid z = layer), every member planned for its share of the
    chip -- same results as the layer alone within the slab rounding, and the reference holds for every member"""
    cases = {1: [(16, 64, 128, 1, 48, 48), (16, 128, 256, 1, 24, 24), (16, 256, 512, 1, 12, 12), (3, 64, 64, 1, 13, 11)],
             2: [(16, 64, 64, 2, 96, 96), (16, 128, 128, 2, 48, 48), (16, 256, 256, 2, 24, 24), (16, 512, 512, 2, 12, 12)]}[stride]
    members = [_wgrad_case(E, L, c, 'affine_act', 'bnact_bwd', seed=7 * i) for i, c in enumerate(cases)]
    alone = [_run_wgrad(E, p, ref, x_op, g_op) for p, ref, x_op, g_op, _, _ in members]
    wb, pending, wg = E.WgradDeepBatch(), E.PendingSlabs(), E.WeightGradBatch()
    before = E.KERNEL_COUNTS.get('wgrad_deep_batch', 0)
    for p, ref, x_op, g_op, _, _ in members:
        red = wb.add(p, x_op, g_op)
        assert red is not None
        wg.add(p, red)
    wb.run(pending)
    assert E.KERNEL_COUNTS.get('wgrad_deep_batch', 0) == before + 1 and len(pending.jobs) == len(members)
    pending.flush()
    res = wg.run()
    for (p, ref, x_op, g_op, gw_ref, g), (gw1, gb1) in zip(members, alone):
        gw, gb = res[id(ref)]
        assert maxrel(gw, gw_ref) < TOL and maxrel(gw, gw1) < TOL
        l1 = g.abs().sum(dim=(0, 2, 3))
        assert float(((gb.double().cpu() - g.sum(dim=(0, 2, 3))).abs() / l1.clamp_min(1e-30)).max()) < 2e-3


@pytest.mark.parametrize('shape', [(32, 18432, 1024), (48, 2048, 256), (128, 16384, 1024), (2, 128, 64), (200, 1024, 64), (70, 256, 128)])
def test_classifier_head_weight_gradient_from_gathered_factors(E, shape):
    """sisr_fc_wgrad_rows: dW = scale * dy^T x over the rows of ALL ranks (exact-fp32 matrix instruction) against torch in double"""
    rows, k, n = shape
    dy, x = _rand((rows, n), 91), _rand((rows, k), 92)
    w = torch.empty(n, k, device='cuda')
    assert E.fc_wgrad_rows_ok(rows, k, n)
    got = E.fc_wgrad_rows(dy.cuda(), x.cuda(), w, 1.0 / 8)
    want = (dy.double().t() @ x.double()) / 8
    assert maxrel(got, want) < 1e-5
    assert torch.equal(got, E.fc_wgrad_rows(dy.cuda(), x.cuda(), w, 1.0 / 8))
