"""Host-side engine: turns the reference's layers into launches of the gfx950 kernels.

PyTorch is plumbing here (device memory through the caching allocator, the current HIP stream,
autograd bookkeeping); every FLOP of the path runs in libsisr_hip.so.  An activation is carried
as a *lazy operand*: a raw NHWC tensor plus the per-channel affine (BatchNorm apply) and the
leaky-relu slope (PReLU / LeakyReLU / ReLU) that its consumer applies while staging tiles into
LDS, so BatchNorm / activation layers never make their own pass over HBM.
"""
import ctypes as C
import os

import torch

from . import _lib as L

# 'fp32': exact-fp32 MFMA everywhere (the parity build).  'bf16': layers with Cin % 32 == 0 run their
# contraction on the bf16 matrix cores (fp32 accumulate, fp32 statistics) and keep their NHWC tensors as bf16 in HBM
# (storage_bf16() below).
PRECISION = os.environ.get('SISR_PRECISION', 'fp32')


def set_precision(p):
    """'fp32': fp32 tensors, exact fp32 matrix instructions.  'bf16x3': fp32 tensors and arithmetic as 'fp32', the trunk
    contractions on the bf16 matrix instruction over (hi, lo) bf16 pairs of every fp32 operand (SisrConvDesc.mfma_split; operands
    good to 2^-17 relative -- inside the 1e-3 parity bar by two orders of magnitude).  'bf16': bf16 tensors in HBM."""
    global PRECISION
    assert p in ('fp32', 'bf16x3', 'bf16')
    PRECISION = p


def mfma_split():
    return int(PRECISION == 'bf16x3')


def storage_bf16():
    """bf16 build: NHWC activation / gradient tensors whose channel count is a multiple of 32 live in HBM as bf16
    (SURVEY 8d's bf16 bytes); arithmetic, accumulation and BatchNorm statistics stay fp32.  SISR_STORAGE=f32 keeps
    fp32 tensors with bf16 matrix-core operands (the earlier layout; A/B switch)."""
    return PRECISION == 'bf16' and os.environ.get('SISR_STORAGE', 'bf16') != 'f32'


def act_dtype(channels=64):
    return torch.bfloat16 if storage_bf16() and channels % 32 == 0 else torch.float32


def _bf(t):
    return int(t is not None and t.dtype == torch.bfloat16)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _copy_struct(s):
    return type(s).from_buffer_copy(s)


def _align4(n):
    return (n + 3) & ~3


class Operand:
    """x1 (and x2) + prologue: see SISR_PRO_* in include/sisr_hip.h.  dims = logical (N,H,W,C)."""
    __slots__ = ('x1', 'x2', 'pa', 'pb', 'pd', 'ps', 'pt', 'mode', 'pro', 'slope', 'dims', 'x_out', 'fin')

    def __init__(self, x1, dims, pro=L.PRO_NONE, mode=L.X_NHWC, x2=None, pa=None, pb=None, pd=None,
                 ps=None, pt=None, slope=None):
        self.x1, self.x2, self.pa, self.pb, self.pd, self.ps, self.pt = x1, x2, pa, pb, pd, ps, pt
        self.mode, self.pro, self.slope, self.dims = mode, pro, slope, dims
        self.x_out = None
        self.fin = None                     # LazyBN whose scale / shift are this operand's pa / pd (deferred finalisation)

    @staticmethod
    def plain(t, dims=None, mode=L.X_NHWC):
        return Operand(t, dims or tuple(t.shape), mode=mode)

    @staticmethod
    def act(t, slope, dims=None):
        """lrelu(t, slope); slope: device scalar tensor (PReLU weight) or float."""
        return Operand(t, dims or tuple(t.shape), pro=L.PRO_ACT, slope=slope)

    @staticmethod
    def affine_act(t, scale, shift, slope=1.0):
        return Operand(t, tuple(t.shape), pro=L.PRO_AFFINE_ACT, pa=scale, pd=shift, slope=slope)

    @staticmethod
    def res_affine(res, res_slope, t, scale, shift, out):
        """lrelu(res, res_slope) + (scale*t + shift) -- a residual block's skip sum x + BN2(c2) (model_generator.py:19)
        formed in the consuming conv's staging; the conv also stores the sum to `out` (persistent trunk kernels only:
        ask trunk_takes_skip_sum() first)."""
        op = Operand(res, tuple(res.shape), pro=L.PRO_RES_AFFINE, x2=t, pa=scale, pd=shift, slope=res_slope)
        op.x_out = out
        return op

    def fill(self, d, g=False):
        """write this operand into a ConvDesc / WgradDesc (g=True: the output-gradient operand)"""
        names = (('g1', 'g2', 'qa', 'qb', 'qd', 'qs', 'qt', 'g_mode', 'gpro_mode', 'gpro_slope_p', 'gpro_slope')
                 if g else
                 ('x1', 'x2', 'pa', 'pb', 'pd', 'ps', 'pt', 'x_mode', 'pro_mode', 'pro_slope_p', 'pro_slope'))
        vals = (self.x1, self.x2, self.pa, self.pb, self.pd, self.ps, self.pt)
        for n, v in zip(names[:7], vals):
            setattr(d, n, _ptr(v))
        assert self.x2 is None or self.x2.dtype == self.x1.dtype, 'operand pair with mixed storage types'
        setattr(d, 'g_bf16' if g else 'x_bf16', _bf(self.x1))
        setattr(d, names[7], self.mode)
        setattr(d, names[8], self.pro)
        if not g and hasattr(d, 'x_out'):
            d.x_out = _ptr(self.x_out)
        if isinstance(self.slope, torch.Tensor):
            setattr(d, names[9], self.slope.data_ptr())
            setattr(d, names[10], 1.0)
        else:
            setattr(d, names[9], None)
            setattr(d, names[10], 1.0 if self.slope is None else float(self.slope))


class ConvGeom:
    """Static geometry of one convolution + cached kernel plans per input shape."""

    def __init__(self, cin, cout, k, stride=1, pad=None, shuffle2=False, deep_dgrad=False):
        self.cin, self.cout, self.k, self.stride = cin, cout, k, stride
        self.pad = (k // 2) if pad is None else pad
        self.shuffle2 = shuffle2
        # trunk-shaped layers (3x3, 64 -> 64, stride 1) whose DATA GRADIENT never arrives with a BatchNorm-backward prologue
        # (the frozen VGG stack): the persistent trunk kernel refuses it, so it is planned for conv_deep.hip instead
        self.deep_dgrad = deep_dgrad
        # the data gradient of this layer never arrives through a BatchNorm-backward prologue (frozen VGG): its staging is
        # cheap, so the planner may pick the 64-cout tiles that plenty of pixel tiles favour (sisr_conv2d_deep_plan's prefer_bn)
        self.light_backward = deep_dgrad
        self._plans = {}

    def _deep_ok(self, role, h, w):
        """this role (0 forward, 1 data gradient) of the layer is planned for the split-K implicit-GEMM family (conv_deep.hip):
        bf16 tensors, 3x3, channels in 32s / 64s, no PixelShuffle store; the trunk geometry stays with the persistent trunk
        kernels wherever they take it (H % 8 == 0, W % 16 == 0)"""
        if PRECISION != 'bf16' or not storage_bf16() or os.environ.get('SISR_DEEP', '1') == '0':
            return False
        if self.k != 3 or self.shuffle2:
            return False
        cin, cout = (self.cin, self.cout) if role == 0 else (self.cout, self.cin)
        if cin % 32 or cout % 64:
            return False
        trunk_shape = (self.cin == 64 and self.cout == 64 and self.stride == 1 and h % 8 == 0 and w % 16 == 0
                       and os.environ.get('SISR_TRUNK', '1') != '0')
        if trunk_shape and not (role == 1 and self.deep_dgrad):
            return False
        return True

    def out_hw(self, h, w):
        return ((h + 2 * self.pad - self.k) // self.stride + 1, (w + 2 * self.pad - self.k) // self.stride + 1)

    def _s2_class_plan(self, lib, n, h, w, ho, wo, py, px):
        khc, pady, r0y = _s2_taps(self.k, self.pad, py)
        kwc, padx, r0x = _s2_taps(self.k, self.pad, px)
        hc, wc = (h - py + 1) // 2, (w - px + 1) // 2
        if khc == 0 or kwc == 0 or hc <= 0 or wc <= 0:
            return None
        d = L.ConvDesc()
        d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, ho, wo, self.cout, hc, wc, self.cin
        d.KH, d.KW, d.stride, d.pad_y, d.pad_x = khc, kwc, 1, pady, padx
        d.y_sy = d.y_sx = 2
        d.y_oy, d.y_ox, d.y_H, d.y_W = py, px, h, w
        bf = PRECISION == 'bf16' and self.cout % 32 == 0 and lib.sisr_conv2d_plan_bf16(C.byref(d)) == 0
        if not bf:
            L.check(lib.sisr_conv2d_plan(C.byref(d)), 'sisr_conv2d_plan(dgrad stride-2 class)')
        elif self._deep_ok(1, h, w) and lib.sisr_conv2d_deep_plan(C.byref(d), 0, 0 if self.light_backward else 128, 1) == 0:
            bf = 2                                                    # conv_deep.hip
        return (d, r0y, r0x, bf if bf == 2 else bool(bf))

    def _s2_deep_plan(self, lib, n, h, w, ho, wo):
        """stride-2 data gradient as ONE conv_deep.hip launch over the four output-parity classes (even sizes): -> (descriptor of
        the 2 x 2-tap class convolution, [(taps y, taps x, R0y, R0x)] per class c = 2 py + px) or None"""
        if h % 2 or w % 2 or self.k != 3 or self.pad != 1 or not self._deep_ok(1, h, w) or os.environ.get('SISR_DEEP_S2X4', '1') == '0':
            return None
        d = L.ConvDesc()
        d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, ho, wo, self.cout, h // 2, w // 2, self.cin
        d.KH = d.KW = 2
        d.stride, d.pad_y, d.pad_x = 1, 0, 0
        d.y_sy = d.y_sx = 2
        d.y_oy = d.y_ox = 0
        d.y_H, d.y_W = h, w
        if (ho, wo) != (h // 2, w // 2) or lib.sisr_conv2d_plan_bf16(C.byref(d)) != 0:
            return None
        if lib.sisr_conv2d_deep_plan(C.byref(d), 0, 0 if self.light_backward else 128, 4) != 0:
            return None
        cls = []
        for py in (0, 1):
            for px in (0, 1):
                khc, pady, r0y = _s2_taps(self.k, self.pad, py)
                kwc, padx, r0x = _s2_taps(self.k, self.pad, px)
                assert pady == 0 and padx == 0 and khc <= 2 and kwc <= 2
                cls.append((khc, kwc, r0y, r0x))
        return d, cls

    def plans(self, n, h, w, max_pixel_blocks=None):
        """-> (fwd desc, dgrad desc | [4 class descs] | None, wgrad desc, kinds) where kinds =
        (fwd_bf16, dgrad_bf16, wgrad_bf16) tells which kernel family each template was planned for."""
        if max_pixel_blocks is None:
            max_pixel_blocks = int(os.environ.get('SISR_WGRAD_PIXEL_BLOCKS', '512'))      # A/B knob of the wgrad grids
        key = (n, h, w, PRECISION, storage_bf16(), max_pixel_blocks, os.environ.get('SISR_DEEP', '1'), os.environ.get('SISR_WGRAD_DEEP', '1'),
               os.environ.get('SISR_WGRAD_DEEP_PB', ''))
        if key in self._plans:
            return self._plans[key]
        lib = L.lib()
        ho, wo = self.out_hw(h, w)
        f = L.ConvDesc()
        f.N, f.H, f.W, f.Cin, f.Ho, f.Wo, f.Cout = n, h, w, self.cin, ho, wo, self.cout
        f.KH = f.KW = self.k
        f.stride, f.pad_y, f.pad_x = self.stride, self.pad, self.pad
        f.y_sy = f.y_sx = 1
        f.y_oy = f.y_ox = 0
        f.y_H, f.y_W = ho, wo
        f.y_mode = L.Y_SHUFFLE2 if self.shuffle2 else L.Y_NHWC
        want_bf16 = PRECISION == 'bf16' and self.k * self.k <= 9
        f_bf = want_bf16 and self.cin % 32 == 0 and lib.sisr_conv2d_plan_bf16(C.byref(f)) == 0
        if not f_bf:
            L.check(lib.sisr_conv2d_plan(C.byref(f)), 'sisr_conv2d_plan(fwd)')
        elif self._deep_ok(0, h, w) and lib.sisr_conv2d_deep_plan(C.byref(f), 0, 0, 1) == 0:
            f_bf = 2                                                  # kind 2: conv_deep.hip (its own weight image)
        d = None
        d_bf = False
        if self.stride == 1:
            d = L.ConvDesc()      # data gradient: conv over dy with flipped taps, roles swapped
            d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, ho, wo, self.cout, h, w, self.cin
            d.KH = d.KW = self.k
            d.stride = 1
            d.pad_y = d.pad_x = self.k - 1 - self.pad
            d.y_sy = d.y_sx = 1
            d.y_H, d.y_W = h, w
            d_bf = want_bf16 and self.cout % 32 == 0 and lib.sisr_conv2d_plan_bf16(C.byref(d)) == 0
            if not d_bf:
                L.check(lib.sisr_conv2d_plan(C.byref(d)), 'sisr_conv2d_plan(dgrad)')
            elif self._deep_ok(1, h, w) and lib.sisr_conv2d_deep_plan(C.byref(d), 0, 0 if self.light_backward else 128, 1) == 0:
                d_bf = 2
        else:
            x4 = self._s2_deep_plan(lib, n, h, w, ho, wo) if want_bf16 else None
            if x4 is not None:
                d, d_bf = S2x4(*x4), 3                               # kind 3: one launch over the four parity classes
            else:
                d = [self._s2_class_plan(lib, n, h, w, ho, wo, py, px) for py in (0, 1) for px in (0, 1)]
        g = L.WgradDesc()
        g.N, g.H, g.W, g.Cin, g.Ho, g.Wo, g.Cout = n, h, w, self.cin, ho, wo, self.cout
        g.KH = g.KW = self.k
        g.stride, g.pad_y, g.pad_x = self.stride, self.pad, self.pad
        g_bf = False
        if want_bf16 and self.cin % 32 == 0:
            # few-channel outputs (the generator's 3-channel last conv): the bf16 kernel runs on the gradient
            # image padded to 4 channels (conv_wgrad materialises it), 16x the exact-fp32 matrix rate
            g.Cout = (self.cout + 3) // 4 * 4 if self.cout < 32 else self.cout
            g_bf = g.Cout % 4 == 0 and lib.sisr_wgrad_plan_bf16(C.byref(g), max_pixel_blocks) == 0
            if g_bf and self.k == 3 and self.pad == 1:
                lib.sisr_wgrad_deep_plan(C.byref(g), 0)               # 3x3, channels in 64s: wgrad_deep.hip (g.deep.enabled)
        if not g_bf:
            g.Cout = self.cout
            L.check(lib.sisr_wgrad_plan(C.byref(g), max_pixel_blocks), 'sisr_wgrad_plan')
        g.slab_stride = g.slab_elems + g.CoutPad
        self._plans[key] = (f, d, g, (f_bf if f_bf == 2 else bool(f_bf), d_bf if d_bf in (2, 3) else bool(d_bf), bool(g_bf)))
        return self._plans[key]


class S2x4:
    """the data gradient of a stride-2 layer planned as one conv_deep.hip launch over its four output-parity classes"""

    def __init__(self, desc, classes):
        self.desc, self.classes = desc, classes          # classes[c] = (taps y, taps x, R0y, R0x), c = 2 py + px

    def image_slots(self, c, cout_fwd, cin_fwd):
        """fp32 slots of class c's weight image [cout_fwd / 32][taps y][cin_fwd][2 * 32 + 8] bf16 (KW = 2 row format)"""
        return ((cout_fwd // 32) * self.classes[c][0] * cin_fwd * 72 + 1) // 2


def _s2_taps(k, pad, parity):
    """stride-2 data gradient along one axis, output index 2a+parity: contributing dy index a+delta
    for forward taps r with (parity+pad-r) even.  -> (n_taps, pad', R0) with r = R0 - 2 r'."""
    deltas = sorted((parity + pad - r) // 2 for r in range(k) if (parity + pad - r) % 2 == 0)
    if not deltas:
        return 0, 0, 0
    return len(deltas), -deltas[0], parity + pad - 2 * deltas[0]


class ConvRef:
    """One convolution of a network: geometry + where its parameters / spectral-norm buffers live."""

    def __init__(self, geom, weight, bias, u=None, v=None):
        self.geom, self.weight, self.bias, self.u, self.v = geom, weight, bias, u, v


class Prepared:
    """Per-forward products of sisr_weights_prepare for one conv (kept for the backward pass)."""
    __slots__ = ('ref', 'plans', 'kinds', 'wpk_fwd', 'wpk_dgrad', 'sigma', 'inv_sigma', 'u_used', 'v_used', 'lanes', 'ldsimg')


def _trunk_ldsimg(gm, plan_f, plan_d, kinds):
    """(forward, data-gradient) mode of the LDS-order weight image of the fp32-tensor trunk conv (SisrWeightDesc.f_ldsimg /
    d_ldsimg): 0 none, 1 fp32 values, 2 split pairs -- for 3x3 64 -> 64 convs on the fp32 kernels"""
    if os.environ.get('SISR_TRUNK_LDSIMG', '1') == '0' or gm.k != 3 or gm.cin != 64 or gm.cout != 64 or gm.stride != 1:
        return 0, 0
    mode = 2 if mfma_split() else 1
    okf = not kinds[0] and plan_f.plan.CK == 32 and plan_f.plan.CoutPad == 64 and plan_f.plan.n_chunk == 2
    okd = (plan_d is not None and not isinstance(plan_d, list) and not kinds[1] and plan_d.plan.CK == 32
           and plan_d.plan.CoutPad == 64 and plan_d.plan.n_chunk == 2)
    return (mode if okf else 0), (mode if okd else 0)


def _trunk_lanes(gm, plan_f, plan_d, kinds):
    """(forward, data-gradient): the bf16 weight buffer also gets the lane-order image of the persistent trunk kernels
    (SisrWeightDesc.bf_f_lanes / bf_d_lanes) -- only where those kernels can take the layer: 3x3, stride 1, 64 input channels,
    64 couts (or the 256 of the upscale conv) on the generic bf16 family (kind True, not the deep family's 2)"""
    if os.environ.get('SISR_TRUNK_LANES', '1') == '0' or gm.k != 3 or gm.stride != 1:
        return False, False
    lf = kinds[0] is True and gm.cin == 64 and plan_f.plan.CK == 32 and plan_f.plan.CoutPad in (64, 256)
    ld = (kinds[1] is True and plan_d is not None and not isinstance(plan_d, list) and gm.cout == 64 and gm.cin == 64
          and plan_d.plan.CK == 32 and plan_d.plan.CoutPad == 64)
    return lf, ld


def _img_slots(desc, kind, lanes=False, ldsimg=0):
    """fp32 slots of one packed weight image: kind 2 = conv_deep.hip's bf16 image, True = the generic bf16 image (twice with the
    lane-order copy), False = the fp32 image (plus the LDS-order copy)"""
    if kind == 2:
        return (desc.deep.wimg_elems + 1) // 2
    if kind:
        return ((desc.plan.wpk_elems + 1) // 2) * (2 if lanes else 1)
    return desc.plan.wpk_elems + (L.WLDS_WORDS if ldsimg else 0)


_WEIGHT_EPOCH = [0]


def invalidate_weight_caches():
    """Packed weight images kept across forwards (prepare_weights(cache=...)) are valid only while the weights they were made from
    are: anything that changes parameters WITHOUT torch's version counter noticing calls this -- the fused Adam step (its kernel
    writes through raw pointers) and every HIP-graph capture / segment begin (an optimizer step the capture cannot see runs at
    the boundary when the graph is replayed, and the cached buffer of another capture is that graph's private memory)."""
    _WEIGHT_EPOCH[0] += 1


def prepare_weights(items, training, need_dgrad=True, cache=None):
    """items: [(ConvRef, n, h, w)].  Spectral-norm power iteration (in place on u/v when training), sigma, and the packed images
    for fwd and dgrad.  Images of the generic / trunk kernels hold W / sigma and are rebuilt on every call; the conv_deep.hip
    images (kind 2) hold W_orig itself -- those kernels apply 1 / sigma in their epilogue (Prepared.inv_sigma) -- so with `cache`
    (a dict owned by the module) they are packed once per optimizer step, not once per forward: the discriminator runs three
    forwards per SRGAN iteration, two of them on unchanged weights (train.py:132,156,174)."""
    lib = L.lib()
    dev = items[0][0].weight.device
    total, dtotal, small = 0, 0, 0
    metas = []

    def alloc(desc, kind, lanes=False, ldsimg=0):
        nonlocal total, dtotal
        n_ = _align4(_img_slots(desc, kind, lanes, ldsimg))
        if kind == 2:
            off = ('d', dtotal, n_)
            dtotal += n_
        else:
            off = ('b', total, n_)
            total += n_
        return off
    for ref, n, h, w in items:
        f, d, g, kinds = ref.geom.plans(n, h, w)
        lanes = _trunk_lanes(ref.geom, f, d, kinds)
        ldsimg = _trunk_ldsimg(ref.geom, f, d, kinds)
        off_f = alloc(f, kinds[0], lanes[0], ldsimg[0])
        off_d = None
        if need_dgrad and isinstance(d, S2x4):
            off_d = []
            for c in range(4):
                n_ = _align4(d.image_slots(c, ref.geom.cout, ref.geom.cin))
                off_d.append(('d', dtotal, n_))
                dtotal += n_
        elif need_dgrad and isinstance(d, list):
            off_d = [None if cls is None else alloc(cls[0], cls[3]) for cls in d]
        elif need_dgrad and d is not None:
            off_d = alloc(d, kinds[1], lanes[1], ldsimg[1])
        off_s = small
        rows_, cols_ = ref.geom.cout, ref.geom.cin * ref.geom.k * ref.geom.k
        small += 4 + (_align4(rows_) + _align4(cols_) if ref.u is not None else 0)
        off_w = small                              # power-iteration scratch: [ceil(rows/16)][cols] + [rows]
        small += _align4((rows_ + 15) // 16 * cols_ + rows_) if ref.u is not None else 0
        metas.append((off_f, off_d, off_s, off_w))
    big = torch.empty(total, dtype=torch.float32, device=dev)
    sm = torch.empty(small, dtype=torch.float32, device=dev)
    deep_hit = False
    deep = None
    if dtotal:
        key = (_WEIGHT_EPOCH[0], dtotal, need_dgrad, torch.cuda.is_current_stream_capturing(),
               tuple((id(ref.weight), ref.weight.data_ptr(), ref.weight._version, n, h, w) for ref, n, h, w in items))
        if cache is not None and cache.get('key') == key and os.environ.get('SISR_WCACHE', '1') != '0':
            deep, deep_hit = cache['deep'], True
        else:
            deep = torch.empty(dtotal, dtype=torch.float32, device=dev)
            if cache is not None:
                cache['key'], cache['deep'] = key, deep

    def view(off):
        return None if off is None else (deep if off[0] == 'd' else big)[off[1]:off[1] + off[2]]
    table = (L.WeightDesc * len(items))()
    out = []
    max_rows = max_cols = 1
    deep_cout = deep_cin = 0
    for i, ((ref, n, h, w), (off_f, off_d, off_s, off_w)) in enumerate(zip(items, metas)):
        f, d, g, kinds = ref.geom.plans(n, h, w)
        gm = ref.geom
        p = Prepared()
        p.ref, p.plans, p.kinds = ref, (f, d, g), kinds
        p.lanes = _trunk_lanes(gm, f, d, kinds)
        p.ldsimg = _trunk_ldsimg(gm, f, d, kinds)
        p.wpk_fwd = view(off_f)
        p.wpk_dgrad = [view(o) for o in off_d] if isinstance(off_d, list) else view(off_d)
        p.sigma = sm[off_s:off_s + 1]
        p.inv_sigma = sm[off_s + 1:off_s + 2]          # written next to sigma: the conv_deep.hip epilogue scale
        t = table[i]
        t.w_orig = ref.weight.data_ptr()
        t.sigma = p.sigma.data_ptr()
        t.wdp_scaled = 0
        has_deep = False
        if kinds[0] == 2:
            has_deep = True
            if not deep_hit:
                t.wdp_fwd = p.wpk_fwd.data_ptr()
        elif kinds[0]:
            t.wbf_fwd, t.bf_f_CoutPad, t.bf_f_CK = p.wpk_fwd.data_ptr(), f.plan.CoutPad, f.plan.CK
            t.bf_f_lanes = int(p.lanes[0])
        else:
            t.wpk_fwd = p.wpk_fwd.data_ptr()
            t.f_ldsimg = p.ldsimg[0]
        if kinds[1] == 3:
            if off_d is not None:
                has_deep = True
                t.wdp_cls_kw = 2
                for ci, ((khc, kwc, r0y, r0x), buf) in enumerate(zip(d.classes, p.wpk_dgrad)):
                    t.c_KH[ci], t.c_KW[ci], t.c_R0y[ci], t.c_R0x[ci] = khc, kwc, r0y, r0x
                    if not deep_hit:
                        t.wdp_dcls[ci] = buf.data_ptr()
        elif kinds[1] == 2:
            if off_d is not None:
                has_deep = True
                if not deep_hit:
                    t.wdp_dgrad = p.wpk_dgrad.data_ptr()
        elif kinds[1]:
            if off_d is not None:                                     # (need_dgrad=False: no data-gradient image is packed)
                t.wbf_dgrad, t.bf_d_CoutPad, t.bf_d_CK = p.wpk_dgrad.data_ptr(), d.plan.CoutPad, d.plan.CK
                t.bf_d_lanes = int(p.lanes[1])
        else:
            t.wpk_dgrad = None if isinstance(p.wpk_dgrad, list) else _ptr(p.wpk_dgrad)
            t.d_ldsimg = p.ldsimg[1] if (off_d is not None and not isinstance(off_d, list)) else 0
        t.Cout, t.Cin, t.KH, t.KW = gm.cout, gm.cin, gm.k, gm.k
        t.training, t.shuffle2 = int(training), int(gm.shuffle2)
        t.f_CK, t.f_PS, t.f_KROWP, t.f_n_chunk, t.f_CoutPad = (f.plan.CK, f.plan.PS, f.plan.KROWP,
                                                                 f.plan.n_chunk, f.plan.CoutPad)
        if isinstance(off_d, list) and not isinstance(d, S2x4):
            for ci, (cls, buf) in enumerate(zip(d, p.wpk_dgrad)):
                if cls is None:
                    continue
                cd, r0y, r0x, cbf = cls
                t.c_KH[ci], t.c_KW[ci], t.c_R0y[ci], t.c_R0x[ci] = cd.KH, cd.KW, r0y, r0x
                if cbf == 2:
                    has_deep = True
                    if not deep_hit:
                        t.wdp_dcls[ci] = buf.data_ptr()
                    continue
                if cbf:
                    t.wbf_dcls[ci], t.bf_c_CoutPad[ci] = buf.data_ptr(), cd.plan.CoutPad
                    continue
                t.wpk_dcls[ci] = buf.data_ptr()
                t.c_CK[ci], t.c_PS[ci], t.c_KROWP[ci] = cd.plan.CK, cd.plan.PS, cd.plan.KROWP
                t.c_n_chunk[ci], t.c_CoutPad[ci] = cd.plan.n_chunk, cd.plan.CoutPad
        elif off_d is not None and not kinds[1]:
            t.d_CK, t.d_PS, t.d_KROWP, t.d_n_chunk, t.d_CoutPad = (d.plan.CK, d.plan.PS, d.plan.KROWP,
                                                                     d.plan.n_chunk, d.plan.CoutPad)
        if has_deep:
            deep_cout, deep_cin = max(deep_cout, gm.cout), max(deep_cin, gm.cin)
        p.u_used = p.v_used = None
        max_rows, max_cols = max(max_rows, gm.cout), max(max_cols, gm.cin * gm.k * gm.k)
        if ref.u is not None:
            rows, cols = gm.cout, gm.cin * gm.k * gm.k
            t.sn_work = sm[off_w:].data_ptr()
            p.u_used = sm[off_s + 4:off_s + 4 + rows]
            p.v_used = sm[off_s + 4 + _align4(rows):off_s + 4 + _align4(rows) + cols]
            t.u, t.v = ref.u.data_ptr(), ref.v.data_ptr()
            t.u_used, t.v_used = p.u_used.data_ptr(), p.v_used.data_ptr()
        out.append(p)
    tab_dev = _table_to_device(table, dev)
    any_sn = any(ref.u is not None for ref, _, _, _ in items)
    # sigma must exist before any image that holds W / sigma is packed; weights without spectral norm get sigma = 1 from the
    # finishing kernel of the power iteration (one workgroup per weight) -- skipped only when nothing but cached images is left
    L.check(lib.sisr_weights_sn(tab_dev.data_ptr(), len(items), max_rows, max_cols, _stream()), 'sisr_weights_sn')
    if total:
        L.check(lib.sisr_weights_pack(tab_dev.data_ptr(), len(items), max_rows, max_cols, _stream()), 'sisr_weights_pack')
    if dtotal and not deep_hit:
        L.check(lib.sisr_weights_pack_deep(tab_dev.data_ptr(), len(items), deep_cout, deep_cin, _stream()), 'sisr_weights_pack_deep')
    return out, (big, sm, tab_dev, deep)


_PIN_RING = {'buf': None, 'off': 0, 'half': 0, 'events': [[], []], 'streams': {}}
_PIN_CAPTURE = {'buf': None, 'off': 0}   # bump allocator for tables referenced by captured graphs
_PIN_KEEP = []                           # ... whose pinned storage must outlive every replay
_PIN_RING_BYTES = 4 << 20
_PIN_CAPTURE_CHUNK = 1 << 20


def reserve_capture_tables(nbytes):
    """Make sure the pinned staging area for descriptor tables of CAPTURED launches has `nbytes` free -- called by
    graph.GraphedStep BEFORE capture_begin with what its warm-up runs consumed: a pinned host allocation inside a
    stream capture invalidates the capture (hipHostMalloc is not capturable), so the bump allocator must never have to
    grow while one is running."""
    cap = _PIN_CAPTURE
    if cap['buf'] is None or cap['off'] + nbytes > cap['buf'].numel():
        assert not torch.cuda.is_current_stream_capturing()
        cap['buf'] = torch.empty(max(nbytes, _PIN_CAPTURE_CHUNK), dtype=torch.uint8, pin_memory=True)
        cap['off'] = 0
        _PIN_KEEP.append(cap['buf'])


_TABLE_BYTES = [0]                       # bytes of tables staged so far (GraphedStep sizes its reservation from the delta)


def table_bytes_staged():
    return _TABLE_BYTES[0]


def _table_to_device(table, dev):
    """Descriptor table -> device without a host synchronisation: staged in pinned memory and copied
    asynchronously on the current stream.  Eager mode uses a 4 MB pinned ring in two halves: when a half is full an
    event is recorded on EVERY stream that issued copies out of it (tables are also uploaded from warm-up side streams
    and from the capture stream), and the host waits for those events before it writes into the half again (normally
    long complete: a half holds thousands of tables), so a slot is never rewritten while its copy may still be pending
    however far the GPU lags behind the host.  Under HIP-graph capture every table gets its own slice of a pinned
    bump buffer that is kept alive forever, because the captured copy node re-reads it on each replay; GraphedStep
    reserves that buffer before the capture begins (reserve_capture_tables)."""
    raw = bytes(table)
    n = (len(raw) + 255) & ~255
    _TABLE_BYTES[0] += n
    if torch.cuda.is_current_stream_capturing():
        cap = _PIN_CAPTURE
        if cap['buf'] is None or cap['off'] + n > cap['buf'].numel():
            # not reserved (a caller capturing without GraphedStep): the allocation below invalidates the capture on
            # ROCm 7.2 -- say why instead of failing later with a generic capture error
            raise RuntimeError('descriptor-table staging exhausted inside a stream capture: call '
                               'engine.reserve_capture_tables(nbytes) before capture_begin (GraphedStep does)')
        host = cap['buf'][cap['off']:cap['off'] + n]
        cap['off'] += n
        host[:len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
        return host.to(dev, non_blocking=True)
    ring = _PIN_RING
    if ring['buf'] is None:
        ring['buf'] = torch.empty(_PIN_RING_BYTES, dtype=torch.uint8, pin_memory=True)
        reserve_capture_tables(_PIN_CAPTURE_CHUNK)                          # allocated OUTSIDE capture
    half_bytes = _PIN_RING_BYTES // 2
    assert n <= half_bytes, 'descriptor table larger than half the pinned ring'
    if ring['off'] + n > (ring['half'] + 1) * half_bytes:          # this half is full: fence it, move to the other
        evs = []
        for st in ring['streams'].values():                         # every stream that copied out of the full half
            ev = torch.cuda.Event()
            ev.record(st)
            evs.append(ev)
        ring['events'][ring['half']] = evs
        ring['streams'] = {}
        ring['half'] ^= 1
        ring['off'] = ring['half'] * half_bytes
        for ev in ring['events'][ring['half']]:
            ev.synchronize()                                        # copies out of the half we are about to rewrite
        ring['events'][ring['half']] = []
    host = ring['buf'][ring['off']:ring['off'] + n]
    ring['off'] += n
    host[:len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
    cur = torch.cuda.current_stream()
    ring['streams'][cur.cuda_stream] = cur
    return host.to(dev, non_blocking=True)


def _attach_deep(desc, image, dev, prep):
    """descriptor planned for conv_deep.hip: hand it that family's weight image (instead of `wpk`), the 1 / sigma its epilogue
    applies (the image holds W_orig) and a split workspace"""
    desc.wdeep, desc.wpk = image.data_ptr(), None
    desc.epi_scale_p = prep.inv_sigma.data_ptr()
    ws = None
    if desc.deep.ws_bytes > 0:
        ws = torch.empty((desc.deep.ws_bytes // 4,), dtype=torch.float32, device=dev)
        desc.deep_ws = ws.data_ptr()
    return ws


def conv_forward(prep, op, bias=None, y_mode=None, epi=L.EPI_NONE, stats=False, res=None, out=None):
    """Launch the forward conv of `prep` on lazy operand `op`.  Returns (y, stat_part, cnt_part)."""
    lib = L.lib()
    f = _copy_struct(prep.plans[0])
    gm = prep.ref.geom
    n, h, w, c = op.dims
    assert (n, h, w, c) == (f.N, f.H, f.W, f.Cin), ((n, h, w, c), (f.N, f.H, f.W, f.Cin))
    if y_mode is not None:
        f.y_mode = y_mode
    dev = op.x1.device
    if out is None:
        if f.y_mode == L.Y_NCHW:
            out = torch.empty((n, gm.cout, f.Ho, f.Wo), dtype=torch.float32, device=dev)
        elif f.y_mode == L.Y_SHUFFLE2:
            out = torch.empty((n, 2 * f.Ho, 2 * f.Wo, gm.cout // 4), dtype=act_dtype(gm.cout // 4), device=dev)
        else:
            out = torch.empty((n, f.Ho, f.Wo, gm.cout), dtype=act_dtype(gm.cout) if res is None else res.dtype,
                              device=dev)
    op.fill(f)
    f.wpk, f.bias, f.res, f.y = prep.wpk_fwd.data_ptr(), _ptr(bias), _ptr(res), out.data_ptr()
    f.y_bf16, f.res_bf16 = _bf(out), _bf(res)
    f.epi_act = epi
    ws = _attach_deep(f, prep.wpk_fwd, dev, prep) if prep.kinds[0] == 2 else None          # (kept alive until the launch below)
    f.mfma_split = mfma_split()
    f.plan.variant = int(prep.lanes[0]) | (2 * prep.ldsimg[0])           # bit 0: lane-order bf16 image; bits 1-2: LDS-order fp32 image (mode)
    fin = op.fin
    if fin is not None and not fin.done:
        # deferred BatchNorm finalisation: by this conv when it runs on a persistent trunk kernel, else stand-alone first
        fin.fill(f)
        if os.environ.get('SISR_FUSE_BNFIN', '1') != '0' and \
                prep.kinds[0] != 2 and \
                (lib.sisr_conv2d_trunk_eligible if prep.kinds[0] else lib.sisr_conv2d_trunk_f32_eligible)(C.byref(f)) == 1:
            fin.done = True
        else:
            f.fin_stat = None
            fin.ensure()
    sp = cp = None
    if stats:
        # rows of the statistics partials: one per tile, or one per workgroup on the persistent trunk kernel.  The row
        # count depends on which kernel takes the descriptor, and that depends on the fusions requested (the upscale
        # variant of the trunk kernel has no statistics epilogue): ask with the statistics pointers already non-null
        f.stat_part = f.cnt_part = f.y
        rows = (lib.sisr_conv2d_bf16_parts if prep.kinds[0] else lib.sisr_conv2d_f32_parts)(C.byref(f))
        sp = torch.empty((rows, 2, gm.cout), dtype=torch.float32, device=dev)
        cp = torch.empty((rows,), dtype=torch.float32, device=dev)
        f.stat_part, f.cnt_part = sp.data_ptr(), cp.data_ptr()
    if prep.kinds[0]:
        L.check(lib.sisr_conv2d_bf16(C.byref(f), _stream()), 'sisr_conv2d_bf16(fwd)')
    else:
        L.check(lib.sisr_conv2d_f32(C.byref(f), _stream()), 'sisr_conv2d_f32(fwd)')
    return out, sp, cp


def trunk_takes_skip_sum(prep, res, t):
    """the forward conv of `prep` runs on a persistent trunk kernel that can form the skip sum lrelu(res) + BN(t) in its
    staging (Operand.res_affine); SISR_FUSE_SKIP=0 keeps the separate elementwise pass"""
    if os.environ.get('SISR_FUSE_SKIP', '1') == '0' or res.dtype != t.dtype or tuple(res.shape) != tuple(t.shape):
        return False
    f = _copy_struct(prep.plans[0])
    if (f.N, f.H, f.W, f.Cin) != tuple(res.shape):
        return False
    lib = L.lib()
    f.x1 = f.x2 = f.x_out = f.pa = f.pd = f.wpk = f.y = res.data_ptr()         # (non-null placeholders: eligibility only)
    f.pro_mode = L.PRO_RES_AFFINE
    f.x_bf16 = f.y_bf16 = _bf(res)
    return (lib.sisr_conv2d_trunk_eligible if prep.kinds[0] else lib.sisr_conv2d_trunk_f32_eligible)(C.byref(f)) == 1


def can_fuse_bn_backward(prep):
    """the data-gradient conv of `prep` can also emit the backward reductions of the BatchNorm its output feeds
    (generic bf16 kernel, one cout tile)"""
    d = prep.plans[1]
    if isinstance(d, S2x4):
        return True
    if isinstance(d, list):                       # stride 2: the four parity classes, all on the deep family
        return all(c is not None and c[3] == 2 for c in d)
    if d is None:
        return False
    if prep.kinds[1] == 2:                        # conv_deep.hip: any number of cout tiles
        return True
    if not prep.kinds[1]:
        # fp32 build: only the persistent trunk kernel (conv_trunk_f32.hip) has that epilogue; conv_dgrad() falls back
        # to the plain launch (and returns no partial rows) when the filled descriptor turns out not to be eligible
        gm = prep.ref.geom
        return (gm.cin == 64 and gm.cout == 64 and gm.k == 3 and gm.stride == 1 and d.H % 8 == 0 and d.W % 16 == 0
                and os.environ.get('SISR_TRUNK', '1') != '0' and os.environ.get('SISR_TRUNK_F32CONV', '1') != '0')
    return d.plan.variant == 0 and d.plan.CoutPad == d.plan.nsub * 32


def _fill_bnb(d, consts, slope):
    d.bnb_scale, d.bnb_shift, d.bnb_mean, d.bnb_invstd = (consts[0].data_ptr(), consts[1].data_ptr(),
                                                          consts[2].data_ptr(), consts[3].data_ptr())
    d.bnb_act = 0 if slope is None else 1
    if isinstance(slope, torch.Tensor):
        d.bnb_slope_p, d.bnb_slope = slope.data_ptr(), 1.0
    else:
        d.bnb_slope_p, d.bnb_slope = None, 1.0 if slope is None else float(slope)


def conv_dgrad(prep, dy_op, res=None, y_mode=L.Y_NHWC, bnb=None):
    """Data gradient: conv over the (lazy) output gradient with the flipped packed weights.
    bnb = (x, consts [4,C], slope | None): the result is the gradient arriving at BatchNorm(x) (through a leaky
    activation when slope is given); returns (out, partial rows for bn_backward_finalize) in that case."""
    lib = L.lib()
    gm = prep.ref.geom
    if isinstance(prep.plans[1], S2x4):          # stride 2, one launch over the four output-parity classes (conv_deep.hip)
        x4 = prep.plans[1]
        d = _copy_struct(x4.desc)
        assert tuple(dy_op.dims) == (d.N, d.H, d.W, d.Cin) and y_mode == L.Y_NHWC, (dy_op.dims, (d.N, d.H, d.W, d.Cin))
        dev = dy_op.x1.device
        f = prep.plans[0]
        out = torch.empty((f.N, f.H, f.W, gm.cin), dtype=act_dtype(gm.cin), device=dev)
        dy_op.fill(d)
        d.bias, d.res, d.y = None, _ptr(res), out.data_ptr()
        d.y_bf16, d.res_bf16 = _bf(out), _bf(res)
        for c, buf in enumerate(prep.wpk_dgrad):
            d.wdeep_c[c], d.deep_ckh[c] = buf.data_ptr(), x4.classes[c][0]
        ws = _attach_deep(d, prep.wpk_dgrad[3], dev, prep)
        part = None
        if bnb is not None:
            x, consts, slope = bnb
            assert tuple(x.shape) == tuple(out.shape)
            d.bnb_x, d.bnbx_bf16, d.bnb_part = x.data_ptr(), _bf(x), d.y
            rows = lib.sisr_conv2d_bf16_parts(C.byref(d))
            part = torch.empty((rows, 2 * gm.cin + 1), dtype=torch.float32, device=dev)
            d.bnb_part = part.data_ptr()
            _fill_bnb(d, consts, slope)
        L.check(lib.sisr_conv2d_bf16(C.byref(d), _stream()), 'sisr_conv2d_bf16(dgrad s2 x4)')
        return out if bnb is None else (out, part)
    if isinstance(prep.plans[1], list):          # stride 2: four output-parity classes
        f = prep.plans[0]
        dev = dy_op.x1.device
        assert y_mode == L.Y_NHWC
        complete = all(c is not None for c in prep.plans[1])
        out = (torch.empty if complete else torch.zeros)((f.N, f.H, f.W, gm.cin), dtype=act_dtype(gm.cin), device=dev)
        # fused BatchNorm-backward reductions: every class of the deep family emits the partial rows of ITS quarter of the pixels
        fuse = bnb is not None and all(c is not None and c[3] == 2 for c in prep.plans[1])
        descs = []
        for cls, buf in zip(prep.plans[1], prep.wpk_dgrad):
            if cls is None:
                continue
            d = _copy_struct(cls[0])
            assert tuple(dy_op.dims) == (d.N, d.H, d.W, d.Cin), (dy_op.dims, (d.N, d.H, d.W, d.Cin))
            dy_op.fill(d)
            d.wpk, d.bias, d.res, d.y = buf.data_ptr(), None, _ptr(res), out.data_ptr()
            d.y_bf16, d.res_bf16 = _bf(out), _bf(res)
            ws = _attach_deep(d, buf, dev, prep) if cls[3] == 2 else None
            descs.append((d, cls[3], ws))
        part = None
        if fuse:
            x, consts, slope = bnb
            assert tuple(x.shape) == tuple(out.shape)
            rows = []
            for d, _, _ in descs:
                d.bnb_x, d.bnbx_bf16, d.bnb_part = x.data_ptr(), _bf(x), d.y
                rows.append(lib.sisr_conv2d_bf16_parts(C.byref(d)))
            part = torch.empty((sum(rows), 2 * gm.cin + 1), dtype=torch.float32, device=dev)
            r0 = 0
            for (d, _, _), nr in zip(descs, rows):
                d.bnb_part = part[r0:].data_ptr()
                r0 += nr
                _fill_bnb(d, consts, slope)
        for d, kind, ws in descs:
            if kind:
                L.check(lib.sisr_conv2d_bf16(C.byref(d), _stream()), 'sisr_conv2d_bf16(dgrad s2)')
            else:
                L.check(lib.sisr_conv2d_f32(C.byref(d), _stream()), 'sisr_conv2d_f32(dgrad s2)')
        return out if bnb is None else (out, part)
    d = _copy_struct(prep.plans[1])
    assert tuple(dy_op.dims) == (d.N, d.H, d.W, d.Cin), (dy_op.dims, (d.N, d.H, d.W, d.Cin))
    dev = dy_op.x1.device
    d.y_mode = y_mode
    if y_mode == L.Y_NCHW:
        out = torch.empty((d.N, gm.cin, d.Ho, d.Wo), dtype=torch.float32, device=dev)
    else:
        out = torch.empty((d.N, d.Ho, d.Wo, gm.cin), dtype=act_dtype(gm.cin) if res is None else res.dtype, device=dev)
    dy_op.fill(d)
    d.wpk, d.bias, d.res, d.y = prep.wpk_dgrad.data_ptr(), None, _ptr(res), out.data_ptr()
    d.y_bf16, d.res_bf16 = _bf(out), _bf(res)
    d.mfma_split = mfma_split()
    ws = _attach_deep(d, prep.wpk_dgrad, dev, prep) if prep.kinds[1] == 2 else None        # (kept alive until the launch below)
    d.plan.variant = int(prep.lanes[1]) | (2 * prep.ldsimg[1])
    part = None
    if bnb is not None:
        x, consts, slope = bnb
        assert tuple(x.shape) == tuple(out.shape) and y_mode == L.Y_NHWC
        d.bnb_x, d.bnbx_bf16 = x.data_ptr(), _bf(x)
        d.bnb_part = d.y                            # (any non-null value: the row count depends on the fusions requested)
        rows = lib.sisr_conv2d_bf16_parts(C.byref(d)) if prep.kinds[1] else lib.sisr_conv2d_f32_bnb_parts(C.byref(d))
        if rows > 0:
            part = torch.empty((rows, 2 * gm.cin + 1), dtype=torch.float32, device=dev)
            d.bnb_part = part.data_ptr()
            _fill_bnb(d, consts, slope)
        else:                                        # fp32 build, descriptor not taken by the persistent kernel: no fusion
            d.bnb_x, d.bnb_part = None, None
    if prep.kinds[1]:
        L.check(lib.sisr_conv2d_bf16(C.byref(d), _stream()), 'sisr_conv2d_bf16(dgrad)')
    else:
        L.check(lib.sisr_conv2d_f32(C.byref(d), _stream()), 'sisr_conv2d_f32(dgrad)')
    return out if bnb is None else (out, part)


KERNEL_COUNTS = {}          # launches per kernel family, for the tests that must see a family run (not a timing path: host counters)


class PendingSlabs:
    """Slab reductions that have not been launched yet.  conv_wgrad(..., defer=pending) leaves the fixed-order sum of
    its per-workgroup slabs here instead of launching sisr_slab_reduce_f32; the next bn_backward(..., part=rows,
    slabs=pending) carries one of them in ITS launch (sisr_bn_bwd_finalize_slab: the two jobs are independent and
    adjacent in a residual block's backward schedule -- one launch instead of two), and flush() launches whatever is
    left the ordinary way.  The reduced buffers are valid in stream order after either."""

    def __init__(self):
        self.jobs = []                      # (slab tensor, reduced tensor, n_slabs, stride, leading bf16 elements per row)

    def pop(self):
        return self.jobs.pop(0) if self.jobs else None

    def flush(self):
        """whatever is left, in ONE launch per eight jobs (sisr_slab_reduce_multi: the jobs travel in the kernel arguments)"""
        if not self.jobs:
            return
        jobs, self.jobs = self.jobs, []
        n = len(jobs)
        slabs = (C.c_void_p * n)(*[j[0].data_ptr() for j in jobs])
        outs = (C.c_void_p * n)(*[j[1].data_ptr() for j in jobs])
        counts = (C.c_int32 * n)(*[j[2] for j in jobs])
        elems = (C.c_int64 * n)(*[j[3] for j in jobs])
        leads = (C.c_int64 * n)(*[j[4] for j in jobs])
        L.check(L.lib().sisr_slab_reduce_multi(C.addressof(slabs), C.addressof(outs), C.addressof(counts), C.addressof(elems),
                                               C.addressof(leads), n, _stream()), 'sisr_slab_reduce_multi')


def conv_wgrad(prep, x_op, dy_op, defer=None):
    """Weight + bias gradient in packed layout: returns the reduced [slab_elems + CoutPad] buffer.
    defer (PendingSlabs or None): leave the slab reduction to a later launch (see PendingSlabs)."""
    lib = L.lib()
    g = _copy_struct(prep.plans[2])
    cout = prep.ref.geom.cout
    assert tuple(x_op.dims) == (g.N, g.H, g.W, g.Cin) and tuple(dy_op.dims) == (g.N, g.Ho, g.Wo, cout), \
        (x_op.dims, dy_op.dims)
    dev = x_op.x1.device
    direct = False
    if g.Cout != cout and dy_op.mode == L.X_NCHW:
        # the generator's last conv (64 -> 3) has a kernel that reads the NCHW image gradient itself (wgrad_toimage.hip)
        x_op.fill(g)
        dy_op.fill(g, g=True)
        direct = bool(lib.sisr_wgrad_toimage_eligible(C.byref(g)))
    if g.Cout != cout and not direct:   # bf16 kernel on a channel-padded NHWC copy of the (NCHW, few-channel) gradient
        if dy_op.mode != L.X_NCHW or dy_op.pro not in (L.PRO_NONE, L.PRO_TANH_BWD):
            raise RuntimeError('padded weight gradient: NCHW gradient with no / tanh-backward prologue expected')
        g4 = torch.empty((g.N, g.Ho, g.Wo, g.Cout), dtype=torch.float32, device=dev)
        L.check(lib.sisr_nchw_grad_to_nhwc4(dy_op.x1.data_ptr(), _ptr(dy_op.x2) if dy_op.pro == L.PRO_TANH_BWD else None,
                                            g4.data_ptr(), g.N, cout, g.Ho, g.Wo, g.Cout, _stream()),
                'sisr_nchw_grad_to_nhwc4')
        dy_op = Operand.plain(g4)
    stride = g.slab_stride
    x_op.fill(g)
    dy_op.fill(g, g=True)
    g.mfma_split = mfma_split()
    n_slabs = (lib.sisr_wgrad_bf16_slabs if prep.kinds[2] else lib.sisr_wgrad_f32_slabs)(C.byref(g))
    if prep.kinds[2] and g.deep.enabled and lib.sisr_wgrad_deep_eligible(C.byref(g)):
        KERNEL_COUNTS['wgrad_deep'] = KERNEL_COUNTS.get('wgrad_deep', 0) + 1
    lead = int(lib.sisr_wgrad_bf16_slab_lead(C.byref(g))) if prep.kinds[2] else 0     # the persistent bf16 kernel's slabs are bf16
    slab = torch.empty((n_slabs, stride), dtype=torch.float32, device=dev)
    g.slab = slab.data_ptr()
    g.bias_slab = slab.data_ptr() + 4 * g.slab_elems
    if prep.kinds[2]:
        L.check(lib.sisr_conv2d_wgrad_bf16(C.byref(g), _stream()), 'sisr_conv2d_wgrad_bf16')
    else:
        L.check(lib.sisr_conv2d_wgrad_f32(C.byref(g), _stream()), 'sisr_conv2d_wgrad_f32')
    red = torch.empty((stride,), dtype=torch.float32, device=dev)
    if defer is not None and os.environ.get('SISR_FUSE_SLABRED', '1') != '0':
        defer.jobs.append((slab, red, n_slabs, stride, lead))
        return red
    L.check(lib.sisr_slab_reduce_f32(slab.data_ptr(), red.data_ptr(), n_slabs, stride, lead, _stream()),
            'sisr_slab_reduce_f32')
    return red


class WgradDeepBatch:
    """Weight gradients of the layers that run on wgrad_deep.hip, collected during a backward pass and launched TOGETHER (one launch
    per stride: grid z = layer).  Nothing consumes a weight gradient before the optimizer step, so a schedule may hold them back;
    at 96 x 96 a layer alone spreads 4 tiles per workgroup over the chip and pays ~15 us of prologue and slab stores for ~6 us of
    work, while a batch plans every member for its SHARE of the chip: 3-4 times the tiles per workgroup behind the same fixed costs,
    a third of the slabs.  add() returns the buffer the reduced gradient WILL be in -- valid after run(pending) and pending.flush()."""

    def __init__(self):
        self.items = []
        self.trunk = []                 # layers the persistent trunk kernel keeps (LR 96): batched per gradient-prologue kind

    def add(self, prep, x_op, dy_op):
        """-> reduced-gradient buffer, or None when the layer does not qualify (the caller then runs conv_wgrad as usual)"""
        if os.environ.get('SISR_WGRAD_BATCH', '1') == '0':
            return None
        if not prep.kinds[2]:
            # fp32-tensor builds: the persistent trunk kernel's plain layers (Cout = 64) are batched like the bf16 build's
            g = _copy_struct(prep.plans[2])
            x_op.fill(g)
            dy_op.fill(g, g=True)
            g.mfma_split = mfma_split()
            if (g.Cout != 64 or os.environ.get('SISR_WGRAD_TRUNK_BATCH', '1') == '0' or dy_op.mode != L.X_NHWC
                    or not L.lib().sisr_wgrad_trunk_f32_eligible(C.byref(g))):
                return None
            red = torch.empty((g.slab_stride,), dtype=torch.float32, device=x_op.x1.device)
            self.trunk.append((prep, g, x_op, dy_op, red))
            return red
        g = _copy_struct(prep.plans[2])
        if not g.deep.enabled:
            return None
        x_op.fill(g)
        dy_op.fill(g, g=True)
        lib = L.lib()
        # (the last conv's kernel comes first in sisr_conv2d_wgrad_bf16's dispatch and stays; so does the persistent trunk kernel
        # where its 8 x 16 tiles fill the chip -- at LR 48 they are 288 for 256 CUs: 144 workgroups of two, and the batch measured
        # 366 us for the 33 trunk layers against 33 x 16.5 us)
        if lib.sisr_wgrad_toimage_eligible(C.byref(g)) or not lib.sisr_wgrad_deep_eligible(C.byref(g)):
            return None
        if lib.sisr_wgrad_trunk_eligible(C.byref(g)):
            if g.N * g.H * g.W >= int(os.environ.get('SISR_WGRAD_BATCH_TRUNK_PIXELS', 384 * 128)) or os.environ.get('SISR_WGRAD_BATCH_TRUNK', '1') == '0':
                # the persistent kernel keeps the layer -- and, for the plain trunk layers (Cout = 64), its launches are batched
                # too (run(): sisr_wgrad_trunk_batch, workgroups [z * wpl, (z + 1) * wpl) serve layer z)
                if g.Cout != 64 or os.environ.get('SISR_WGRAD_TRUNK_BATCH', '1') == '0':
                    return None
                red = torch.empty((g.slab_stride,), dtype=torch.float32, device=x_op.x1.device)
                self.trunk.append((prep, g, x_op, dy_op, red))
                return red
            # (sisr_wgrad_bf16_slab_lead answers for the trunk kernel then: the same SISR_SLAB_BF16 rule as wgrad_deep.hip's)
        red = torch.empty((g.slab_stride,), dtype=torch.float32, device=x_op.x1.device)
        self.items.append((prep, g, x_op, dy_op, red))              # (the operands' tensors stay alive until run())
        return red

    def run(self, pending):
        """launch what was collected; the slab sums are left with `pending` (PendingSlabs: the caller flushes it)"""
        self._run_trunk(pending)
        if not self.items:
            return
        lib = L.lib()
        items, self.items = self.items, []
        groups = {}
        for it in items:
            g = it[1]
            groups.setdefault((g.stride, g.gpro_mode in (L.PRO_BNBWD, L.PRO_BNACT_BWD, L.PRO_ACT_BWD)), []).append(it)
        for group in groups.values():
            work = [float(g.N) * g.Ho * g.Wo * g.Cin * g.Cout for _, g, _, _, _ in group]
            tot = sum(work)
            table = (L.WgradDesc * len(group))()
            keep, first_wg = [], 0
            for i, (prep, g, x_op, dy_op, red) in enumerate(group):
                if len(group) > 1:
                    blocks = (g.Cin // 64) * (g.Cout // 64)
                    share = max(blocks, int(round(256.0 * work[i] / tot)))
                    cache, key = prep.ref.geom._plans, ('wgrad_deep_share', g.N, g.H, g.W, share, os.environ.get('SISR_SLAB_BF16', '1'))
                    if key not in cache:
                        t = _copy_struct(g)
                        L.check(lib.sisr_wgrad_deep_plan(C.byref(t), share), 'sisr_wgrad_deep_plan(share)')
                        cache[key] = _copy_struct(t.deep)
                    g.deep = cache[key]
                n_slabs = g.deep.n_pb
                g.deep.batch_first_wg = first_wg
                first_wg += g.deep.n_cib * g.deep.n_cob * n_slabs
                slab = torch.empty((n_slabs, g.slab_stride), dtype=torch.float32, device=red.device)
                g.slab = slab.data_ptr()
                g.bias_slab = slab.data_ptr() + 4 * g.slab_elems
                table[i] = g
                lead = int(lib.sisr_wgrad_bf16_slab_lead(C.byref(g)))
                pending.jobs.append((slab, red, n_slabs, g.slab_stride, lead))
                keep.append(slab)
            dev = _table_to_device(table, group[0][4].device)
            L.check(lib.sisr_wgrad_deep_batch(table, dev.data_ptr(), len(group), _stream()), 'sisr_wgrad_deep_batch')
            KERNEL_COUNTS['wgrad_deep'] = KERNEL_COUNTS.get('wgrad_deep', 0) + len(group)
            KERNEL_COUNTS['wgrad_deep_batch'] = KERNEL_COUNTS.get('wgrad_deep_batch', 0) + 1


    def _run_trunk(self, pending):
        if not self.trunk:
            return
        lib = L.lib()
        items, self.trunk = self.trunk, []
        groups = {}
        for it in items:
            groups.setdefault((bool(it[0].kinds[2]), it[1].gpro_mode), []).append(it)
        for (is_bf16, _), group in groups.items():
            n = len(group)
            f_bytes, f_args, f_run, f_lead = ((lib.sisr_wgrad_trunk_batch_arg_bytes, lib.sisr_wgrad_trunk_batch_args, lib.sisr_wgrad_trunk_batch,
                                               lambda gg: int(lib.sisr_wgrad_bf16_slab_lead(C.byref(gg)))) if is_bf16 else
                                              (lib.sisr_wgrad_trunk_f32_batch_arg_bytes, lib.sisr_wgrad_trunk_f32_batch_args,
                                               lib.sisr_wgrad_trunk_f32_batch, lambda gg: 0))
            nbytes = f_bytes()
            # 256 workgroup slots over the layers: every workgroup walks its share of ONE layer's tiles back to back (17 layers of 1,152
            # tiles: 15 workgroups x 77; alone, a layer is 231 workgroups x 5 with a tenth of the chip idle)
            wpl = max(1, min(256 // n, 231))
            table = (L.WgradDesc * n)()
            for i, (prep, g, x_op, dy_op, red) in enumerate(group):
                slab = torch.empty((wpl, g.slab_stride), dtype=torch.float32, device=red.device)
                g.slab = slab.data_ptr()
                g.bias_slab = slab.data_ptr() + 4 * g.slab_elems
                table[i] = g
                pending.jobs.append((slab, red, wpl, g.slab_stride, f_lead(g)))
            args = (C.c_char * (n * nbytes))()
            L.check(f_args(table, n, C.addressof(args)), 'sisr_wgrad_trunk(_f32)_batch_args')
            dev = _table_to_device(args, group[0][4].device)
            L.check(f_run(table, dev.data_ptr(), n, wpl, _stream()), 'sisr_wgrad_trunk(_f32)_batch')
            KERNEL_COUNTS['wgrad_trunk_batch'] = KERNEL_COUNTS.get('wgrad_trunk_batch', 0) + 1


class WeightGradBatch:
    """Collects (prepared conv, reduced packed gradient) pairs; one launch un-packs them all."""

    def __init__(self):
        self.items = []

    def add(self, prep, red, want_w=True, want_b=True):
        self.items.append((prep, red, want_w, want_b))

    def run(self):
        """returns {id(ConvRef): (grad_w or None, grad_b or None)}"""
        if not self.items:
            return {}
        # 3x3 layers whose gradient slabs come from the bf16 kernels (layout 1) with channels in 32s take the whole-tile
        # un-packing (sisr_weights_grad_fast); the rest (9x9 / 3-channel / fp32-slab layers) the generic pair
        fast, slow = [], []
        for it in self.items:
            gm, g = it[0].ref.geom, it[0].plans[2]
            ok = (it[0].kinds[2] and gm.k == 3 and gm.cin % 32 == 0 and gm.cout % 32 == 0 and not gm.shuffle2
                  and g.CoutPad >= gm.cout and os.environ.get('SISR_WGRAD_FAST', '1') != '0')
            (fast if ok else slow).append(it)
        res = {}
        self._keep = []
        for group, is_fast in ((fast, True), (slow, False)):
            if group:
                res.update(self._run_group(group, is_fast))
        return res

    def _run_group(self, items, is_fast):
        lib = L.lib()
        table = (L.WeightGradDesc * len(items))()
        res = {}
        dev = items[0][1].device
        for i, (p, red, want_w, want_b) in enumerate(items):
            g = p.plans[2]
            gm = p.ref.geom
            t = table[i]
            gw = torch.empty_like(p.ref.weight) if want_w else None
            gb = torch.empty_like(p.ref.bias) if (want_b and p.ref.bias is not None) else None
            t.dwpk, t.w_orig, t.grad = red.data_ptr(), p.ref.weight.data_ptr(), _ptr(gw)
            t.u_used, t.v_used, t.sigma = _ptr(p.u_used), _ptr(p.v_used), p.sigma.data_ptr()
            t.dbias_pk = red.data_ptr() + 4 * g.slab_elems
            t.grad_bias = _ptr(gb)
            t.Cout, t.Cin, t.KH, t.KW, t.shuffle2 = gm.cout, gm.cin, gm.k, gm.k, int(gm.shuffle2)
            t.CK, t.PS, t.KROWP, t.n_chunk, t.CoutPad = g.CK, g.PS, g.KROWP, g.n_chunk, g.CoutPad
            t.layout = 1 if p.kinds[2] else 0
            res[id(p.ref)] = (gw, gb)
        tab = _table_to_device(table, dev)
        if is_fast:
            mco, mci = max(it[0].ref.geom.cout for it in items), max(it[0].ref.geom.cin for it in items)
            work = torch.empty((len(items) * ((mco + 31) // 32) * (mci // 32),), dtype=torch.float32, device=dev)
            L.check(lib.sisr_weights_grad_fast(tab.data_ptr(), len(items), work.data_ptr(), mco, mci, _stream()), 'sisr_weights_grad_fast')
        else:
            parts = max(L.check_count(lib.sisr_weights_grad_tiles(C.byref(t)), 'sisr_weights_grad_tiles') for t in table)
            work = torch.empty((parts * len(items),), dtype=torch.float32, device=dev)
            L.check(lib.sisr_weights_grad(tab.data_ptr(), len(items), work.data_ptr(), parts, _stream()), 'sisr_weights_grad')
        self._keep.append((tab, work))
        return res


class LazyBN:
    """BatchNorm constants [4, C] (scale, shift, batch mean, invstd) that are not computed yet: the statistics rows of
    the producing conv plus the module.  Where the conv that APPLIES this BatchNorm in its prologue runs on a persistent
    trunk kernel, that kernel finalises the statistics itself (SisrConvDesc.fin_*: one launch less per BatchNorm);
    anything else calls ensure(), which runs the stand-alone sisr_bn_finalize.  Either way `k` is valid in stream
    order after the consumer (or ensure()) has been launched."""
    __slots__ = ('sp', 'cp', 'bn', 'eps', 'momentum', 'k', 'done')

    def __init__(self, sp, cp, bn, eps=1e-5, momentum=0.1):
        self.sp, self.cp, self.bn, self.eps, self.momentum = sp, cp, bn, eps, momentum
        self.k = torch.empty((4, bn.weight.numel()), dtype=torch.float32, device=sp.device)
        self.done = False

    def ensure(self):
        if not self.done:
            lib = L.lib()
            bn, k = self.bn, self.k
            L.check(lib.sisr_bn_finalize(self.sp.data_ptr(), self.cp.data_ptr(), self.sp.shape[0], k.shape[1],
                                         bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                         bn.running_var.data_ptr(), self.momentum, self.eps, k[0].data_ptr(),
                                         k[1].data_ptr(), k[2].data_ptr(), k[3].data_ptr(), _stream()), 'sisr_bn_finalize')
            self.done = True
        return self.k

    def fill(self, d):
        """hand the finalisation to the conv of descriptor d"""
        bn = self.bn
        d.fin_stat, d.fin_cnt, d.fin_rows = self.sp.data_ptr(), self.cp.data_ptr(), self.sp.shape[0]
        d.fin_gamma, d.fin_beta = bn.weight.data_ptr(), bn.bias.data_ptr()
        d.fin_rm, d.fin_rv, d.fin_k = bn.running_mean.data_ptr(), bn.running_var.data_ptr(), self.k.data_ptr()
        d.fin_momentum, d.fin_eps = self.momentum, self.eps


def bn_finalize(sp, cp, bn, eps=1e-5, momentum=0.1):
    """-> consts [4, C]: scale, shift, batch mean, invstd; updates bn.running_* in place."""
    lib = L.lib()
    cch = bn.weight.numel()
    k = torch.empty((4, cch), dtype=torch.float32, device=sp.device)
    L.check(lib.sisr_bn_finalize(sp.data_ptr(), cp.data_ptr(), sp.shape[0], cch, bn.weight.data_ptr(),
                                 bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                 momentum, eps, k[0].data_ptr(), k[1].data_ptr(), k[2].data_ptr(),
                                 k[3].data_ptr(), _stream()), 'sisr_bn_finalize')
    return k


def bn_eval_consts(bn, eps=1e-5):
    lib = L.lib()
    cch = bn.weight.numel()
    k = torch.empty((4, cch), dtype=torch.float32, device=bn.weight.device)
    L.check(lib.sisr_bn_eval_consts(bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                    bn.running_var.data_ptr(), eps, cch, k[0].data_ptr(), k[1].data_ptr(),
                                    _stream()), 'sisr_bn_eval_consts')
    return k


def bn_backward(dy, x, consts, gamma, slope=None, part=None, slabs=None):
    """Reductions of BatchNorm backward (+ the leaky activation after it when slope is given).
    Returns (q [3,C] = qa,qb,qd ; dgamma ; dbeta ; dslope or None).  part: per-tile partial rows already written
    by the conv that produced dy (conv_dgrad(..., bnb=...)); only the finishing kernel runs then -- and, given `slabs`
    (PendingSlabs), that launch also carries one deferred slab reduction."""
    lib = L.lib()
    cch = x.shape[-1]
    d = L.BnBwdDesc()
    d.P, d.C = x.numel() // cch, cch
    d.act_mode = 0 if slope is None else 1
    if isinstance(slope, torch.Tensor):
        d.slope_p, d.slope = slope.data_ptr(), 1.0
    else:
        d.slope_p, d.slope = None, 1.0 if slope is None else float(slope)
    dev = x.device
    if part is None:
        L.check(lib.sisr_bn_bwd_plan(C.byref(d)), 'sisr_bn_bwd_plan')
        work = torch.empty((d.grid, 2 * cch + 1), dtype=torch.float32, device=dev)
    else:
        work, d.grid = part, part.shape[0]
    q = torch.empty((3, cch), dtype=torch.float32, device=dev)
    dgamma = torch.empty((cch,), dtype=torch.float32, device=dev)
    dbeta = torch.empty((cch,), dtype=torch.float32, device=dev)
    dslope = torch.empty((1,), dtype=torch.float32, device=dev) if slope is not None else None
    d.dy, d.x = dy.data_ptr(), x.data_ptr()
    d.dy_bf16, d.x_bf16 = _bf(dy), _bf(x)
    d.scale, d.shift, d.mean, d.invstd = (consts[0].data_ptr(), consts[1].data_ptr(), consts[2].data_ptr(),
                                          consts[3].data_ptr())
    d.gamma, d.work = gamma.data_ptr(), work.data_ptr()
    d.qa, d.qb, d.qd = q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr()
    d.dgamma, d.dbeta, d.dslope = dgamma.data_ptr(), dbeta.data_ptr(), _ptr(dslope)
    if part is None:
        L.check(lib.sisr_bn_bwd(C.byref(d), _stream()), 'sisr_bn_bwd')
    else:
        job = slabs.pop() if slabs is not None else None
        if job is not None:                     # this launch also sums the slabs of the weight gradient computed before it
            slab, red, n_slabs, stride, lead = job
            L.check(lib.sisr_bn_bwd_finalize_slab(C.byref(d), slab.data_ptr(), red.data_ptr(), n_slabs, stride, lead, _stream()),
                    'sisr_bn_bwd_finalize_slab')
        else:
            L.check(lib.sisr_bn_bwd_finalize(C.byref(d), _stream()), 'sisr_bn_bwd_finalize')
    return q, dgamma, dbeta, dslope


def eltwise_res_affine(x1, slope1, x2=None, pa=None, pd=None):
    """y = lrelu(x1, slope1) + (pa*x2 + pd | x2 | 0) over NHWC tensors."""
    lib = L.lib()
    cch = x1.shape[-1]
    y = torch.empty(x1.shape, dtype=act_dtype(cch), device=x1.device)
    sp, sv = (slope1.data_ptr(), 1.0) if isinstance(slope1, torch.Tensor) else \
        (None, 1.0 if slope1 is None else float(slope1))
    dt = _bf(x1) | (_bf(x2) << 1) | (_bf(y) << 2)
    L.check(lib.sisr_eltwise_res_affine(x1.data_ptr(), sp, sv, _ptr(x2), _ptr(pa), _ptr(pd), y.data_ptr(),
                                        x1.numel() // cch, cch, dt, _stream()), 'sisr_eltwise_res_affine')
    return y


def prelu_slope_grad(dy, pre):
    lib = L.lib()
    work = torch.empty((1024,), dtype=torch.float32, device=dy.device)
    out = torch.empty((1,), dtype=torch.float32, device=dy.device)
    L.check(lib.sisr_prelu_slope_grad(dy.data_ptr(), pre.data_ptr(), dy.numel(), work.data_ptr(),
                                      out.data_ptr(), _bf(dy) | (_bf(pre) << 1), _stream()), 'sisr_prelu_slope_grad')
    return out


def add(a, b):
    lib = L.lib()
    if a.dtype != b.dtype:
        raise RuntimeError('add: operands with mixed storage types')
    y = torch.empty_like(a)
    L.check(lib.sisr_add(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), 7 * _bf(a), _stream()), 'sisr_add')
    return y


def require_gpu_tensor(x, what):
    if not (isinstance(x, torch.Tensor) and x.is_cuda):
        raise RuntimeError('%s: this path runs only on an MI355X device tensor (got %s); there is no '
                           'CPU fallback' % (what, getattr(x, 'device', type(x))))
    if x.dtype != torch.float32:
        raise RuntimeError('%s: fp32 tensors expected at the module boundary, got %s' % (what, x.dtype))


def nhwc_to_nchw(x, out, dst_stride, pa=None, pd=None, slope=None):
    """materialise lrelu(pa*x+pd, slope) from NHWC x into an NCHW destination (rows of `out`)."""
    n, h, w, c = x.shape
    sp, sv = (slope.data_ptr(), 1.0) if isinstance(slope, torch.Tensor) else (None, 1.0 if slope is None else float(slope))
    L.check(L.lib().sisr_nhwc_to_nchw(x.data_ptr(), _ptr(pa), _ptr(pd), sp, sv, out.data_ptr(), dst_stride,
                                      n, h, w, c, _bf(x), _stream()), 'sisr_nhwc_to_nchw')


def nchw_to_nhwc(src, src_stride, n, h, w, c):
    y = torch.empty((n, h, w, c), dtype=act_dtype(c), device=src.device)
    L.check(L.lib().sisr_nchw_to_nhwc(src.data_ptr(), src_stride, y.data_ptr(), n, h, w, c, _bf(y), _stream()),
            'sisr_nchw_to_nhwc')
    return y


FC_MAX_BATCH = 16          # rows the FC kernels keep in registers (FC_B in layout_fc.hip); larger batches run in slices


def fc_forward(x, w, b, in_slope=1.0, sigmoid=False):
    bsz, k = x.shape
    y = torch.empty((bsz, w.shape[0]), dtype=torch.float32, device=x.device)
    for b0 in range(0, bsz, FC_MAX_BATCH):
        nb = min(FC_MAX_BATCH, bsz - b0)
        L.check(L.lib().sisr_fc_forward(x[b0:b0 + nb].data_ptr(), in_slope, w.data_ptr(), _ptr(b), y[b0:b0 + nb].data_ptr(),
                                        nb, k, w.shape[0], int(sigmoid), _stream()), 'sisr_fc_forward')
    return y


def fc_backward(dy, x, w, in_slope=1.0, need_dx=True):
    """-> (dx wrt lrelu(x) [B,K] or None, dW, db)"""
    lib = L.lib()
    bsz, k = x.shape
    nout = w.shape[0]
    dw = db = None
    dx = torch.empty((bsz, k), dtype=torch.float32, device=x.device) if need_dx else None
    for b0 in range(0, bsz, FC_MAX_BATCH):
        nb = min(FC_MAX_BATCH, bsz - b0)
        dyc, xc = dy[b0:b0 + nb], x[b0:b0 + nb]
        dwc = torch.empty_like(w)
        dbc = torch.empty((nout,), dtype=torch.float32, device=x.device)
        L.check(lib.sisr_fc_wgrad(dyc.data_ptr(), xc.data_ptr(), in_slope, dwc.data_ptr(), dbc.data_ptr(), nb, k, nout,
                                  _stream()), 'sisr_fc_wgrad')
        dw, db = (dwc, dbc) if dw is None else (add(dw, dwc), add(db, dbc))     # batch slices sum into the gradient
        if need_dx:
            splits = lib.sisr_fc_dgrad_splits(k, nout)
            work = torch.empty((splits, nb, k), dtype=torch.float32, device=x.device)
            L.check(lib.sisr_fc_dgrad(dyc.data_ptr(), w.data_ptr(), dx[b0:b0 + nb].data_ptr(), work.data_ptr(), nb, k, nout,
                                      _stream()), 'sisr_fc_dgrad')
    return dx, dw, db


def fc_head_ok(bsz, k, n):
    """the classifier head of D runs on fc_head.hip (exact-fp32 MFMA weight streaming): up to 16 batch rows, K in 64s, N in 128s"""
    return bsz <= FC_MAX_BATCH and k % 64 == 0 and n % 128 == 0 and os.environ.get('SISR_FC_HEAD', '1') != '0'


def fc_head_forward(x, w1, b1, w2, b2, slope):
    """-> (h1 [B, N] pre-activation, y [B, 1]) of Linear -> LeakyReLU -> Linear(N, 1) -> Sigmoid"""
    lib = L.lib()
    bsz, k = x.shape
    n = w1.shape[0]
    h1 = torch.empty((bsz, n), dtype=torch.float32, device=x.device)
    y = torch.empty((bsz, 1), dtype=torch.float32, device=x.device)
    ws = torch.empty((lib.sisr_fc_head_ws_floats(n),), dtype=torch.float32, device=x.device)
    L.check(lib.sisr_fc_head_forward(x.data_ptr(), w1.data_ptr(), _ptr(b1), w2.data_ptr(), _ptr(b2), slope, h1.data_ptr(),
                                     y.data_ptr(), ws.data_ptr(), bsz, k, n, _stream()), 'sisr_fc_head_forward')
    return h1, y


def fc_head_backward(g, y, h1, w2, slope):
    """-> (d1 [B, N], dW2 like w2, db2 [1], db1 [N])"""
    lib = L.lib()
    bsz, n = h1.shape
    dev = h1.device
    d1 = torch.empty_like(h1)
    dw2 = torch.empty_like(w2)
    db2 = torch.empty((1,), dtype=torch.float32, device=dev)
    db1 = torch.empty((n,), dtype=torch.float32, device=dev)
    L.check(lib.sisr_fc_head_backward(g.data_ptr(), y.data_ptr(), h1.data_ptr(), w2.data_ptr(), slope, d1.data_ptr(),
                                      dw2.data_ptr(), db2.data_ptr(), db1.data_ptr(), bsz, n, _stream()), 'sisr_fc_head_backward')
    return d1, dw2, db2, db1


def fc1_dgrad(d1, w1):
    bsz, n = d1.shape
    k = w1.shape[1]
    dx = torch.empty((bsz, k), dtype=torch.float32, device=d1.device)
    L.check(L.lib().sisr_fc1_dgrad(d1.data_ptr(), w1.data_ptr(), dx.data_ptr(), bsz, k, n, _stream()), 'sisr_fc1_dgrad')
    return dx


def fc_wgrad_rows_ok(rows, k, n):
    """sisr_fc_wgrad_rows takes the gathered factors of `rows` batch rows (all ranks)"""
    return rows <= 256 and k % 128 == 0 and n % 64 == 0


def fc_wgrad_rows(dy_all, x_all, w, scale):
    """dW = scale * dy_all^T x_all over the rows of ALL ranks (the gathered factors of the classifier head's weight gradient)"""
    rows, k = x_all.shape
    dw = torch.empty_like(w)
    L.check(L.lib().sisr_fc_wgrad_rows(dy_all.data_ptr(), x_all.data_ptr(), float(scale), dw.data_ptr(), rows, k, w.shape[0], _stream()),
            'sisr_fc_wgrad_rows')
    KERNEL_COUNTS['fc_wgrad_rows'] = KERNEL_COUNTS.get('fc_wgrad_rows', 0) + 1
    return dw


def fc_wgrad_only(dy, x, w, in_slope=1.0):
    """dW = dy^T lrelu(x) (no bias gradient, no data gradient)"""
    bsz, k = x.shape
    dw = torch.empty_like(w)
    L.check(L.lib().sisr_fc_wgrad(dy.data_ptr(), x.data_ptr(), in_slope, dw.data_ptr(), None, bsz, k, w.shape[0], _stream()),
            'sisr_fc_wgrad')
    return dw


def act_bwd(dy, ref, kind, slope=0.0):
    out = torch.empty_like(dy)
    L.check(L.lib().sisr_act_bwd(dy.data_ptr(), ref.data_ptr(), out.data_ptr(), dy.numel(), kind, slope,
                                 _stream()), 'sisr_act_bwd')
    return out


def maxpool2(x):
    n, h, w, c = x.shape
    y = torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
    L.check(L.lib().sisr_maxpool2_fwd(x.data_ptr(), y.data_ptr(), n, h, w, c, 3 * _bf(x), _stream()), 'sisr_maxpool2_fwd')
    return y


def maxpool2_relu_bwd(dy, x):
    n, h, w, c = x.shape
    if dy.dtype != x.dtype:
        raise RuntimeError('maxpool2_relu_bwd: gradient and activation with mixed storage types')
    dx = torch.empty_like(x)
    L.check(L.lib().sisr_maxpool2_relu_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), n, h, w, c, 7 * _bf(x), _stream()),
            'sisr_maxpool2_relu_bwd')
    return dx


def add_relu_masked(a, b, ref):
    out = torch.empty_like(b)
    dt = _bf(a) | (_bf(b) << 1) | (_bf(ref) << 2) | (_bf(out) << 3)
    L.check(L.lib().sisr_add_relu_masked(_ptr(a), b.data_ptr(), ref.data_ptr(), out.data_ptr(), b.numel(), dt,
                                         _stream()), 'sisr_add_relu_masked')
    return out
