"""CPU: the C-ABI library loads and exports every symbol include/sisr_hip.h declares; the host
planners (pure host code) produce valid plans; the product refuses CPU tensors (no fallback)."""
import ctypes
import importlib
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pkg(sub):
    return importlib.import_module('single-image-super-resolution_amd.' + sub)


def test_library_exports_every_declared_symbol():
    L = _pkg('_lib')
    if not os.path.exists(L.LIB_PATH):
        L.build()
    hdr = open(os.path.join(ROOT, 'include', 'sisr_hip.h')).read()
    declared = set(re.findall(r'\b(sisr_[a-z0-9_]+)\s*\(', hdr))
    lib = ctypes.CDLL(L.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    assert L.lib().sisr_version().startswith(b'sisr_hip')          # also checks the struct mirror


@pytest.mark.parametrize('shape', [(16, 64, 64, 3, 1, 96, 96), (16, 3, 64, 9, 1, 96, 96), (16, 64, 256, 3, 1, 96, 96),
                                   (16, 64, 3, 3, 1, 192, 192), (16, 256, 256, 3, 2, 24, 24), (2, 1, 3, 3, 1, 5, 7)])
def test_planners_fit_lds_and_cover_the_problem(shape):
    E = _pkg('engine')
    n, cin, cout, k, s, h, w = shape
    f, d, g, kinds = E.ConvGeom(cin, cout, k, s, k // 2).plans(n, h, w)
    assert kinds == (False, False, False)          # default precision: the exact-fp32 kernels
    dgrad = [c[0] for c in d if c is not None] if isinstance(d, list) else ([d] if d is not None else [])
    if s == 2:      # stride-2 data gradient: four output-parity classes that tile the input exactly
        assert len(dgrad) == 4 and sum(c.Ho * c.Wo for c in dgrad) == h * w
        assert sorted((c.KH, c.KW) for c in dgrad) == [(1, 1), (1, 2), (2, 1), (2, 2)]
    for pl, (ho, wo) in [(f, (f.Ho, f.Wo))] + [(c, (c.Ho, c.Wo)) for c in dgrad]:
        p = pl.plan
        assert 0 < p.lds_bytes <= 160 * 1024
        assert p.tiles_y * p.TH >= ho and p.tiles_x * p.TW >= wo and p.n_groups * p.TN >= n
        assert p.TN * p.TH * p.TW <= 4 * p.msub * 32
        assert p.PS % 2 == 1 and p.KROWP % 4 == 0 and p.KROWP >= pl.KW * p.PS
        assert p.n_chunk * p.CK >= pl.Cin and p.CoutPad % (32 * p.nsub) == 0
    assert 0 < g.lds_bytes <= 160 * 1024 and g.KH * g.NT <= 9 and g.NJ * g.NP == 4
    assert g.tiles_y * g.TH >= g.Ho and g.tiles_x * g.TW >= g.Wo and 1 <= g.grid_x <= g.n_tiles


def test_cpu_tensors_are_refused():
    mg, ut = _pkg('model_generator'), _pkg('utils')
    with pytest.raises(RuntimeError):
        mg.Generator(1, 16, 64, [2])(torch.zeros(1, 3, 8, 8))
    with pytest.raises(RuntimeError):
        ut.lr_from_hr(torch.zeros(1, 3, 8, 8), (4, 4))


def test_modules_mirror_reference_state_dict_layout():
    """key names / order of the reference's state_dict (captured in the golden fixtures)."""
    import numpy as np
    mg, mp_ = _pkg('model_generator'), _pkg('model_generator_progressive')
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'gen_x4_suffix_w32.npz'))
    keys = [k[6:] for k in z.files if k.startswith('state/')]
    g = mg.GeneratorSuffix(mg.Generator(1, 32, 128, [2], use_sn=True))
    assert list(g.state_dict().keys()) == keys
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'prog_x8_w64.npz'))
    keys = [k[6:] for k in z.files if k.startswith('state/')]
    g1 = mp_.GeneratorSuffix(mp_.GeneratorProgresiveBase(1, 64), 64)
    g3 = mp_.GeneratorSuffix(mp_.GeneratorSuffix(g1.beginning, 16).beginning, 4)
    assert list(g3.state_dict().keys()) == keys


def test_shape_specific_kernels_claim_exactly_their_descriptors(monkeypatch):
    """Host-side dispatch rules of the thin-layer kernels (pure host code, no GPU): the generator's first / last conv
    descriptors are claimed at the bench sizes, and ragged sizes, other prologues and SISR_THIN=0 fall back to the
    generic kernels (model_generator.py:32, 52)."""
    E, L = _pkg('engine'), _pkg('_lib')
    lib = L.lib()
    monkeypatch.delenv('SISR_THIN', raising=False)
    # first conv (9x9, 3 -> 64): fp32 planner descriptors; the bf16 build stores its output as bf16
    f, _, g, _ = E.ConvGeom(3, 64, 9, 1, 4).plans(16, 96, 96)
    f = L.ConvDesc.from_buffer_copy(f)
    f.x_mode, f.y_mode, f.y_bf16 = L.X_NCHW, L.Y_NHWC, 1
    assert lib.sisr_conv2d_thin_eligible(f) == 1
    f.y_bf16 = 0
    assert lib.sisr_conv2d_thin_eligible(f) == 0             # fp32 output: the parity build keeps the exact-fp32 kernel
    f.y_bf16, f.pro_mode = 1, L.PRO_ACT
    assert lib.sisr_conv2d_thin_eligible(f) == 0
    g = L.WgradDesc.from_buffer_copy(g)
    g.x_mode, g.g_mode, g.g_bf16, g.gpro_mode = L.X_NCHW, L.X_NHWC, 1, L.PRO_ACT_BWD
    assert lib.sisr_wgrad_thin_eligible(g) == 1
    assert lib.sisr_wgrad_f32_slabs(g) <= 256 * 2            # one slab per workgroup of the persistent kernel
    g48 = L.WgradDesc.from_buffer_copy(E.ConvGeom(3, 64, 9, 1, 4).plans(16, 48, 48)[2])
    g48.x_mode, g48.g_mode, g48.g_bf16, g48.gpro_mode = L.X_NCHW, L.X_NHWC, 1, L.PRO_ACT_BWD
    assert lib.sisr_wgrad_thin_eligible(g48) == 0            # W % 32 != 0
    # last conv (3x3, 64 -> 3) with fp32 tensors: forward and weight gradient
    f3, _, g3, _ = E.ConvGeom(64, 3, 3, 1, 1).plans(16, 192, 192)
    f3 = L.ConvDesc.from_buffer_copy(f3)
    f3.x_mode, f3.y_mode, f3.pro_mode, f3.epi_act = L.X_NHWC, L.Y_NCHW, L.PRO_ACT, L.EPI_TANH
    assert lib.sisr_conv2d_toimage_f32_eligible(f3) == 1 and lib.sisr_conv2d_toimage_eligible(f3) == 0
    f3.y_mode = L.Y_NHWC
    assert lib.sisr_conv2d_toimage_f32_eligible(f3) == 0
    g3 = L.WgradDesc.from_buffer_copy(g3)
    g3.x_mode, g3.pro_mode, g3.g_mode, g3.gpro_mode = L.X_NHWC, L.PRO_ACT, L.X_NCHW, L.PRO_TANH_BWD
    assert lib.sisr_wgrad_toimage_f32_eligible(g3) == 1 and lib.sisr_wgrad_toimage_eligible(g3) == 0
    monkeypatch.setenv('SISR_THIN', '0')
    f.pro_mode, f3.y_mode = L.PRO_NONE, L.Y_NCHW
    assert lib.sisr_conv2d_thin_eligible(f) == 0 and lib.sisr_wgrad_thin_eligible(g) == 0
    assert lib.sisr_conv2d_toimage_f32_eligible(f3) == 0 and lib.sisr_wgrad_toimage_f32_eligible(g3) == 0


def test_fp32_trunk_kernels_claim_the_upscale_conv_only_as_they_can_run_it(monkeypatch):
    """Host-side dispatch rules of the fp32-tensor persistent kernels for the generator's upscale conv (model_generator.py:43-48):
    forward (Cout 256 stored through PixelShuffle(2)) without the fusions that variant has no code for; data gradient (Cin 256
    read through the un-shuffling view) only with the activation-backward prologue and both of its tensors; weight gradient only
    with the shuffled gradient view + activation-backward prologue.  SisrConvDesc.mfma_split never changes who takes a descriptor."""
    E, L = _pkg('engine'), _pkg('_lib')
    lib = L.lib()
    for v in ('SISR_TRUNK', 'SISR_TRUNK_UP', 'SISR_TRUNK_F32CONV', 'SISR_TRUNK_WGRAD', 'SISR_PERSIST_MAX_WG'):
        monkeypatch.delenv(v, raising=False)
    E.set_precision('fp32')
    one = 1
    f, d, g, kinds = E.ConvGeom(64, 256, 3, 1, 1, shuffle2=True).plans(16, 96, 96)
    assert kinds == (False, False, False)
    u = L.ConvDesc.from_buffer_copy(f)
    u.x1 = u.wpk = u.y = one
    assert u.y_mode == L.Y_SHUFFLE2 and lib.sisr_conv2d_trunk_f32_eligible(u) == 1
    for split in (0, 1):
        u.mfma_split = split
        assert lib.sisr_conv2d_trunk_f32_eligible(u) == 1
    for field in ('stat_part', 'res', 'bnb_part', 'fin_stat'):
        v = L.ConvDesc.from_buffer_copy(u)
        setattr(v, field, one)
        assert lib.sisr_conv2d_trunk_f32_eligible(v) == 0, field
    v = L.ConvDesc.from_buffer_copy(u)
    v.y_mode = L.Y_NHWC
    assert lib.sisr_conv2d_trunk_f32_eligible(v) == 0              # Cout 256 WITHOUT the shuffled store
    v = L.ConvDesc.from_buffer_copy(u)
    v.pro_mode = L.PRO_RES_AFFINE
    v.x2 = v.x_out = v.pa = v.pd = one
    assert lib.sisr_conv2d_trunk_f32_eligible(v) == 0
    # ---- data gradient: 256 -> 64 over the un-shuffling view
    dd = L.ConvDesc.from_buffer_copy(d)
    assert (dd.Cin, dd.Cout) == (256, 64)
    dd.x1 = dd.x2 = dd.wpk = dd.y = one
    dd.x_mode, dd.pro_mode = L.X_UNSHUFFLE2, L.PRO_ACT_BWD
    assert lib.sisr_conv2d_trunk_f32_eligible(dd) == 2
    dd.res = one
    assert lib.sisr_conv2d_trunk_f32_eligible(dd) == 2             # the skip gradient rides on the first phase's launch
    for field, val in (('x2', None), ('pro_mode', L.PRO_NONE), ('x_mode', L.X_NHWC), ('bias', one), ('bnb_part', one), ('stat_part', one)):
        v = L.ConvDesc.from_buffer_copy(dd)
        setattr(v, field, val)
        assert lib.sisr_conv2d_trunk_f32_eligible(v) == 0, field
    # ---- weight gradient: the gradient read through the strided view of each PixelShuffle phase
    w = L.WgradDesc.from_buffer_copy(g)
    w.x1 = w.g1 = w.g2 = w.slab = one
    w.g_mode, w.gpro_mode = L.X_UNSHUFFLE2, L.PRO_ACT_BWD
    assert lib.sisr_wgrad_trunk_f32_eligible(w) == 1 and lib.sisr_wgrad_f32_slabs(w) <= 64      # one slab per tile stream (4 workgroups)
    for field, val in (('g_mode', L.X_NHWC), ('gpro_mode', L.PRO_BNBWD)):
        v = L.WgradDesc.from_buffer_copy(w)
        setattr(v, field, val)
        assert lib.sisr_wgrad_trunk_f32_eligible(v) == 0, field
    monkeypatch.setenv('SISR_TRUNK_UP', '0')
    assert lib.sisr_conv2d_trunk_f32_eligible(u) == 0 and lib.sisr_conv2d_trunk_f32_eligible(dd) == 0
    assert lib.sisr_wgrad_trunk_f32_eligible(w) == 0


def test_trunk_kernels_claim_exactly_their_descriptors(monkeypatch):
    """Host-side dispatch rules of the persistent trunk kernels (pure host code): sisr_conv2d_trunk_eligible returns
    1 (forward role) / 2 (data-gradient role) only for 3x3 64 -> 64 bf16 NHWC descriptors on 8 x 16 tile grids, and for
    the UPSCALE variant (Cout 256 stored through PixelShuffle(2), model_generator.py:45-46) only without the fusions that
    variant has no code for -- statistics, residual, fused BatchNorm reductions, skip-sum prologue, deferred BatchNorm
    finalisation: each of those must send the descriptor to the generic kernel, not into a kernel that would ignore it
    (and the statistics row count sisr_conv2d_bf16_parts must follow the kernel that actually runs: ADVICE r2)."""
    E, L = _pkg('engine'), _pkg('_lib')
    lib = L.lib()
    for v in ('SISR_TRUNK', 'SISR_TRUNK_UP', 'SISR_PERSIST_MAX_WG'):
        monkeypatch.delenv(v, raising=False)
    E.set_precision('bf16')
    try:
        one = 1                                                    # (any non-null pointer value: eligibility only)
        f, d, _, kinds = E.ConvGeom(64, 64, 3, 1, 1).plans(16, 96, 96)
        assert kinds[0] and kinds[1]
        f = L.ConvDesc.from_buffer_copy(f)
        f.x1 = f.wpk = f.y = one
        f.x_bf16 = f.y_bf16 = 1
        assert lib.sisr_conv2d_trunk_eligible(f) == 1
        f.stat_part = f.cnt_part = one
        assert lib.sisr_conv2d_trunk_eligible(f) == 1 and lib.sisr_conv2d_bf16_parts(f) <= 256     # one row per workgroup
        f.y_bf16 = 0
        assert lib.sisr_conv2d_trunk_eligible(f) == 0 and lib.sisr_conv2d_bf16_parts(f) == f.plan.n_tiles
        f.y_bf16, f.epi_act = 1, L.EPI_TANH
        assert lib.sisr_conv2d_trunk_eligible(f) == 0
        f.epi_act, f.pro_mode = 0, L.PRO_RES_AFFINE
        assert lib.sisr_conv2d_trunk_eligible(f) == 0              # skip-sum prologue without its operands
        f.x2 = f.x_out = f.pa = f.pd = one
        assert lib.sisr_conv2d_trunk_eligible(f) == 1
        dd = L.ConvDesc.from_buffer_copy(d)
        dd.x1 = dd.x2 = dd.wpk = dd.y = dd.pa = dd.pb = dd.pd = one
        dd.x_bf16 = dd.y_bf16 = 1
        dd.pro_mode = L.PRO_BNBWD
        assert lib.sisr_conv2d_trunk_eligible(dd) == 2
        dd.bias = one
        assert lib.sisr_conv2d_trunk_eligible(dd) == 0             # the data-gradient role has no bias
        ragged = L.ConvDesc.from_buffer_copy(E.ConvGeom(64, 64, 3, 1, 1).plans(2, 20, 48)[0])
        ragged.x1 = ragged.wpk = ragged.y = one
        ragged.x_bf16 = ragged.y_bf16 = 1
        assert lib.sisr_conv2d_trunk_eligible(ragged) == 0         # H % 8 != 0
        # ---- the upscale variant
        u = L.ConvDesc.from_buffer_copy(E.ConvGeom(64, 256, 3, 1, 1, shuffle2=True).plans(16, 96, 96)[0])
        u.x1 = u.wpk = u.y = one
        u.x_bf16 = u.y_bf16 = 1
        assert u.y_mode == L.Y_SHUFFLE2 and lib.sisr_conv2d_trunk_eligible(u) == 1
        for field in ('stat_part', 'res', 'bnb_part', 'fin_stat'):
            v = L.ConvDesc.from_buffer_copy(u)
            setattr(v, field, one)
            assert lib.sisr_conv2d_trunk_eligible(v) == 0, field
            if field == 'stat_part':                               # ... and its statistics are sized for the generic kernel
                v.cnt_part = one
                assert lib.sisr_conv2d_bf16_parts(v) == v.plan.n_tiles
        v = L.ConvDesc.from_buffer_copy(u)
        v.y_mode = L.Y_NHWC
        assert lib.sisr_conv2d_trunk_eligible(v) == 0              # Cout 256 WITHOUT the shuffled store
        v = L.ConvDesc.from_buffer_copy(u)
        v.pro_mode = L.PRO_RES_AFFINE
        v.x2 = v.x_out = v.pa = v.pd = one
        assert lib.sisr_conv2d_trunk_eligible(v) == 0
        v = L.ConvDesc.from_buffer_copy(u)
        v.pro_mode = L.PRO_BNBWD                                   # a data-gradient prologue on the 256-cout forward variant
        v.x2 = v.pa = v.pb = v.pd = one
        assert lib.sisr_conv2d_trunk_eligible(v) == 0
        monkeypatch.setenv('SISR_TRUNK_UP', '0')
        assert lib.sisr_conv2d_trunk_eligible(u) == 0 and lib.sisr_conv2d_trunk_eligible(f) == 1
        monkeypatch.setenv('SISR_TRUNK', '0')
        assert lib.sisr_conv2d_trunk_eligible(f) == 0
        # SISR_PERSIST_MAX_WG caps the grid of every persistent kernel (the multi-tile test knob)
        monkeypatch.delenv('SISR_TRUNK')
        monkeypatch.delenv('SISR_TRUNK_UP')
        # the persistent weight-gradient kernel stores the gradient part of its slabs as bf16: the reduction must be told
        g = L.WgradDesc.from_buffer_copy(E.ConvGeom(64, 64, 3, 1, 1).plans(16, 96, 96)[2])
        g.x1 = g.g1 = g.g2 = g.slab = g.qa = g.qb = g.qd = one
        g.x_bf16 = g.g_bf16 = 1
        g.gpro_mode = L.PRO_BNBWD
        assert lib.sisr_wgrad_trunk_eligible(g) == 1 and lib.sisr_wgrad_bf16_slab_lead(g) == g.slab_elems and g.slab_elems % 4 == 0
        monkeypatch.setenv('SISR_SLAB_BF16', '0')
        assert lib.sisr_wgrad_bf16_slab_lead(g) == 0
        monkeypatch.delenv('SISR_SLAB_BF16')
        g.g_bf16 = 0
        assert lib.sisr_wgrad_trunk_eligible(g) == 0 and lib.sisr_wgrad_bf16_slab_lead(g) == 0     # generic kernel: fp32 slabs
        f.pro_mode, f.x2, f.x_out = L.PRO_NONE, None, None
        monkeypatch.setenv('SISR_PERSIST_MAX_WG', '5')
        assert lib.sisr_conv2d_bf16_parts(f) <= 5
    finally:
        E.set_precision('fp32')


def test_resize_coefficient_tables_match_the_oracle():
    """sisr_resize_coeffs is a HOST function (Pillow's precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter):
    its tables equal the oracle's restatement entry for entry -- down-scaling (anti-aliased, wide support), up-scaling,
    identity and degenerate sizes"""
    import ctypes as C
    import importlib
    import numpy as np
    from oracle import ops as oo
    L = importlib.import_module('single-image-super-resolution_amd._lib')
    lib = L.lib()
    for n_in, n_out in ((178, 64), (218, 96), (500, 192), (333, 192), (28, 14), (37, 64), (64, 64), (5, 3), (7, 2), (1, 4)):
        ks = lib.sisr_resize_coeffs(n_in, n_out, None, None)
        bounds, kk = np.zeros((n_out, 2), np.int32), np.zeros((n_out, ks), np.int32)
        assert lib.sisr_resize_coeffs(n_in, n_out, bounds.ctypes.data_as(C.c_void_p), kk.ctypes.data_as(C.c_void_p)) == ks
        ks_o, b_o, kk_o = oo.pil_bilinear_coeffs(n_in, n_out)
        assert ks == ks_o and np.array_equal(bounds, b_o) and np.array_equal(kk, kk_o), (n_in, n_out)
    assert lib.sisr_resize_coeffs(0, 4, None, None) < 0
    # more than 64 taps per output index (a down-scale above ~31x): refused by the size query as well as by the fill
    assert lib.sisr_resize_coeffs(4096, 64, None, None) < 0 and lib.sisr_resize_coeffs(1984, 64, None, None) == 63
