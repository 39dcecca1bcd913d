"""Host logic of the 64 x 64 x 9-tap weight-gradient family (csrc/wgrad_deep.hip): the planner's tiling and the index arithmetic the
kernel derives from it -- halo coordinates, the stride-2 parity planes, tiles of the flattened (image, row) space that straddle images,
the split into pixel blocks -- replayed in numpy against a direct weight-gradient sum on the CPU (no GPU, no kernel launch).

The LDS images persist across tiles and start as garbage, as on the device: a position whose dy must be zero but is not, or an x pixel
that a valid position reads but the producers did not stage for THIS tile, changes the result."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module('single-image-super-resolution_amd._lib')


def _plan(n, h, w, cin, cout, stride, target=0):
    g = L.WgradDesc()
    ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    g.N, g.H, g.W, g.Cin, g.Ho, g.Wo, g.Cout = n, h, w, cin, ho, wo, cout
    g.KH = g.KW = 3
    g.stride, g.pad_y, g.pad_x = stride, 1, 1
    assert L.lib().sisr_wgrad_plan_bf16(C.byref(g), 512) == 0
    rc = L.lib().sisr_wgrad_deep_plan(C.byref(g), target)
    return rc, g


def _replay(g, x, dy):
    """dW[ky, kx] (one channel pair) through the kernel's maps; x: [N, H, W], dy: [N, Ho, Wo] integers"""
    p, S = g.deep, g.stride
    N, H, W, Ho, Wo = g.N, g.H, g.W, g.Ho, g.Wo
    NQ, IW, IWd = N * Ho, p.IW, p.IWd
    XPL = p.XP_max // 4 if S == 2 else p.XP_max
    rng = np.random.default_rng(5)
    bufs = [(rng.integers(-9, 9, p.XP_max).astype(np.float64), rng.integers(-9, 9, p.NPOS_max).astype(np.float64)) for _ in range(2)]
    assert p.NITX * 32 >= p.IH_max * IW and p.NITD * 32 >= p.NPOS_max
    assert p.lds_bytes == 4 * (p.XP_max + p.NPOS_max) * 64 + 7 * 64 * 4 <= 160 * 1024

    def rbase(q):
        n = q // Ho
        return n * p.PR + (q - n * Ho) * S
    dw = np.zeros((p.n_pb, 3, 3))
    seen = np.zeros(p.n_tiles, np.int32)
    assert (p.n_pb - 1) * p.tiles_per_pb < p.n_tiles <= p.n_pb * p.tiles_per_pb
    for pb in range(p.n_pb):
        t_begin = pb * p.tiles_per_pb
        ntile = min(p.n_tiles, t_begin + p.tiles_per_pb) - t_begin
        for i in range(ntile):
            t = t_begin + i
            seen[t] += 1
            xb, db = bufs[i & 1]
            tq, tx = divmod(t, p.tiles_x)
            q0, ox0 = tq * p.TH, tx * p.TW
            qend = min(q0 + p.TH, NQ)
            pb0 = rbase(q0)
            span = rbase(qend - 1) - pb0
            assert span % S == 0
            npix, npos = (span + 3) * IW, ((span // S + 1) * IWd + 15) & ~15
            assert span + 3 <= p.IH_max and npos <= p.NPOS_max
            # ---- producers: x halo
            for pix in range(32 * p.NITX):
                hr, hc = divmod(pix, IW)
                P = pb0 + hr
                n = P // p.PR
                iy, ix = P - n * p.PR - 1, ox0 * S - 1 + hc
                ok = pix < npix and n < N and 0 <= iy < H and 0 <= ix < W
                lof = pix if S == 1 else ((hr & 1) * 2 + (hc & 1)) * XPL + (hr >> 1) * IWd + (hc >> 1)
                if pix < npix:
                    assert lof < p.XP_max
                    xb[lof] = x[n, iy, ix] if ok else 0.0
            # ---- producers: dy in halo coordinates
            for pos in range(32 * p.NITD):
                hrow, c = divmod(pos, IWd)
                P = pb0 + hrow * S
                n = P // p.PR
                oy = (P - n * p.PR) // S
                q, ox = n * Ho + oy, ox0 + c
                ok = pos < npos and oy < Ho and q < qend and c < p.TW and ox < Wo
                if pos < npos:
                    db[pos] = dy.reshape(NQ, Wo)[q, ox] if ok else 0.0
            # ---- consumers
            for ky in range(3):
                for kx in range(3):
                    off = ky * IW + kx if S == 1 else ((ky & 1) * 2 + (kx & 1)) * XPL + (ky >> 1) * IWd + (kx >> 1)
                    assert off + npos <= p.XP_max
                    dw[pb, ky, kx] += float(np.dot(xb[off:off + npos], db[:npos]))
    assert (seen == 1).all()
    return dw.sum(0)


def _direct(x, dy, S):
    N, H, W = x.shape
    _, Ho, Wo = dy.shape
    xp = np.zeros((N, H + 2 + S, W + 2 + S))
    xp[:, 1:H + 1, 1:W + 1] = x
    dw = np.zeros((3, 3))
    for ky in range(3):
        for kx in range(3):
            dw[ky, kx] = (xp[:, ky:ky + S * Ho:S, kx:kx + S * Wo:S][:, :Ho, :Wo] * dy).sum()
    return dw


CASES = [
    # the discriminator's stack at 96 x 96 and 192 x 192 patches (model_discriminator.py:39-44), B = 16
    (16, 96, 96, 64, 64, 2), (16, 48, 48, 64, 128, 1), (16, 48, 48, 128, 128, 2), (16, 24, 24, 128, 256, 1),
    (16, 24, 24, 256, 256, 2), (16, 12, 12, 256, 512, 1), (16, 12, 12, 512, 512, 2),
    (16, 192, 192, 64, 64, 2), (16, 96, 96, 64, 128, 1),
    # ragged: odd sizes, a single image, widths that are no multiple of any tile width
    (3, 7, 9, 64, 64, 1), (3, 7, 9, 64, 64, 2), (1, 24, 24, 64, 64, 1), (5, 13, 70, 64, 128, 1), (2, 11, 70, 64, 64, 2),
    (16, 6, 6, 64, 64, 1), (16, 6, 6, 64, 64, 2),
]


@pytest.mark.parametrize('n,h,w,cin,cout,s', CASES)
def test_maps_reproduce_the_weight_gradient(n, h, w, cin, cout, s):
    rc, g = _plan(n, h, w, cin, cout, s)
    assert rc == 0 and g.deep.enabled == 1
    rng = np.random.default_rng(n * 1000 + h * 10 + s)
    x = rng.integers(-4, 5, (n, h, w)).astype(np.float64)
    dy = rng.integers(-4, 5, (n, g.Ho, g.Wo)).astype(np.float64)
    np.testing.assert_array_equal(_replay(g, x, dy), _direct(x, dy, s))


def test_pixel_blocks_follow_the_target_and_unsupported_geometries_are_refused():
    rc, g = _plan(16, 48, 48, 64, 128, 1, target=64)
    assert rc == 0 and g.deep.n_pb * g.deep.n_cib * g.deep.n_cob <= 64
    for bad in ((16, 48, 48, 32, 64, 1), (16, 48, 48, 64, 96, 1)):
        assert _plan(*bad)[0] != 0
    g = L.WgradDesc()
    g.N, g.H, g.W, g.Cin, g.Ho, g.Wo, g.Cout = 16, 48, 48, 64, 48, 48, 64
    g.KH = g.KW = 1
    g.stride, g.pad_y, g.pad_x = 1, 0, 0
    assert L.lib().sisr_wgrad_plan_bf16(C.byref(g), 512) == 0
    assert L.lib().sisr_wgrad_deep_plan(C.byref(g), 0) != 0 and g.deep.enabled == 0
