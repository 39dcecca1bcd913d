"""Primitive restatements (CPU, fp32) -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Every function is differentiable through torch autograd on CPU so that the same restatement
yields the backward-pass oracle.  Citations are file:line under /root/reference (call sites)
and, where the arithmetic is PyTorch's, the published algorithm that is restated.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# spectral norm (legacy torch.nn.utils.spectral_norm hook; model_generator.py:3,10,13,33,39,45,
# 52,123 and model_discriminator.py:2,10,39).  Published algorithm (torch/nn/utils/
# spectral_norm.py, SpectralNorm.compute_weight): with W_mat = W_orig.reshape(Cout, -1),
#   training: v <- normalize(W_mat^T u), u <- normalize(W_mat v)   (one iteration, eps=1e-12,
#             in place on the buffers, no grad), then
#   always:   sigma = u . (W_mat v)   (u, v constants for autograd), W = W_orig / sigma.
# --------------------------------------------------------------------------------------------
def _normalize(t, eps):
    return t / torch.clamp(torch.linalg.vector_norm(t), min=eps)


def spectral_norm_weight(w_orig, u, v, training, eps=1e-12):
    """Returns (W, u_new, v_new).  u_new/v_new are detached; W carries grad to w_orig."""
    w_mat = w_orig.reshape(w_orig.shape[0], -1)
    u = u.detach()
    v = v.detach()
    if training:
        with torch.no_grad():
            v = _normalize(torch.mv(w_mat.t(), u), eps)
            u = _normalize(torch.mv(w_mat, v), eps)
    sigma = torch.dot(u, torch.mv(w_mat, v))
    return w_orig / sigma, u, v


# --------------------------------------------------------------------------------------------
# BatchNorm2d (model_generator.py:11,14,40; model_discriminator.py:11).  Published algorithm:
# training: per-channel mean and *biased* variance over (N,H,W); y = (x-mean)/sqrt(var+eps)*g+b;
# running_mean <- (1-m) rm + m mean; running_var <- (1-m) rv + m * var * n/(n-1); m = 0.1.
# eval: the running statistics are used.
# --------------------------------------------------------------------------------------------
def batch_norm(x, weight, bias, running_mean, running_var, training, momentum=0.1, eps=1e-5):
    """Returns (y, running_mean_new, running_var_new)."""
    if training:
        n = x.shape[0] * x.shape[2] * x.shape[3]
        mean = x.mean(dim=(0, 2, 3))
        var = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
        with torch.no_grad():
            rm = (1 - momentum) * running_mean + momentum * mean
            rv = (1 - momentum) * running_var + momentum * var * (n / max(n - 1, 1))
    else:
        mean, var = running_mean, running_var
        rm, rv = running_mean, running_var
    inv = torch.rsqrt(var + eps)
    y = (x - mean[None, :, None, None]) * (inv * weight)[None, :, None, None] + bias[None, :, None, None]
    return y, rm.detach(), rv.detach()


# Test hook (tests/test_gpu_full_size.py): a list of boolean tensors, one per PReLU / LeakyReLU call in forward order.  When set, an
# activation takes the next one as its mask instead of (x > 0) -- the oracle then differentiates the SAME piecewise-linear function as
# the implementation whose masks these are, so a comparison of gradients is not blurred by pre-activations within rounding of zero that
# the two sides sort differently.  None (always, outside that test): the reference's own semantics.
ACT_MASKS = None


def _positive(x):
    if ACT_MASKS is None:
        return x > 0
    m = ACT_MASKS.pop(0)
    assert m.shape == x.shape and m.dtype == torch.bool, (tuple(m.shape), tuple(x.shape))
    return m


def prelu(x, a):
    """nn.PReLU() with ONE shared scalar slope (model_generator.py:12,34,48,59,126)."""
    return torch.where(_positive(x), x, a * x)


def leaky_relu(x, slope=0.01):
    """nn.LeakyReLU() default slope 0.01 (model_discriminator.py:12,40,50)."""
    return torch.where(_positive(x), x, slope * x)


def pixel_shuffle(x, r):
    """nn.PixelShuffle(r) (model_generator.py:47,58,125): out[n,c,h*r+i,w*r+j] = in[n,c*r*r+i*r+j,h,w]."""
    n, c, h, w = x.shape
    co = c // (r * r)
    x = x.reshape(n, co, r, r, h, w).permute(0, 1, 4, 2, 5, 3)
    return x.reshape(n, co, h * r, w * r)


def conv2d(x, w, b, stride=1, padding=0):
    """nn.Conv2d cross-correlation; the primitive itself is PyTorch's (torch 2.10 CPU)."""
    return F.conv2d(x, w, b, stride=stride, padding=padding)


def max_pool2x2(x):
    """nn.MaxPool2d(2, 2): floors odd sizes (model_content_extractor.py:84-92)."""
    return F.max_pool2d(x, kernel_size=2, stride=2)


def linear(x, w, b):
    return x @ w.t() + b


# --------------------------------------------------------------------------------------------
# bicubic, align_corners=True (utils.py:16-17), published algorithm (ATen UpSampleBicubic2d):
#   scale = (in-1)/(out-1)  (0 when out == 1);  src = scale*dst;  i0 = floor(src); t = src-i0
#   taps i0-1..i0+2 clamped to [0, in-1], weights (A = -0.75):
#     w0 = c2(t+1), w1 = c1(t), w2 = c1(1-t), w3 = c2(2-t)
#     c1(x) = ((A+2)x - (A+3))x^2 + 1 ;  c2(x) = ((A x - 5A) x + 8A) x - 4A
# Separable, so it is restated as two dense interpolation matrices.
# --------------------------------------------------------------------------------------------
_A = -0.75


def _c1(x):
    return ((_A + 2.0) * x - (_A + 3.0)) * x * x + 1.0


def _c2(x):
    return ((_A * x - 5.0 * _A) * x + 8.0 * _A) * x - 4.0 * _A


def bicubic_matrix(n_in, n_out):
    """Dense (n_out, n_in) float32 interpolation matrix for one axis."""
    m = np.zeros((n_out, n_in), dtype=np.float64)
    scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
    for d in range(n_out):
        # the index arithmetic is done in float32 like ATen's (scalar_t = float)
        src = np.float32(scale) * np.float32(d)
        i0 = int(math.floor(src))
        t = float(np.float32(src) - np.float32(i0))
        ws = (_c2(t + 1.0), _c1(t), _c1(1.0 - t), _c2(2.0 - t))
        for k, wk in enumerate(ws):
            idx = min(max(i0 - 1 + k, 0), n_in - 1)
            m[d, idx] += wk
    return m.astype(np.float32)


def bicubic_align_corners(x, size):
    """F.interpolate(x, size, mode='bicubic', align_corners=True) for NCHW x (utils.py:17)."""
    oh, ow = size
    my = torch.from_numpy(bicubic_matrix(x.shape[2], oh))
    mx = torch.from_numpy(bicubic_matrix(x.shape[3], ow))
    return torch.einsum('oh,nchw,pw->ncop', my, x, mx)


def lr_from_hr(img_hr, image_size_lr):
    """utils.lr_from_hr (utils.py:22-31): bicubic then clamp to [-1, 1] (utils.py:19-20)."""
    return torch.clamp(bicubic_align_corners(img_hr, image_size_lr), -1.0, 1.0)


# --------------------------------------------------------------------------------------------
# patch pipeline: the transforms the reference's dataset applies to every decoded image
# (config.py:225-231: transforms.Resize(image_size_hr[1:]), ToTensor(), Normalize(.5, .5)) followed by
# utils.lr_from_hr (train.py:45-46).  The arithmetic lives in third-party code that is not in the reference tree:
#   * torchvision.transforms.Resize on a PIL image = PIL.Image.resize((w, h), BILINEAR), Pillow's ImagingResample
#     (src/libImaging/Resample.c), restated below: a separable triangle filter whose support grows with the
#     down-scaling factor (anti-aliasing), coefficients quantised to 22 fractional bits, int32 accumulation with a
#     half-unit offset, and a ROUNDING TO uint8 AFTER EACH of the two passes (horizontal first);
#   * ToTensor = uint8 HWC -> float32 CHW / 255;  Normalize(.5, .5) = (t - 0.5) / 0.5.
# Pinned against Pillow itself (tests/test_oracle_golden.py, tests/golden/patch_pipeline.npz).
# --------------------------------------------------------------------------------------------
PIL_PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc of Resample.c for the whole-image box and the BILINEAR filter
    (support 1.0).  -> (ksize, bounds [out, 2] = (first input index, tap count), kk [out, ksize] int32)"""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size          # (in1 - in0) are floats in the C code
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(ksize, dtype=np.float64)
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
        ww = 0.0                                   # (accumulated in tap order, as the C code does)
        for x in range(xmax):
            ww += w[x]
        if ww != 0.0:
            w[:xmax] /= ww
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PIL_PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PIL_PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pil_resample_axis(img, out_size, axis):
    """one 8-bit pass of ImagingResample along `axis` of an HWC uint8 array"""
    in_size = img.shape[axis]
    ksize, bounds, kk = pil_bilinear_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        xmin, xmax = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PIL_PRECISION_BITS - 1), dtype=np.int64)
        for x in range(xmax):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PIL_PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize_bilinear(img_u8, out_h, out_w):
    """PIL.Image.fromarray(img).resize((out_w, out_h), Image.BILINEAR) for an HWC uint8 array: horizontal pass, then
    vertical pass, each skipped when the size does not change (ImagingResample's need_horizontal / need_vertical)"""
    x = img_u8
    if x.shape[1] != out_w:
        x = _pil_resample_axis(x, out_w, 1)
    if x.shape[0] != out_h:
        x = _pil_resample_axis(x, out_h, 0)
    return x


def patch_pipeline(imgs_u8, image_size_hr, image_size_lr):
    """config.py:225-231 + train.py:45-46 on a batch of decoded images [N, H0, W0, C] uint8:
    -> (img_hr [N, C, H, W] float32 in [-1, 1], img_lr = lr_from_hr(img_hr))"""
    h, w = image_size_hr
    hr = np.stack([pil_resize_bilinear(im, h, w) for im in imgs_u8])
    t = torch.from_numpy(hr).permute(0, 3, 1, 2).contiguous().to(torch.float32).div(255)      # ToTensor
    t = (t - 0.5) / 0.5                                                                       # Normalize(.5, .5)
    return t, lr_from_hr(t, image_size_lr)
