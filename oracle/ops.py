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


def prelu(x, a):
    """nn.PReLU() with ONE shared scalar slope (model_generator.py:12,34,48,59,126)."""
    return torch.where(x > 0, x, a * x)


def leaky_relu(x, slope=0.01):
    """nn.LeakyReLU() default slope 0.01 (model_discriminator.py:12,40,50)."""
    return torch.where(x > 0, x, slope * x)


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
