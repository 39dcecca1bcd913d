"""Drop-in for the hot-path part of the reference's ``utils`` module: ``lr_from_hr``
(utils.py:16-31) -- bicubic (align_corners=True, A=-0.75) HR->LR degradation followed by a clamp
to [-1, 1] -- on the gfx950 kernels of csrc/resample.hip.  Differentiable (the reference applies
it to ``fake`` in its unsupervised branch, train.py:96).  The plotting / checkpoint helpers of the
reference's utils.py are outside the hot path and not provided.
"""
import torch

from . import _lib as L
from .engine import _stream, require_gpu_tensor


class _Bicubic(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, clamp):
        require_gpu_tensor(x, 'lr_from_hr input')
        x = x.contiguous()
        n, c, h, w = x.shape
        oh, ow = int(size[0]), int(size[1])
        y = torch.empty((n, c, oh, ow), dtype=torch.float32, device=x.device)
        L.check(L.lib().sisr_bicubic_fwd(x.data_ptr(), y.data_ptr(), n * c, h, w, oh, ow, int(clamp), _stream()),
                'sisr_bicubic_fwd')
        ctx.shape, ctx.clamp = (n, c, h, w, oh, ow), clamp
        if clamp:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w, oh, ow = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty((n, c, h, w), dtype=torch.float32, device=dy.device)
        yc = ctx.saved_tensors[0] if ctx.clamp else None
        L.check(L.lib().sisr_bicubic_bwd(dy.data_ptr(), None if yc is None else yc.data_ptr(), dx.data_ptr(),
                                         n * c, h, w, oh, ow, _stream()), 'sisr_bicubic_bwd')
        return dx, None, None


def _subsampling_interpolation(img_hr, image_size_lr):
    """utils.py:16-17"""
    return _Bicubic.apply(img_hr, tuple(image_size_lr), False)


def lr_from_hr(img_hr, image_size_lr, device='cpu'):
    """utils.py:22-31.  ``device`` is accepted for signature compatibility (the reference only uses
    it to place the clamp bounds); the result lives on ``img_hr``'s device."""
    return _Bicubic.apply(img_hr, tuple(image_size_lr), True)
