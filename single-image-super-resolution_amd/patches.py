"""On-device patch pipeline: what happens to every image between the decoder and the generator.

The reference transforms each decoded image on the host, one at a time, inside its DataLoader workers
(config.py:225-231: ``transforms.Resize(image_size_hr[1:])`` -- Pillow's anti-aliased BILINEAR resize of the 8-bit
image --, ``ToTensor()``, ``Normalize((.5, .5, .5), (.5, .5, .5))``), moves the float batch to the device
(train.py:45) and degrades it there (train.py:46 ``utils.lr_from_hr``).  ``PatchPipeline`` takes the batch of DECODED
images as uint8 on the device (a quarter of the float bytes over PCIe) and produces both ``img_hr`` and ``img_lr``
there: one launch for Resize + ToTensor + Normalize (csrc/resample.hip, integer arithmetic bit-exact with Pillow),
one for the bicubic degradation + clamp (the kernel behind ``utils.lr_from_hr``), which reads ``img_hr`` back out of
L2.  ``img_hr`` has to be materialised anyway: it is the discriminator's real batch and the content-loss target.

>>> pipe = PatchPipeline(image_size_hr[1:], image_size_lr[1:])
>>> img_hr, img_lr = pipe(batch_u8.to(device))          # batch_u8: [B, H0, W0, C] uint8, e.g. CelebA 218 x 178 x 3
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .engine import _stream
from .utils import lr_from_hr


class PatchPipeline:
    def __init__(self, image_size_hr, image_size_lr, mean=0.5, std=0.5):
        self.hr, self.lr = (int(image_size_hr[0]), int(image_size_hr[1])), (int(image_size_lr[0]), int(image_size_lr[1]))
        self.mean, self.std = float(mean), float(std)
        self._tables = {}                  # (axis input size, output size, device) -> (bounds, kk, ksize) on the device

    def _axis(self, n_in, n_out, dev):
        key = (n_in, n_out, str(dev))
        if key not in self._tables:
            lib = L.lib()
            ks = L.check_count(lib.sisr_resize_coeffs(n_in, n_out, None, None), 'sisr_resize_coeffs')
            bounds = np.zeros((n_out, 2), dtype=np.int32)
            kk = np.zeros((n_out, ks), dtype=np.int32)
            L.check_count(lib.sisr_resize_coeffs(n_in, n_out, bounds.ctypes.data_as(C.c_void_p), kk.ctypes.data_as(C.c_void_p)),
                          'sisr_resize_coeffs')
            self._tables[key] = (torch.from_numpy(bounds).to(dev), torch.from_numpy(kk).to(dev), ks)
        return self._tables[key]

    def resize_normalize(self, imgs_u8):
        """transforms.Resize + ToTensor + Normalize of config.py:225-231 on a batch: [N, H0, W0, C] uint8 -> [N, C, H, W] float32"""
        if not (isinstance(imgs_u8, torch.Tensor) and imgs_u8.is_cuda and imgs_u8.dtype == torch.uint8 and imgs_u8.dim() == 4):
            raise RuntimeError('PatchPipeline: a [N, H0, W0, C] uint8 batch on the MI355X is expected (got %s %s on %s); there is no '
                               'CPU fallback' % (getattr(imgs_u8, 'dtype', type(imgs_u8)), tuple(getattr(imgs_u8, 'shape', ())),
                                                 getattr(imgs_u8, 'device', '?')))
        x = imgs_u8.contiguous()
        n, h0, w0, c = x.shape
        h, w = self.hr
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
        bx, kx, ksx = self._axis(w0, w, x.device) if w0 != w else (None, None, 0)
        by, ky, ksy = self._axis(h0, h, x.device) if h0 != h else (None, None, 0)
        ptr = lambda t: None if t is None else t.data_ptr()
        L.check(L.lib().sisr_resize_u8_normalize(x.data_ptr(), out.data_ptr(), n, h0, w0, c, h, w, ptr(bx), ptr(kx), ksx,
                                                 ptr(by), ptr(ky), ksy, self.mean, self.std, _stream()),
                'sisr_resize_u8_normalize')
        return out

    def __call__(self, imgs_u8):
        img_hr = self.resize_normalize(imgs_u8)
        return img_hr, lr_from_hr(img_hr, self.lr)            # train.py:46
