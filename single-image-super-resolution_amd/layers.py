"""Parameter containers that reproduce the reference's module tree, so ``state_dict()`` keys,
``parameters()`` order and default initialisation (same RNG consumption order under
``torch.manual_seed``) match keyber/Single-Image-Super-Resolution exactly.  They hold state only:
the arithmetic is scheduled by the owning network on the HIP kernels (generator_engine.py ...).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .engine import ConvGeom


def _conv_init(cout, cin, k):
    """nn.Conv2d.reset_parameters(): kaiming_uniform_(a=sqrt(5)) then bias U(-1/sqrt(fan_in), .)."""
    w = torch.empty(cout, cin, k, k)
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    bound = 1.0 / math.sqrt(cin * k * k)
    b = torch.empty(cout).uniform_(-bound, bound)
    return w, b


class _ConvBase(nn.Module):
    def _fused_only(self):
        raise RuntimeError('%s is a parameter container of the fused MI355X path; call the owning '
                           'network (Generator / Discriminator / ...) instead' % type(self).__name__)

    def forward(self, x):
        self._fused_only()


class Conv2d(_ConvBase):
    """state of nn.Conv2d(cin, cout, k, stride, padding): keys ``weight``, ``bias``."""

    def __init__(self, cin, cout, k, stride=1, padding=0, shuffle2=False):
        super().__init__()
        w, b = _conv_init(cout, cin, k)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(b)
        self.geom = ConvGeom(cin, cout, k, stride, padding, shuffle2)


class SNConv2d(_ConvBase):
    """state of torch.nn.utils.spectral_norm(nn.Conv2d(...)) (legacy hook API, as imported at
    model_generator.py:3 / model_discriminator.py:2): keys ``bias``, ``weight_orig``, ``weight_u``,
    ``weight_v``."""

    def __init__(self, cin, cout, k, stride=1, padding=0, shuffle2=False):
        super().__init__()
        w, b = _conv_init(cout, cin, k)
        self.bias = nn.Parameter(b)
        self.weight_orig = nn.Parameter(w)
        # spectral_norm.py SpectralNorm.apply: u, v ~ normalize(N(0,1)), eps 1e-12
        u = F.normalize(w.new_empty(cout).normal_(0, 1), dim=0, eps=1e-12)
        v = F.normalize(w.new_empty(cin * k * k).normal_(0, 1), dim=0, eps=1e-12)
        self.register_buffer('weight_u', u)
        self.register_buffer('weight_v', v)
        self.geom = ConvGeom(cin, cout, k, stride, padding, shuffle2)


class ConvRef:
    """Engine-facing view of a conv container (always reads the module's CURRENT tensors, so
    ``.to(device)`` / ``load_state_dict`` are honoured)."""

    def __init__(self, module):
        self.m = module
        self.geom = module.geom

    @property
    def weight(self):
        return self.m.weight_orig if isinstance(self.m, SNConv2d) else self.m.weight

    @property
    def bias(self):
        return self.m.bias

    @property
    def u(self):
        return self.m.weight_u if isinstance(self.m, SNConv2d) else None

    @property
    def v(self):
        return self.m.weight_v if isinstance(self.m, SNConv2d) else None


def make_conv(sn, cin, cout, k, stride=1, padding=0, shuffle2=False):
    return (SNConv2d if sn else Conv2d)(cin, cout, k, stride, padding, shuffle2)


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d used as a state container (weight, bias, running_*, num_batches_tracked)."""

    def forward(self, x):
        raise RuntimeError('BatchNorm2d is fused into the neighbouring convolutions of the MI355X path')


class PReLU(nn.PReLU):
    """nn.PReLU() (one shared slope, init 0.25) as a state container."""

    def forward(self, x):
        raise RuntimeError('PReLU is fused into the neighbouring convolutions of the MI355X path')


class Marker(nn.Module):
    """parameter-free placeholder that keeps nn.Sequential indices aligned with the reference
    (PixelShuffle / Tanh / LeakyReLU / Sigmoid positions)."""

    def __init__(self, what):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return self.what

    def forward(self, x):
        raise RuntimeError('%s is fused into the neighbouring kernels of the MI355X path' % self.what)
