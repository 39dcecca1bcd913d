"""Drop-in for the reference's ``model_content_extractor`` module
(model_content_extractor.py:1-73): ``identity()``, ``MaskedVGG(mask)``, ``get_size`` and the module
constants, with ``MaskedVGG.forward`` on the gfx950 kernels (vgg_engine.py).

Weights: the reference builds ``torchvision.models.vgg19(pretrained=True).features[:k]``
(model_content_extractor.py:43), a remote fetch, and fails if that fails.  So does this drop-in:
``MaskedVGG(mask)`` loads the torchvision weights (or the state_dict file named by
``SISR_VGG19_WEIGHTS``: torchvision's ``vgg19`` checkpoint, keys ``features.<i>.weight|bias``) and
RAISES when neither is available -- a content loss against random features would train silently
wrong.  Only an explicit ``pretrained=False`` gives the seeded default initialisation (tests,
benchmarks); load weights afterwards with ``load_state_dict`` (keys ``layers.<i>.weight|bias``, as
in the reference).
"""
import os

import torch
import torch.nn as nn

from . import vgg_engine as VE
from .layers import Conv2d, ConvRef, Marker

# indices of the MaxPool layers in VGG19.features (the last one is never used)
maxPool_indexes = (4, 9, 18, 27, 36)
maxPool_indexes_before_act = [x - 1 for x in maxPool_indexes]
# feature-map widths (for tests)
layersSize = (64, 128, 256, 512, 512)

_VGG19_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M')


def identity():
    """plain pixel MSE (model_content_extractor.py:12-14)"""
    return nn.Identity()


def _vgg19_feature_modules(n_layers, width_div=1):
    mods, cin = [], 3
    for v in _VGG19_CFG:
        if v == 'M':
            mods.append(Marker('MaxPool2d(2,2)'))
        else:
            mods.append(Conv2d(cin, v // width_div, 3, 1, 1))
            # frozen stack: its data gradients never carry a BatchNorm-backward prologue, so the trunk-shaped conv1_2 (64 -> 64)
            # takes conv_deep.hip for that role and the planner may choose tiles for a light prologue (engine.ConvGeom)
            mods[-1].geom.deep_dgrad = mods[-1].geom.light_backward = True
            mods.append(Marker('ReLU'))
            cin = v // width_div
    return mods[:n_layers]


class _VGGPrefix(nn.Sequential):
    """``vgg19.features[:n]`` as the reference's vgg_4conv_1maxPool returns it: a frozen nn.Sequential (state_dict keys
    ``<i>.weight|bias``) whose forward gives the NCHW feature map behind its last layer, here a ReLU"""

    def _program(self):
        if getattr(self, '_prog', None) is None:
            convs, pool_pending = [], False
            mods = list(self)
            last_conv = max(i for i, m in enumerate(mods) if isinstance(m, Conv2d))
            for i, m in enumerate(mods):
                if isinstance(m, Conv2d):
                    convs.append(dict(ref=ConvRef(m), pool_before=pool_pending, tap=0 if i == last_conv else None))
                    pool_pending = False
                elif m.what.startswith('MaxPool'):
                    pool_pending = True
            assert not pool_pending and last_conv == len(mods) - 2          # ... conv, ReLU
            object.__setattr__(self, '_prog', VE.Program(convs, 1, last_tap_relu=True))
        return self._prog

    def forward(self, x):
        n, _, h, w = x.shape
        pools = sum(1 for m in self if not isinstance(m, Conv2d) and m.what.startswith('MaxPool'))
        cout = [m for m in self if isinstance(m, Conv2d)][-1].weight.shape[0]
        return VE.vgg_apply(self._program(), x).view(n, cout, h >> pools, w >> pools)


def vgg_4conv_1maxPool(pretrained=True):
    """model_content_extractor.py:16-31: the VGG19 feature map in front of the second MaxPool -- ``features[:9]`` (conv1_1,
    conv1_2, pool, conv2_1, conv2_2, each conv with its ReLU), frozen, shape (B, 128, H/2, W/2).  Weights as for MaskedVGG:
    the torchvision checkpoint (or SISR_VGG19_WEIGHTS), an error when neither is available, seeded default initialisation
    only with an explicit ``pretrained=False``."""
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(1234)
    net = _VGGPrefix(*_vgg19_feature_modules(9))
    torch.random.set_rng_state(gen_state)
    if pretrained:
        path = os.environ.get('SISR_VGG19_WEIGHTS')
        try:
            if path:
                sd = torch.load(path, map_location='cpu', weights_only=True)      # a state_dict: tensors only, no pickled code
                sd = {k[len('features.'):]: v for k, v in sd.items() if k.startswith('features.')} or sd
                sd = {k: v for k, v in sd.items() if int(k.split('.')[0]) < 9}
            else:
                import torchvision.models as models
                sd = models.vgg19(pretrained=True).features[:9].state_dict()
            net.load_state_dict(sd, strict=True)
        except Exception as e:                                  # noqa: BLE001
            raise RuntimeError('vgg_4conv_1maxPool(pretrained=True): the VGG19 weights could not be loaded (%s: %s); the '
                               'reference fails here too (model_content_extractor.py:19)' % (type(e).__name__, e)) from e
    net.eval()
    net.requires_grad = False
    for param in net.parameters():
        param.requires_grad = False
    return net


class MaskedVGG(nn.Module):
    """concatenates the VGG19 feature maps taken right before the MaxPools whose mask bit is set;
    output shape (B, -1) (model_content_extractor.py:33-60)."""

    def __init__(self, mask, width_div=1, pretrained=True):
        super().__init__()
        assert 0 < mask < 32
        self.intermediate_layers_kept = [maxPool_indexes_before_act[i] for i in range(5) if mask & (1 << i)]
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(1234)
        self.layers = nn.Sequential(*_vgg19_feature_modules(self.intermediate_layers_kept[-1], width_div))
        torch.random.set_rng_state(gen_state)
        if pretrained:
            if width_div != 1:
                raise ValueError('MaskedVGG: pretrained weights exist only for width_div=1')
            self._load_pretrained()
        self.layers.eval()
        self.layers.requires_grad = False
        for param in self.layers.parameters():
            param.requires_grad = False
        convs, pool_pending = [], False
        for i, m in enumerate(self.layers, 1):
            if isinstance(m, Conv2d):
                tap = self.intermediate_layers_kept.index(i) if i in self.intermediate_layers_kept else None
                convs.append(dict(ref=ConvRef(m), pool_before=pool_pending, tap=tap))
                pool_pending = False
            elif m.what.startswith('MaxPool'):
                pool_pending = True
        self._prog = VE.Program(convs, len(self.intermediate_layers_kept))

    def _load_pretrained(self):
        n = self.intermediate_layers_kept[-1]
        path = os.environ.get('SISR_VGG19_WEIGHTS')
        try:
            if path:
                sd = torch.load(path, map_location='cpu', weights_only=True)      # a state_dict: tensors only, no pickled code
                sd = {k[len('features.'):]: v for k, v in sd.items() if k.startswith('features.')} or sd
                sd = {k: v for k, v in sd.items() if int(k.split('.')[0]) < n}
            else:
                import torchvision.models as models
                sd = models.vgg19(pretrained=True).features[:n].state_dict()
            self.layers.load_state_dict(sd, strict=True)
        except Exception as e:                                  # noqa: BLE001
            raise RuntimeError(
                'MaskedVGG(pretrained=True): the VGG19 weights could not be loaded (%s: %s).  The reference fails '
                'here too (model_content_extractor.py:43).  Provide torchvision with its vgg19 checkpoint, or point '
                'SISR_VGG19_WEIGHTS at a torchvision vgg19 state_dict file, or pass pretrained=False explicitly and '
                'call load_state_dict yourself.' % (type(e).__name__, e)) from e

    def forward(self, x):
        return VE.vgg_apply(self._prog, x)


def get_size(im, mask):
    """model_content_extractor.py:63-73"""
    assert im.shape[1] == 3
    w, h = im.shape[2], im.shape[3]
    size = 0
    for i in range(len(layersSize)):
        if mask & (1 << i):
            size += (w // 2 ** i) * (h // 2 ** i) * layersSize[i]
    return size
