"""Drop-in for the reference's ``model_generator_progressive`` module
(model_generator_progressive.py:1-65): non-spectral-norm trunk without the long skip, and
stackable suffixes conv(n->n) + PixelShuffle(2) + PReLU, each with its own conv(n/4 -> 3) + Tanh."""
import torch.nn as nn

from . import generator_engine as GE
from .layers import BatchNorm2d, ConvRef, Marker, PReLU, make_conv
from .model_generator import BasicBlock as _SNBlock


class BasicBlock(_SNBlock):
    """model_generator_progressive.py:4-18 (plain convs)"""

    def __init__(self, n_features):
        super().__init__(n_features, sn=False)


class GeneratorProgresiveBase(nn.Module):
    def __init__(self, n_blocks, n_features, input_channels=3):
        super().__init__()
        self.first_layers = nn.Sequential(make_conv(False, input_channels, n_features, 9, 1, 4), PReLU())
        self.block_list = nn.Sequential(*[BasicBlock(n_features) for _ in range(n_blocks)])
        self.block_list_end = nn.Sequential(make_conv(False, n_features, n_features, 3, 1, 1),
                                            BatchNorm2d(n_features))

    def _topology(self):
        t = GE.Topology()
        t.first, t.first_prelu = ConvRef(self.first_layers[0]), self.first_layers[1].weight
        t.blocks = [b.topo_entry() for b in self.block_list]
        t.trunk_end, t.trunk_bn = ConvRef(self.block_list_end[0]), self.block_list_end[1]
        t.long_skip = False
        return t

    def forward(self, x):
        """model_generator_progressive.py:40-44: first conv + PReLU, the residual blocks, conv + BatchNorm -- the trunk's
        n_features-channel output (NCHW), no long skip, no upscale stage"""
        return GE.generator_apply(self._topology(), self, x)


class _Beginning(nn.Sequential):
    """``beginning = Sequential(prefix, conv, PixelShuffle, PReLU)`` (model_generator_progressive.py:52-56)"""

    def _topology(self):
        prefix = self[0]
        t = prefix._topology()
        t.stages = t.stages + [(ConvRef(self[1]), self[3].weight)]
        return t

    def forward(self, x):
        """the Sequential itself (what the next GeneratorSuffix wraps as its prefix, model_generator_progressive.py:73-76):
        prefix, conv, PixelShuffle(2), PReLU -> the activated n_features / 4 channels at twice the resolution (NCHW)"""
        return GE.generator_apply(self._topology(), self, x)


class GeneratorSuffix(nn.Module):
    def __init__(self, prefix, n_features, input_channels=3):
        super().__init__()
        assert n_features % 4 == 0
        self.beginning = _Beginning(prefix, make_conv(False, n_features, n_features, 3, 1, 1, shuffle2=True),
                                    Marker('PixelShuffle(2)'), PReLU())
        self.end = nn.Sequential(make_conv(False, n_features // 4, input_channels, 3, 1, 1), Marker('Tanh'))

    def _topology(self):
        t = self.beginning._topology()
        t.end = ConvRef(self.end[0])
        return t

    def forward(self, x):
        return GE.generator_apply(self._topology(), self, x)
