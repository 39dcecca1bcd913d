"""Drop-in for the reference's ``model_generator`` module (model_generator.py:1-141): same class
names, constructor signatures, attribute names and state_dict keys; ``forward`` runs the fused
gfx950 schedule of generator_engine.py instead of torch.nn primitives."""
import torch
import torch.nn as nn

from . import generator_engine as GE
from .layers import BatchNorm2d, ConvRef, Marker, PReLU, make_conv


class BasicBlock(nn.Module):
    """residual block of G (model_generator.py:5-19): SN-conv, BN, PReLU, SN-conv, BN + skip."""

    def __init__(self, n_features, sn=True):
        super().__init__()
        self.layers = nn.Sequential(
            make_conv(sn, n_features, n_features, 3, 1, 1),
            BatchNorm2d(n_features),
            PReLU(),
            make_conv(sn, n_features, n_features, 3, 1, 1),
            BatchNorm2d(n_features))

    def topo_entry(self):
        l = self.layers
        return dict(c1=ConvRef(l[0]), bn1=l[1], prelu=l[2].weight, c2=ConvRef(l[3]), bn2=l[4])

    def forward(self, x):
        raise RuntimeError('BasicBlock is scheduled by its Generator on the MI355X path')


class Generator(nn.Module):
    def __init__(self, n_blocks, n_features_block, n_features_last, list_scales, use_sn=False, input_channels=3):
        """Same arguments as the reference (model_generator.py:23).  Trunk convs are always
        spectrally normalised; upscale/end convs only if ``use_sn`` (model_generator.py:43-63)."""
        super().__init__()
        assert n_features_last % 4 == 0
        for s in list_scales:
            if s != 2:
                raise NotImplementedError('PixelShuffle factor %r: the MI355X path implements the factor 2 '
                                          'every reference configuration uses' % (s,))
        self.n_features_last = n_features_last
        self.first_layers = nn.Sequential(make_conv(True, input_channels, n_features_block, 9, 1, 4), PReLU())
        self.block_list = nn.Sequential(*[BasicBlock(n_features_block) for _ in range(n_blocks)])
        self.block_list_end = nn.Sequential(make_conv(True, n_features_block, n_features_block, 3, 1, 1),
                                            BatchNorm2d(n_features_block))
        self.upscale = nn.Sequential(*[
            nn.Sequential(make_conv(use_sn, n_features_block if i == 0 else n_features_last // list_scales[i - 1] ** 2,
                                    n_features_last, 3, 1, 1, shuffle2=True),
                          Marker('PixelShuffle(2)'), PReLU())
            for i in range(len(list_scales))])
        self.end = nn.Sequential(make_conv(use_sn, n_features_last // list_scales[-1] ** 2, input_channels, 3, 1, 1),
                                 Marker('Tanh'))

    # ---- reference API ---------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=False):
        """strict=False by default and a coverage report when the checkpoint differs, like the
        reference (model_generator.py:65-84)."""
        super().load_state_dict(state_dict, strict=strict)
        a, b = self.state_dict(), state_dict
        if a.keys() != b.keys() or any(a[k].shape != b[k].shape or torch.any(a[k] != b[k].to(a[k].device))
                                       for k in a.keys()):
            n_a = sum(x.nelement() for x in a.values())
            n_b = sum(x.nelement() for x in b.values())
            n_i = sum(a[k].nelement() for k in set(a.keys()) & set(b.keys()))
            print('generator loaded at %.1f%% (%.2f M)' % (n_i / n_a * 100, n_i * 1e-6))
            print('  - architecture: %d tensors (%.2f M)' % (len(a), n_a * 1e-6))
            print('  - checkpoint  : %d tensors (%.2f M)' % (len(b), n_b * 1e-6))
            print('  - missing     :', len(a.keys() - b.keys()), a.keys() - b.keys())
            print('  - unused      :', len(b.keys() - a.keys()), b.keys() - a.keys())

    def freeze(self, freeze_upscale=False, freeze_end=False):
        """model_generator.py:103-115"""
        layer_list = [self.first_layers, self.block_list, self.block_list_end]
        if freeze_upscale:
            layer_list.append(self.upscale)
        if freeze_end:
            layer_list.append(self.end)
        for layer in layer_list:
            layer.requires_grad = False
            for x in layer.parameters():
                x.requires_grad = False

    # ---- fused schedule ----------------------------------------------------------------------------
    def _topology(self, with_end=True):
        t = GE.Topology()
        t.first, t.first_prelu = ConvRef(self.first_layers[0]), self.first_layers[1].weight
        t.blocks = [b.topo_entry() for b in self.block_list]
        t.trunk_end, t.trunk_bn = ConvRef(self.block_list_end[0]), self.block_list_end[1]
        t.long_skip = True
        t.stages = [(ConvRef(s[0]), s[2].weight) for s in self.upscale]
        t.end = ConvRef(self.end[0]) if with_end else None
        return t

    def _end_module(self):
        return self.end

    def forward_no_end(self, x):
        """model_generator.py:86-96: everything but ``end`` (NCHW activation after the last upscale stage)"""
        return GE.generator_apply(self._topology(with_end=False), self, x)

    def forward(self, x):
        return GE.generator_apply(self._topology(), self, x)


class GeneratorSuffix(nn.Module):
    """model_generator.py:117-141: prefix (a Generator or another GeneratorSuffix) + SN-conv(nl/4 -> nl)
    + PixelShuffle(2) + PReLU, re-using the innermost prefix's ``end`` conv (kept in a list so its
    parameters are registered once, through ``base``)."""

    def __init__(self, prefix, freeze_prefix=False, **kwargs):
        super().__init__()
        self.base = prefix
        self.n_features_last = prefix.n_features_last
        self.upscale = nn.Sequential(make_conv(True, self.n_features_last // 4, self.n_features_last, 3, 1, 1,
                                               shuffle2=True),
                                     Marker('PixelShuffle(2)'), PReLU())
        self.end = [prefix.end[0] if type(prefix.end) == list else prefix.end]
        if freeze_prefix:
            prefix.freeze(**kwargs)

    def freeze(self, **kwargs):
        self.base.freeze(**kwargs)

    def _end_module(self):
        return self.end[0]

    def _topology(self, with_end=True):
        t = self.base._topology(with_end=False)
        t.stages = t.stages + [(ConvRef(self.upscale[0]), self.upscale[2].weight)]
        t.end = ConvRef(self._end_module()[0]) if with_end else None
        return t

    def forward_no_end(self, x):
        """model_generator.py:133-136"""
        return GE.generator_apply(self._topology(with_end=False), self, x)

    def forward(self, x):
        return GE.generator_apply(self._topology(), self, x)
