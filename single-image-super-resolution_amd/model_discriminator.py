"""Drop-in for the reference's ``model_discriminator`` module (model_discriminator.py:1-76): same
class names, constructor signature, attributes (``fc_in``, ``fc_mid``, ``conv``, ``fc``) and
state_dict keys; ``forward`` runs the fused gfx950 schedule of discriminator_engine.py."""
import torch.nn as nn

from . import discriminator_engine as DE
from .layers import BatchNorm2d, ConvRef, Marker, make_conv


class BasicBlock(nn.Module):
    """non-residual block of D (model_discriminator.py:5-15): SN-conv(stride), BN, LeakyReLU"""

    def __init__(self, n_in, n_out, stride):
        super().__init__()
        self.layers = nn.Sequential(make_conv(True, n_in, n_out, 3, stride, 1), BatchNorm2d(n_out),
                                    Marker('LeakyReLU(0.01)'))

    def forward(self, x):
        raise RuntimeError('BasicBlock is scheduled by its Discriminator on the MI355X path')


class _Linear(nn.Linear):
    """nn.Linear as a state container (same init / keys)"""

    def forward(self, x):
        raise RuntimeError('Linear is scheduled by its Discriminator on the MI355X path')


class Discriminator(nn.Module):
    def __init__(self, input_shape, list_n_features, list_stride):
        """Same arguments and checks as the reference (model_discriminator.py:19-53); SRGAN uses
        features [64,64,128,128,256,256,512,512] and strides [1,2,1,2,1,2,1,2]."""
        super().__init__()
        w, h = input_shape[1], input_shape[2]
        for x in list_stride:
            assert x in (1, 2), 'strides of 1 or 2 only'
        assert w * h % 4 ** (sum(list_stride) - len(list_stride)) == 0, \
            'every stride-2 layer halves the size: it has to divide'
        assert len(list_n_features) == len(list_stride)
        self.fc_in = w * h * list_n_features[-1] // (4 ** (sum(list_stride) - len(list_stride)))
        self.fc_mid = list_n_features[-1] * 2
        self.conv = nn.Sequential(
            make_conv(True, input_shape[0], list_n_features[0], 3, list_stride[0], 1),
            Marker('LeakyReLU(0.01)'),
            nn.Sequential(*[BasicBlock(list_n_features[i - 1], list_n_features[i], list_stride[i])
                            for i in range(1, len(list_n_features))]))
        self.fc = nn.Sequential(_Linear(self.fc_in, self.fc_mid), Marker('LeakyReLU(0.01)'),
                                _Linear(self.fc_mid, 1), Marker('Sigmoid'))

    def _topology(self):
        t = DE.Topology()
        t.conv0 = ConvRef(self.conv[0])
        t.blocks = [(ConvRef(b.layers[0]), b.layers[1]) for b in self.conv[2]]
        t.fc1, t.fc2 = self.fc[0], self.fc[2]
        t.cache = self.__dict__.setdefault('_sisr_wcache', {})
        return t

    def forward(self, x):
        out = DE.discriminator_apply(self._topology(), self, x)
        assert out.shape[1] == 1
        return out

    def load_state_dict(self, state_dict, strict=True):
        """strict: nn.Module's; otherwise copy what matches and report the rest
        (model_discriminator.py:64-76)."""
        if strict:
            nn.Module.load_state_dict(self, state_dict, strict)
            return
        own_state = self.state_dict()
        for name, param in state_dict.items():
            if name not in own_state:
                continue
            try:
                own_state[name].copy_(param)
            except Exception as e:                                  # noqa: BLE001 (mirrors the reference)
                print('dis: could not load', name, ' - ', e)
