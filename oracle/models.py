"""Functional restatement of the reference's model forwards -- TEST INFRASTRUCTURE ONLY.

Each ``*_forward`` takes a flat ``state`` dict keyed exactly like the reference module's
``state_dict()`` (so reference checkpoints / golden fixtures plug in directly), runs the forward
with plain CPU tensors and returns ``(output, new_buffers)`` where ``new_buffers`` holds the
values the reference's in-place buffer updates would leave behind (spectral-norm ``weight_u`` /
``weight_v``, BatchNorm ``running_mean`` / ``running_var`` / ``num_batches_tracked``).
Autograd on the returned output gives the backward oracle.
"""
import torch

from . import ops


def param_keys(state):
    """Keys that are nn.Parameters in the reference (everything but SN/BN buffers)."""
    skip = ('weight_u', 'weight_v', 'running_mean', 'running_var', 'num_batches_tracked')
    return [k for k in state if not k.endswith(skip)]


class _Ctx:
    def __init__(self, state, training):
        self.s = state
        self.training = training
        self.new = {}

    def conv(self, p, x, stride=1, padding=1):
        """Conv2d at key prefix ``p`` -- spectrally normalised iff ``p.weight_orig`` exists."""
        s = self.s
        if p + '.weight_orig' in s:
            w, u, v = ops.spectral_norm_weight(s[p + '.weight_orig'], s[p + '.weight_u'],
                                               s[p + '.weight_v'], self.training)
            self.new[p + '.weight_u'] = u
            self.new[p + '.weight_v'] = v
        else:
            w = s[p + '.weight']
        return ops.conv2d(x, w, s[p + '.bias'], stride=stride, padding=padding)

    def bn(self, p, x):
        s = self.s
        y, rm, rv = ops.batch_norm(x, s[p + '.weight'], s[p + '.bias'], s[p + '.running_mean'],
                                   s[p + '.running_var'], self.training)
        if self.training:
            self.new[p + '.running_mean'] = rm
            self.new[p + '.running_var'] = rv
            self.new[p + '.num_batches_tracked'] = s[p + '.num_batches_tracked'] + 1
        return y

    def prelu(self, p, x):
        return ops.prelu(x, self.s[p + '.weight'])


def _count(state, prefix):
    """Number of consecutive integer children under ``prefix`` (e.g. block_list.N)."""
    n = 0
    while any(k.startswith('%s%d.' % (prefix, n)) for k in state):
        n += 1
    return n


# ---- Generator (model_generator.py:22-115) --------------------------------------------------
def _gen_no_end(c, p, x, list_scales):
    """Generator.forward_no_end (model_generator.py:86-96); BasicBlock.forward (:16-19)."""
    x = c.conv(p + 'first_layers.0', x, padding=4)                    # :33 9x9 conv
    x = c.prelu(p + 'first_layers.1', x)                              # :34
    residual = x
    for i in range(_count(c.s, p + 'block_list.')):                   # :36
        b = p + 'block_list.%d.layers.' % i
        r = x
        x = c.conv(b + '0', x)                                        # :10
        x = c.bn(b + '1', x)                                          # :11
        x = c.prelu(b + '2', x)                                       # :12
        x = c.conv(b + '3', x)                                        # :13
        x = c.bn(b + '4', x)                                          # :14
        x = r + x                                                     # :19
    x = c.conv(p + 'block_list_end.0', x)                             # :39
    x = c.bn(p + 'block_list_end.1', x)                               # :40
    x = x + residual                                                  # :93
    for i, sc in enumerate(list_scales):                              # :44-49 / :55-60
        u = p + 'upscale.%d.' % i
        x = c.conv(u + '0', x)
        x = ops.pixel_shuffle(x, sc)
        x = c.prelu(u + '2', x)
    return x


def _gen_end(c, p, x):
    return torch.tanh(c.conv(p + 'end.0', x))                         # :50-53 / :61-63


def generator_forward(state, x, list_scales=(2,), training=True, n_suffix=0):
    """Generator.forward (model_generator.py:98-101) wrapped ``n_suffix`` times by
    GeneratorSuffix (model_generator.py:117-141; the suffix re-uses the innermost ``end``)."""
    c = _Ctx(state, training)
    p = 'base.' * n_suffix
    x = _gen_no_end(c, p, x, list_scales)
    for d in range(n_suffix - 1, -1, -1):                             # innermost suffix first
        q = 'base.' * d
        x = c.conv(q + 'upscale.0', x)                                # :123
        x = ops.pixel_shuffle(x, 2)                                   # :125
        x = c.prelu(q + 'upscale.2', x)                               # :126
    return _gen_end(c, p, x), c.new                                   # :128,140


# ---- progressive generator (model_generator_progressive.py:21-65) ---------------------------
def progressive_forward(state, x, n_suffix, training=True):
    """GeneratorSuffix^n_suffix(GeneratorProgresiveBase) (model_generator_progressive.py:47-65).
    Suffix k (outermost = k=0) holds ``beginning = Sequential(prefix, conv, PixelShuffle, PReLU)``;
    the base trunk has NO long skip (:40-44)."""
    assert n_suffix >= 1
    c = _Ctx(state, training)
    # key prefix of the base trunk: 'beginning.0.' for every suffix level
    base = 'beginning.' + '0.' * n_suffix
    x = c.conv(base + 'first_layers.0', x, padding=4)
    x = c.prelu(base + 'first_layers.1', x)
    for i in range(_count(state, base + 'block_list.')):
        b = base + 'block_list.%d.layers.' % i
        r = x
        x = c.conv(b + '0', x)
        x = c.bn(b + '1', x)
        x = c.prelu(b + '2', x)
        x = c.conv(b + '3', x)
        x = c.bn(b + '4', x)
        x = r + x
    x = c.conv(base + 'block_list_end.0', x)
    x = c.bn(base + 'block_list_end.1', x)
    for lvl in range(n_suffix - 1, -1, -1):                           # innermost first
        q = 'beginning.' + '0.' * lvl
        x = c.conv(q + '1', x)                                        # :54
        x = ops.pixel_shuffle(x, 2)                                   # :55
        x = c.prelu(q + '3', x)                                       # :56
    return torch.tanh(c.conv('end.0', x)), c.new                      # :58-60


# ---- Discriminator (model_discriminator.py:18-62) -------------------------------------------
def discriminator_forward(state, x, list_stride, training=True):
    """Discriminator.forward (model_discriminator.py:55-62)."""
    c = _Ctx(state, training)
    n = x.shape[0]
    x = ops.leaky_relu(c.conv('conv.0', x, stride=list_stride[0]))    # :39-40
    for i in range(1, len(list_stride)):                              # :43-44, BasicBlock :9-12
        b = 'conv.2.%d.layers.' % (i - 1)
        x = c.conv(b + '0', x, stride=list_stride[i])
        x = c.bn(b + '1', x)
        x = ops.leaky_relu(x)
    x = x.reshape(n, -1)                                              # :59 (NCHW flatten)
    x = ops.leaky_relu(ops.linear(x, state['fc.0.weight'], state['fc.0.bias']))   # :49-50
    x = torch.sigmoid(ops.linear(x, state['fc.2.weight'], state['fc.2.bias']))    # :52-53
    return x, c.new


# ---- MaskedVGG (model_content_extractor.py:33-60) -------------------------------------------
# torchvision VGG19 ``features`` (cfg "E"): index -> layer.  'M' = MaxPool2d(2,2); a number is
# Conv2d(k3,p1) to that many channels followed by ReLU(inplace=True).  The MaxPool indices this
# produces are (4, 9, 18, 27, 36) == maxPool_indexes (model_content_extractor.py:6).
VGG19_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M',
             512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M')
MAXPOOL_INDEXES = (4, 9, 18, 27, 36)
MAXPOOL_INDEXES_BEFORE_ACT = tuple(i - 1 for i in MAXPOOL_INDEXES)     # :7
LAYERS_SIZE = (64, 128, 256, 512, 512)                                  # :10


def vgg19_feature_layers():
    """[(kind, cin, cout)] for features[0:37]; kind in {'conv','relu','pool'}."""
    layers, cin = [], 3
    for v in VGG19_CFG:
        if v == 'M':
            layers.append(('pool', cin, cin))
        else:
            layers.append(('conv', cin, v))
            layers.append(('relu', v, v))
            cin = v
    return layers


def masked_vgg_kept(mask):
    return [MAXPOOL_INDEXES_BEFORE_ACT[i] for i in range(5) if mask & (1 << i)]   # :41


def masked_vgg_forward(state, x, mask):
    """MaskedVGG.forward (model_content_extractor.py:51-60) over ``features[:kept[-1]]`` (:43).

    ``state`` is keyed like the module's state_dict: ``layers.<idx>.weight`` / ``.bias`` with
    <idx> the 0-based features index.  torchvision's ReLUs are in-place, so a tensor saved at
    1-based position i (a conv output) is overwritten by the ReLU at position i+1 before the
    final ``cat`` -- every tap except the LAST therefore holds its post-ReLU value; the last tap
    (the stack is cut right after that conv) is pre-activation.  Restated explicitly here."""
    kept = masked_vgg_kept(mask)
    layers = vgg19_feature_layers()[:kept[-1]]
    saved = []
    for i, (kind, _, _) in enumerate(layers, 1):
        if kind == 'conv':
            x = ops.conv2d(x, state['layers.%d.weight' % (i - 1)], state['layers.%d.bias' % (i - 1)],
                           padding=1)
        elif kind == 'relu':
            x = torch.clamp(x, min=0)
            if saved and saved[-1][0] == i - 1:
                saved[-1] = (i - 1, x)            # in-place ReLU aliasing of the saved tap
        else:
            x = ops.max_pool2x2(x)
        if i in kept:
            saved.append((i, x))
    return torch.cat([e.reshape(e.shape[0], -1) for _, e in saved], dim=1)


def get_size(hw, mask):
    """model_content_extractor.get_size (:63-73) known-answer."""
    h, w = hw
    return sum((h // 2 ** i) * (w // 2 ** i) * LAYERS_SIZE[i] for i in range(5) if mask & (1 << i))
