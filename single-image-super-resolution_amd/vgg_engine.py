"""Hand-scheduled forward / data-gradient of the frozen VGG19 feature stack behind ``MaskedVGG``
(model_content_extractor.py:33-60) on the gfx950 kernels.

Every conv writes its raw output once; ReLU is applied by the consumer while staging tiles;
MaxPool(2,2) pools the raw map (it commutes with ReLU) and its backward is fused with ReLU'.  The
weights are frozen (model_content_extractor.py:46-48): no weight gradient, and the packed weight
images are built once and cached.  Taps are written straight into their slot of the concatenated
``(B, sum)`` feature vector in the reference's NCHW flatten order; every tap except the last is
post-ReLU (torchvision's in-place ReLU aliases the saved tensor), the last is pre-activation.
"""
import torch

from . import _lib as L
from . import engine as E
from .engine import Operand


class Program:
    """convs: list of dicts {ref, pool_before, tap: None | index into the tap list}"""

    def __init__(self, convs, n_taps, last_tap_relu=False):
        self.convs, self.n_taps = convs, n_taps
        # MaskedVGG cuts the stack right after a conv, so its last tap is pre-activation; vgg_4conv_1maxPool keeps the ReLU
        # behind its last conv (features[:9]): its single tap is post-ReLU
        self.last_tap_relu = last_tap_relu
        self._cache = {}

    def prepared(self, n, h, w):
        key = (n, h, w, tuple(c['ref'].weight._version for c in self.convs),
               tuple(c['ref'].weight.data_ptr() for c in self.convs))
        hit = self._cache.get(key)
        if hit is None:
            items, hh, ww = [], h, w
            for c in self.convs:
                if c['pool_before']:
                    hh, ww = hh // 2, ww // 2
                items.append((c['ref'], n, hh, ww))
            hit = E.prepare_weights(items, training=False)
            self._cache = {key: hit}
        return hit[0]


class Saved:
    pass


def run_forward(prog, x):
    E.require_gpu_tensor(x, 'MaskedVGG input')
    x = x.contiguous()
    n, cimg, h, w = x.shape
    preps = prog.prepared(n, h, w)
    sv = Saved()
    sv.prog, sv.preps, sv.x = prog, preps, x
    sv.c = []
    cur = Operand.plain(x, dims=(n, h, w, cimg), mode=L.X_NCHW)
    taps = [None] * prog.n_taps
    for j, (c, p) in enumerate(zip(prog.convs, preps)):
        if j > 0:
            prev = sv.c[j - 1]
            cur = Operand.act(E.maxpool2(prev) if c['pool_before'] else prev, 0.0)   # ReLU applied lazily
        y, _, _ = E.conv_forward(p, cur, bias=c['ref'].bias)
        sv.c.append(y)
        if c['tap'] is not None:
            taps[c['tap']] = j
    sizes = [sv.c[j].shape[1] * sv.c[j].shape[2] * sv.c[j].shape[3] for j in taps]
    total = sum(sizes)
    feat = torch.empty((n, total), dtype=torch.float32, device=x.device)
    off, sv.tap_off = 0, {}
    for t, j in enumerate(taps):
        last = t == len(taps) - 1 and not prog.last_tap_relu
        E.nhwc_to_nchw(sv.c[j], feat[:, off:], total, slope=1.0 if last else 0.0)
        sv.tap_off[j] = (off, last)
        off += sizes[t]
    sv.total = total
    return feat, sv


def run_backward(sv, grad_feat):
    """-> gradient wrt the NCHW input image"""
    prog, preps = sv.prog, sv.preps
    grad_feat = grad_feat.contiguous()
    n = sv.x.shape[0]
    g_next = None                       # grad wrt the activated input of conv j+1
    for j in range(len(prog.convs) - 1, -1, -1):
        c = sv.c[j]
        gt = None
        if j in sv.tap_off:
            off, last = sv.tap_off[j]
            gt = E.nchw_to_nhwc(grad_feat[:, off:], sv.total, n, c.shape[1], c.shape[2], c.shape[3])
        pooled_after = j + 1 < len(prog.convs) and prog.convs[j + 1]['pool_before']
        if g_next is None:                                   # deepest conv: only its tap
            dy = Operand.plain(gt) if last else Operand(gt, tuple(c.shape), pro=L.PRO_ACT_BWD, x2=c, slope=0.0)
        elif pooled_after:
            gc = E.maxpool2_relu_bwd(g_next, c)              # MaxPool' and ReLU' fused
            dy = Operand.plain(E.add_relu_masked(gc, gt, c) if gt is not None else gc)
        elif gt is not None:
            dy = Operand.plain(E.add_relu_masked(E.add_relu_masked(None, g_next, c), gt, c))
        else:
            dy = Operand(g_next, tuple(c.shape), pro=L.PRO_ACT_BWD, x2=c, slope=0.0)
        g_next = E.conv_dgrad(preps[j], dy, y_mode=L.Y_NCHW if j == 0 else L.Y_NHWC)
    return g_next


class VGGFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prog, x):
        feat, sv = run_forward(prog, x)
        ctx.sv = sv
        return feat

    @staticmethod
    def backward(ctx, grad_feat):
        gx = run_backward(ctx.sv, grad_feat)
        ctx.sv = None
        return None, gx


def vgg_apply(prog, x):
    if torch.is_grad_enabled() and x.requires_grad:
        return VGGFunction.apply(prog, x)
    return run_forward(prog, x)[0]
