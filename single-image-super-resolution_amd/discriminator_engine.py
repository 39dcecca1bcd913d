"""Hand-scheduled forward / backward of the SRGAN discriminator (SURVEY row a5;
model_discriminator.py:18-62) on the gfx950 kernels:

    SN-conv(3->f0) -> LeakyReLU -> k x [SN-conv(stride 1|2) -> BN -> LeakyReLU] -> flatten (NCHW order)
    -> Linear(fc_in, 2*f_last) -> LeakyReLU -> Linear(., 1) -> Sigmoid

Every conv consumes its producer lazily (BatchNorm-apply + LeakyReLU folded into tile staging) and
emits its own BatchNorm statistics, so no normalisation/activation pass touches HBM; the only
materialisation is the final 16 x fc_in flatten.  The data gradient of the stride-2 convs runs as
four output-parity classes (stride-1 convs with 1/2/2/4 taps) -- no zero-stuffed work.  The FC
layers stream their 75-302 MB weight exactly once per pass.
"""
import os

import torch

from . import _lib as L
from . import engine as E
from .engine import Operand

LEAKY = 0.01          # nn.LeakyReLU() default, model_discriminator.py:12,40,50


class Topology:
    def __init__(self):
        self.conv0 = None        # ConvRef
        self.blocks = []         # (ConvRef, bn module)
        self.fc1 = None          # nn.Linear holders
        self.fc2 = None
        self.cache = None        # dict owned by the module: packed conv_deep.hip images kept between optimizer steps

    def conv_refs(self):
        return [self.conv0] + [b[0] for b in self.blocks]


class Saved:
    pass


KEEP_SAVED = None       # test hook: a list -> run_forward appends its saved state (raw conv outputs, BatchNorm constants: the activation masks)


def run_forward(topo, x, training):
    E.require_gpu_tensor(x, 'discriminator input')
    x = x.contiguous()
    n, cimg, h, w = x.shape
    refs = topo.conv_refs()
    items, hh, ww = [], h, w
    for ref in refs:
        items.append((ref, n, hh, ww))
        hh, ww = ref.geom.out_hw(hh, ww)
    preps, keep = E.prepare_weights(items, training, cache=topo.cache)
    P = {id(r): p for r, p in zip(refs, preps)}
    sv = Saved()
    sv.topo, sv.P, sv.keep, sv.x, sv.training = topo, P, keep, x, training
    x_op = Operand.plain(x, dims=(n, h, w, cimg), mode=L.X_NCHW)
    c0, _, _ = E.conv_forward(P[id(topo.conv0)], x_op, bias=topo.conv0.bias)
    sv.c0 = c0
    cur = Operand.act(c0, LEAKY)
    sv.ins, sv.cs, sv.ks = [], [], []
    for ref, bn in topo.blocks:
        sv.ins.append(cur)
        o = E.conv_forward(P[id(ref)], cur, bias=ref.bias, stats=training)
        k = E.bn_finalize(o[1], o[2], bn) if training else E.bn_eval_consts(bn)
        sv.cs.append(o[0])
        sv.ks.append(k)
        cur = Operand.affine_act(o[0], k[0], k[1], LEAKY)
    last = cur.x1
    fc_in = last.shape[1] * last.shape[2] * last.shape[3]
    flat = torch.empty((n, fc_in), dtype=torch.float32, device=x.device)
    E.nhwc_to_nchw(last, flat, fc_in, cur.pa, cur.pd, LEAKY)            # x.view(B, fc_in) of NCHW
    sv.flat, sv.last_shape = flat, tuple(last.shape)
    sv.head = E.fc_head_ok(n, fc_in, topo.fc1.weight.shape[0]) and topo.fc2.weight.shape[0] == 1
    if sv.head:                                                          # fc_head.hip: W1 streamed once on the fp32 matrix cores
        sv.h1, sv.out = E.fc_head_forward(flat, topo.fc1.weight, topo.fc1.bias, topo.fc2.weight, topo.fc2.bias, LEAKY)
    else:
        sv.h1 = E.fc_forward(flat, topo.fc1.weight, topo.fc1.bias)       # pre-activation
        sv.out = E.fc_forward(sv.h1, topo.fc2.weight, topo.fc2.bias, in_slope=LEAKY, sigmoid=True)
    if training and topo.blocks:
        torch._foreach_add_([bn.num_batches_tracked for _, bn in topo.blocks], 1)
    if KEEP_SAVED is not None:
        KEEP_SAVED.append(sv)
    return sv.out, sv


SINK_CONVS = 3          # conv layers per announced gradient bucket (8 convs: fc + 3 + 3 + final = 4 buckets)


def run_backward(sv, grad_out, need_dx, sink=None, params=()):
    """Returns ({id(param): grad}, grad_x or None).  sink (distributed.GradReducer or None): the schedule announces gradients
    in buckets the moment they are final -- the classifier head first (its 75-302 MB weight gradient is 94 % of D's bytes and
    the FIRST thing this schedule produces: its all-reduce has the whole conv stack's backward to hide behind), then the conv
    layers deepest first in groups of SINK_CONVS (config.py:114-118's DataParallel gradient sum is what this replaces)."""
    if not sv.training:
        raise NotImplementedError('backward through an eval-mode discriminator forward is not implemented')
    topo, P = sv.topo, sv.P
    grads = {}
    wg = E.WeightGradBatch()
    pending = E.PendingSlabs()            # the layers' slab reductions wait for the flush: one launch instead of one per layer
    wb = E.WgradDeepBatch()               # ... and so do the weight-gradient kernels of the 3x3 stack themselves (one launch per stride)
    grad_out = grad_out.contiguous()
    n = sv.x.shape[0]
    by_id = {id(p): p for p in params}
    announced = set()
    refs = topo.conv_refs()

    def flush(tag):
        """un-pack the weight gradients collected so far (one launch) and announce every new gradient to the sink"""
        wb.run(pending)
        pending.flush()
        for ref_id, (gw, gb) in wg.run().items():
            ref = next(r for r in refs if id(r) == ref_id)
            if gw is not None:
                grads[id(ref.weight)] = gw
            if gb is not None:
                grads[id(ref.bias)] = gb
        wg.items = []
        if sink is not None:
            new = [k for k in grads if k not in announced and k in by_id]
            announced.update(new)
            sink.ready([(by_id[k], grads[k]) for k in new], tag)

    factored = None
    if sv.head:
        d1, dw2, db2, db1 = E.fc_head_backward(grad_out, sv.out, sv.h1, topo.fc2.weight, LEAKY)
        world = getattr(sink, 'world', 1) if sink is not None else 1
        if (world > 1 and getattr(sink, 'enabled', False) and hasattr(sink, 'gather') and id(topo.fc1.weight) in by_id
                and E.fc_wgrad_rows_ok(world * n, sv.flat.shape[1], topo.fc1.weight.shape[0]) and os.environ.get('SISR_FC_FACTORED', '1') != '0'):
            # data-parallel: the 75-302 MB weight gradient is d1^T x, a rank-16 product -- the ranks exchange the two FACTORS (an
            # all-gather of N x 1.2-4.8 MB, issued now: it has the whole conv stack's backward to hide behind) and every rank forms
            # the MEAN gradient itself from all ranks' rows at the end of this schedule (same rows, same kernel: the same bits on
            # every rank).  Over point-to-point xGMI an all-reduce of the product would cost 2 (N - 1) / N of it per GPU and pass.
            d1_all = torch.empty((world * n, d1.shape[1]), dtype=d1.dtype, device=d1.device)
            x_all = torch.empty((world * n, sv.flat.shape[1]), dtype=sv.flat.dtype, device=d1.device)
            if sink.gather([(d1, d1_all), (sv.flat, x_all)], 'fc1_factors', done=[topo.fc1.weight]):
                factored = (d1_all, x_all, world)
        dw1 = E.fc_wgrad_only(d1, sv.flat, topo.fc1.weight) if factored is None else None
    else:
        d2 = E.act_bwd(grad_out, sv.out, 1)                               # sigmoid'
        dx2, dw2, db2 = E.fc_backward(d2, sv.h1, topo.fc2.weight, in_slope=LEAKY)
        d1 = E.act_bwd(dx2, sv.h1, 0, LEAKY)                              # LeakyReLU'
        dflat, dw1, db1 = E.fc_backward(d1, sv.flat, topo.fc1.weight)
    grads[id(topo.fc2.weight)], grads[id(topo.fc2.bias)] = dw2, db2
    grads[id(topo.fc1.bias)] = db1
    if dw1 is not None:
        grads[id(topo.fc1.weight)] = dw1
    if sink is not None:
        flush('fc')                                                       # ... before the data gradient of the head is even launched
    if sv.head:
        dflat = E.fc1_dgrad(d1, topo.fc1.weight)
    _, hl, wl, cl = sv.last_shape
    g = E.nchw_to_nhwc(dflat, dflat.shape[1], n, hl, wl, cl)              # grad wrt activated last map

    def conv_bwd(ref, x_op, dy_op, need_dgrad=True, y_mode=L.Y_NHWC, bnb=None):
        """weight gradient + data gradient; bnb = (x, consts, slope) names the BatchNorm (and the LeakyReLU behind it) the data
        gradient arrives at: where the conv kernel can (conv_deep.hip), that BatchNorm's backward reductions come out of its
        epilogue and (gradient, partial rows) is returned"""
        p = P[id(ref)]
        want_w, want_b = ref.weight.requires_grad, ref.bias is not None and ref.bias.requires_grad
        if want_w or want_b:
            red = wb.add(p, x_op, dy_op)
            wg.add(p, red if red is not None else E.conv_wgrad(p, x_op, dy_op, defer=pending), want_w, want_b)
        if not need_dgrad:
            return None
        if bnb is None:
            return E.conv_dgrad(p, dy_op, y_mode=y_mode)
        if E.can_fuse_bn_backward(p):
            return E.conv_dgrad(p, dy_op, y_mode=y_mode, bnb=bnb)
        return E.conv_dgrad(p, dy_op, y_mode=y_mode), None

    part, done = None, 0
    for i in range(len(topo.blocks) - 1, -1, -1):
        ref, bn = topo.blocks[i]
        c, k = sv.cs[i], sv.ks[i]
        q, dgam, dbet, _ = E.bn_backward(g, c, k, bn.weight, slope=LEAKY, part=part)
        grads[id(bn.weight)], grads[id(bn.bias)] = dgam, dbet
        dy = Operand(g, tuple(c.shape), pro=L.PRO_BNACT_BWD, x2=c, pa=q[0], pb=q[1], pd=q[2], ps=k[0], pt=k[1],
                     slope=LEAKY)
        if i > 0:                       # the data gradient arrives at the previous block's BatchNorm + LeakyReLU
            g, part = conv_bwd(ref, sv.ins[i], dy, bnb=(sv.cs[i - 1], sv.ks[i - 1], LEAKY))
        else:
            g, part = conv_bwd(ref, sv.ins[i], dy), None
        done += 1
        if sink is not None and done % SINK_CONVS == 0 and i > 0:
            flush('convs%d' % done)
    dy0 = Operand(g, tuple(sv.c0.shape), pro=L.PRO_ACT_BWD, x2=sv.c0, slope=LEAKY)
    x_op = Operand.plain(sv.x, dims=(n, sv.x.shape[2], sv.x.shape[3], sv.x.shape[1]), mode=L.X_NCHW)
    if need_dx and topo.conv0.geom.stride != 1:
        raise NotImplementedError('input gradient through a stride-2 first convolution')
    gx = conv_bwd(topo.conv0, x_op, dy0, need_dgrad=need_dx, y_mode=L.Y_NCHW)
    flush('final')
    if factored is not None:
        # (replay: this runs in the segment behind the 'final' bucket, which joined the side stream; eager: wait here)
        sink.wait()
        announced.add(id(topo.fc1.weight))
        grads[id(topo.fc1.weight)] = E.fc_wgrad_rows(factored[0], factored[1], topo.fc1.weight, 1.0 / factored[2])
    if sink is not None:
        sink.backward_end()
    return grads, gx


class DiscriminatorFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, topo, training, sink, x, *params):
        out, sv = run_forward(topo, x, training)
        ctx.sv, ctx.params, ctx.sink = sv, params, sink
        return out

    @staticmethod
    def backward(ctx, grad_out):
        grads, gx = run_backward(ctx.sv, grad_out, ctx.needs_input_grad[3], sink=ctx.sink, params=ctx.params)
        return (None, None, None, gx) + tuple(grads.get(id(p)) if p.requires_grad else None for p in ctx.params)


def discriminator_apply(topo, module, x):
    params = list(module.parameters())
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
        # distributed.GradReducer.attach(module) leaves itself here: the backward schedule announces gradients to it
        sink = getattr(module, '_sisr_grad_sink', None)
        return DiscriminatorFunction.apply(topo, module.training, sink, x, *params)
    out, _ = run_forward(topo, x, module.training)
    return out
