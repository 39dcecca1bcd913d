"""Hand-scheduled forward / backward of the generator family on the gfx950 kernels.

One topology covers SURVEY rows a2-a4: ``Generator`` (model_generator.py:22-101), any stack of
``GeneratorSuffix`` wrappers (model_generator.py:117-141) and the progressive generator
(model_generator_progressive.py:21-65):

    9x9 conv -> PReLU -> n x [conv-BN-PReLU-conv-BN + skip] -> conv-BN (+ long skip) ->
    k x [conv -> PixelShuffle(2) -> PReLU] -> conv -> tanh

Schedule (what crosses HBM): every conv writes its raw output once; BatchNorm-apply and PReLU are
folded into the NEXT conv's tile staging; BatchNorm batch statistics come out of the producing
conv's epilogue; only the block skip-sum is materialised by an elementwise pass.  The backward
mirrors it: BatchNorm/PReLU backward are prologues of the data- and weight-gradient convs; the
only extra passes are the two per-BatchNorm reductions.
"""
import torch

from . import _lib as L
from . import engine as E
from .engine import Operand


class Topology:
    """Where the parameters of the generator family live (filled by the nn.Module wrappers)."""

    def __init__(self):
        self.first = None          # ConvRef (9x9)
        self.first_prelu = None    # Parameter [1]
        self.blocks = []           # dicts: c1, bn1, prelu, c2, bn2
        self.trunk_end = None      # ConvRef
        self.trunk_bn = None
        self.long_skip = True
        self.stages = []           # (ConvRef with shuffle2 geometry, prelu Parameter)
        self.end = None            # ConvRef -> tanh

    def conv_refs(self):
        refs = [self.first]
        for b in self.blocks:
            refs += [b['c1'], b['c2']]
        refs.append(self.trunk_end)
        refs += [s[0] for s in self.stages]
        refs.append(self.end)
        return refs

    def bn_modules(self):
        mods = []
        for b in self.blocks:
            mods += [b['bn1'], b['bn2']]
        mods.append(self.trunk_bn)
        return mods


class Saved:
    pass


def run_forward(topo, x, training):
    """x: NCHW fp32 device tensor.  Returns (out NCHW, Saved).  With ``topo.end is None`` the schedule stops
    before the final conv+tanh (``forward_no_end``, model_generator.py:86-96) and returns the activated
    upscale output as an NCHW tensor."""
    E.require_gpu_tensor(x, 'generator input')
    x = x.contiguous()
    n, cimg, h, w = x.shape
    refs = [r for r in topo.conv_refs() if r is not None]
    items = [(topo.first, n, h, w)]
    for b in topo.blocks:
        items += [(b['c1'], n, h, w), (b['c2'], n, h, w)]
    items.append((topo.trunk_end, n, h, w))
    hh, ww = h, w
    for ref, _ in topo.stages:
        items.append((ref, n, hh, ww))
        hh, ww = hh * 2, ww * 2
    if topo.end is not None:
        items.append((topo.end, n, hh, ww))
    preps, keep = E.prepare_weights(items, training)
    P = {id(r): p for r, p in zip(refs, preps)}
    sv = Saved()
    sv.topo, sv.P, sv.keep, sv.x, sv.training = topo, P, keep, x, training
    sv.blocks = []

    def bn_consts(conv_out, bn):
        """-> (k [4, C], LazyBN | None): in training mode the constants are finalised by whoever applies them first"""
        y, sp, cp = conv_out
        if not training:
            return E.bn_eval_consts(bn), None
        lz = E.LazyBN(sp, cp, bn)
        return lz.k, lz

    # first conv (+ lazily applied PReLU)
    x_op = Operand.plain(x, dims=(n, h, w, cimg), mode=L.X_NCHW)
    t0_pre, _, _ = E.conv_forward(P[id(topo.first)], x_op, bias=topo.first.bias)
    sv.t0_pre = t0_pre
    cur_raw, cur_slope = t0_pre, topo.first_prelu       # current activation = lrelu(cur_raw, cur_slope)
    pending = None                                      # (c2, k2, LazyBN) of the previous block: its skip sum is still to be formed

    def next_input(prep):
        """the operand of the next trunk conv: the previous block's output x + BN2(c2).  Where that conv runs on a
        persistent trunk kernel the sum is formed in ITS staging (and stored once as a side effect) instead of in an
        elementwise pass of its own; either way (cur_raw, None) names the materialised sum afterwards."""
        nonlocal cur_raw, cur_slope, pending
        if pending is not None:
            c2, k2, lz = pending
            pending = None
            if E.trunk_takes_skip_sum(prep, cur_raw, c2):
                out = torch.empty_like(cur_raw)
                op = Operand.res_affine(cur_raw, cur_slope, c2, k2[0], k2[1], out)
                op.fin = lz
                cur_raw, cur_slope = out, None
                return op
            if lz is not None:
                lz.ensure()
            cur_raw, cur_slope = E.eltwise_res_affine(cur_raw, cur_slope, c2, k2[0], k2[1]), None
        return Operand.act(cur_raw, cur_slope) if cur_slope is not None else Operand.plain(cur_raw)

    for b in topo.blocks:
        rec = Saved()
        in_op = next_input(P[id(b['c1'])])
        rec.in_raw, rec.in_slope = cur_raw, cur_slope   # (the materialised input: written by this conv when fused)
        o1 = E.conv_forward(P[id(b['c1'])], in_op, bias=b['c1'].bias, stats=training)
        rec.c1 = o1[0]
        rec.k1, lz1 = bn_consts(o1, b['bn1'])
        a1_op = Operand.affine_act(rec.c1, rec.k1[0], rec.k1[1], b['prelu'])
        a1_op.fin = lz1                                 # bn1 is finalised by conv2's kernel where it can
        o2 = E.conv_forward(P[id(b['c2'])], a1_op, bias=b['c2'].bias, stats=training)
        rec.c2 = o2[0]
        rec.k2, lz2 = bn_consts(o2, b['bn2'])
        pending = (rec.c2, rec.k2, lz2)                 # ... and bn2 by the conv that forms the skip sum
        sv.blocks.append(rec)
    in_op = next_input(P[id(topo.trunk_end)])
    sv.xl_raw, sv.xl_slope = cur_raw, cur_slope
    oe = E.conv_forward(P[id(topo.trunk_end)], in_op, bias=topo.trunk_end.bias, stats=training)
    sv.ce = oe[0]
    sv.ke, lze = bn_consts(oe, topo.trunk_bn)
    if lze is not None:
        lze.ensure()                                    # its consumers are the long-skip sum / the upscale conv: generic kernels
    if topo.long_skip:
        sv.t = E.eltwise_res_affine(t0_pre, topo.first_prelu, sv.ce, sv.ke[0], sv.ke[1])
        cur = Operand.plain(sv.t)
    else:
        cur = Operand.affine_act(sv.ce, sv.ke[0], sv.ke[1], 1.0)
    sv.stage_in, sv.stage_pre = [], []
    for ref, slope in topo.stages:
        sv.stage_in.append(cur)
        pre, _, _ = E.conv_forward(P[id(ref)], cur, bias=ref.bias)       # PixelShuffle on store
        sv.stage_pre.append(pre)
        cur = Operand.act(pre, slope)
    sv.end_in = cur
    if topo.end is None:            # forward_no_end: materialise the (lazy) activation in the reference's NCHW layout
        nn_, hh_, ww_, cc_ = cur.dims
        out = torch.empty((nn_, cc_, hh_, ww_), dtype=torch.float32, device=x.device)
        E.nhwc_to_nchw(cur.x1, out, cc_ * hh_ * ww_, cur.pa, cur.pd, cur.slope if cur.pro != L.PRO_NONE else None)
    else:
        out, _, _ = E.conv_forward(P[id(topo.end)], cur, bias=topo.end.bias, y_mode=L.Y_NCHW, epi=L.EPI_TANH)
    sv.out = out
    if training:
        torch._foreach_add_([m.num_batches_tracked for m in topo.bn_modules()], 1)
    if KEEP_SAVED is not None:
        KEEP_SAVED.append(sv)
    return out, sv


KEEP_SAVED = None       # test hook: a list -> run_forward appends its saved state (pre-activations, BatchNorm constants: the PReLU masks)


def run_backward(sv, grad_out, need_dx, sink=None, params=()):
    """Returns ({id(param): grad}, grad_x or None).  sink (distributed.GradReducer or None): gradients are announced
    to it in buckets as they become final -- after the output / upscale / trunk-end convs, after every
    SINK_BLOCKS residual blocks, after the first conv -- so their all-reduce overlaps the rest of this schedule."""
    if not sv.training:
        raise NotImplementedError('backward through an eval-mode (running-statistics) generator forward '
                                  'is not implemented in the HIP path')
    topo, P = sv.topo, sv.P
    E.require_gpu_tensor(grad_out, 'generator grad_output')
    grad_out = grad_out.contiguous()
    grads = {}
    wg = E.WeightGradBatch()
    pending = E.PendingSlabs()            # slab sums of the weight gradients, carried by the next BatchNorm-backward finish
    # trunk sizes the persistent kernels do not take (24 x 24 maps of a x4 / x8 generator): the weight gradients of the 3x3 layers are
    # collected and launched together at the flush (engine.WgradDeepBatch); their slab sums wait in a list of their own -- a
    # BatchNorm-backward finish must not pick up a slab set whose kernel has not run yet
    wb, pending_b = E.WgradDeepBatch(), E.PendingSlabs()
    by_id = {id(p): p for p in params}
    announced = set()
    all_refs = [r for r in topo.conv_refs() if r is not None]

    def flush(tag):
        """un-pack the weight gradients collected so far (one launch) and announce every new gradient to the sink"""
        wb.run(pending_b)
        pending_b.flush()
        pending.flush()
        for ref_id, (gw, gb) in wg.run().items():
            ref = next(r for r in all_refs if id(r) == ref_id)
            if gw is not None:
                grads[id(ref.weight)] = gw
            if gb is not None:
                grads[id(ref.bias)] = gb
        wg.items = []
        if sink is not None:
            new = [k for k in grads if k not in announced and k in by_id]
            announced.update(new)
            sink.ready([(by_id[k], grads[k]) for k in new], tag)

    def conv_bwd(ref, x_op, dy_op, need_dgrad=True, res=None, y_mode=L.Y_NHWC, bnb=None):
        """weight gradient (batched un-packing at the end) + data gradient.  bnb = (x, consts, slope) names the
        BatchNorm the data gradient arrives at: where the conv kernel can, it emits that BatchNorm's backward
        reductions from its epilogue and (gradient, partial rows) is returned instead of the gradient."""
        p = P[id(ref)]
        want_w, want_b = ref.weight.requires_grad, ref.bias is not None and ref.bias.requires_grad
        if want_w or want_b:
            red = wb.add(p, x_op, dy_op)
            wg.add(p, red if red is not None else E.conv_wgrad(p, x_op, dy_op, defer=pending), want_w, want_b)
        if not need_dgrad:
            return None
        if bnb is None:
            return E.conv_dgrad(p, dy_op, res=res, y_mode=y_mode)
        if E.can_fuse_bn_backward(p):
            return E.conv_dgrad(p, dy_op, res=res, y_mode=y_mode, bnb=bnb)
        return E.conv_dgrad(p, dy_op, res=res, y_mode=y_mode), None

    n = sv.x.shape[0]
    # ---- end conv + tanh (or, for forward_no_end, the NCHW -> NHWC change of the incoming gradient) --------
    ho, wo = sv.out.shape[2], sv.out.shape[3]
    if topo.end is None:
        # no `end` conv: the incoming gradient is that of the last upscale stage's activation -- or, with no stage at all
        # (the bare trunk, model_generator_progressive.py:40-44), of the trunk's BatchNorm output itself
        g = E.nchw_to_nhwc(grad_out, sv.out.shape[1] * ho * wo, n, ho, wo, sv.out.shape[1])
    else:
        dy = Operand(grad_out, (n, ho, wo, sv.out.shape[1]), pro=L.PRO_TANH_BWD, mode=L.X_NCHW, x2=sv.out)
        g = conv_bwd(topo.end, sv.end_in, dy)              # grad wrt the (activated) input of `end`
    # ---- upscale stages, last to first ---------------------------------------------------------------
    part = None
    for k in range(len(topo.stages) - 1, -1, -1):
        ref, slope = topo.stages[k]
        pre = sv.stage_pre[k]
        if slope.requires_grad:
            grads[id(slope)] = E.prelu_slope_grad(g, pre)
        hk, wk, cq = pre.shape[1] // 2, pre.shape[2] // 2, pre.shape[3]
        dy = Operand(g, (n, hk, wk, 4 * cq), pro=L.PRO_ACT_BWD, mode=L.X_UNSHUFFLE2, x2=pre, slope=slope)
        if k == 0:                                          # its data gradient arrives at the trunk's BatchNorm
            g, part = conv_bwd(ref, sv.stage_in[k], dy, bnb=(sv.ce, sv.ke, None))
        else:
            g = conv_bwd(ref, sv.stage_in[k], dy)
    # ---- trunk end: conv + BN (+ long skip) ----------------------------------------------------------
    g_t = g                                               # grad wrt BN_e output (and wrt t0 via the skip)
    bn = topo.trunk_bn
    q, dgam, dbet, _ = E.bn_backward(g_t, sv.ce, sv.ke, bn.weight, part=part, slabs=pending)
    grads[id(bn.weight)], grads[id(bn.bias)] = dgam, dbet
    dy = Operand(g_t, tuple(sv.ce.shape), pro=L.PRO_BNBWD, x2=sv.ce, pa=q[0], pb=q[1], pd=q[2])
    xl_op = Operand.act(sv.xl_raw, sv.xl_slope) if sv.xl_slope is not None else Operand.plain(sv.xl_raw)
    # every data gradient below arrives at the next BatchNorm of the chain (conv_bwd's bnb): that BatchNorm's
    # two backward reductions come out of the producing conv's epilogue, not out of a pass of their own
    rblocks = list(zip(reversed(topo.blocks), reversed(sv.blocks)))
    if rblocks:
        g, part = conv_bwd(topo.trunk_end, xl_op, dy, bnb=(rblocks[0][1].c2, rblocks[0][1].k2, None))
    else:
        g = conv_bwd(topo.trunk_end, xl_op, dy)
    if sink is not None:
        flush('tail')
    # ---- residual blocks, last to first ----------------------------------------------------------------
    for bi, (b, rec) in enumerate(rblocks):
        q2, dgam, dbet, _ = E.bn_backward(g, rec.c2, rec.k2, b['bn2'].weight, part=part, slabs=pending)
        grads[id(b['bn2'].weight)], grads[id(b['bn2'].bias)] = dgam, dbet
        dy2 = Operand(g, tuple(rec.c2.shape), pro=L.PRO_BNBWD, x2=rec.c2, pa=q2[0], pb=q2[1], pd=q2[2])
        a1_op = Operand.affine_act(rec.c1, rec.k1[0], rec.k1[1], b['prelu'])
        g_a1, part = conv_bwd(b['c2'], a1_op, dy2, bnb=(rec.c1, rec.k1, b['prelu']))
        q1, dgam, dbet, dsl = E.bn_backward(g_a1, rec.c1, rec.k1, b['bn1'].weight, slope=b['prelu'], part=part, slabs=pending)
        grads[id(b['bn1'].weight)], grads[id(b['bn1'].bias)] = dgam, dbet
        grads[id(b['prelu'])] = dsl
        dy1 = Operand(g_a1, tuple(rec.c1.shape), pro=L.PRO_BNACT_BWD, x2=rec.c1, pa=q1[0], pb=q1[1],
                      pd=q1[2], ps=rec.k1[0], pt=rec.k1[1], slope=b['prelu'])
        in_op = Operand.act(rec.in_raw, rec.in_slope) if rec.in_slope is not None else Operand.plain(rec.in_raw)
        # + skip gradient, fused in the epilogue
        if bi + 1 < len(rblocks):
            nrec = rblocks[bi + 1][1]
            g, part = conv_bwd(b['c1'], in_op, dy1, res=g, bnb=(nrec.c2, nrec.k2, None))
        else:
            g = conv_bwd(b['c1'], in_op, dy1, res=g)
        if sink is not None and (bi + 1) % SINK_BLOCKS == 0 and bi + 1 < len(rblocks):
            flush('blocks%d' % (bi + 1))
    # ---- first conv + PReLU -----------------------------------------------------------------------------
    g_t0 = E.add(g, g_t) if topo.long_skip else g
    if topo.first_prelu.requires_grad:
        grads[id(topo.first_prelu)] = E.prelu_slope_grad(g_t0, sv.t0_pre)
    dy0 = Operand(g_t0, tuple(sv.t0_pre.shape), pro=L.PRO_ACT_BWD, x2=sv.t0_pre, slope=topo.first_prelu)
    x_op = Operand.plain(sv.x, dims=(n, sv.x.shape[2], sv.x.shape[3], sv.x.shape[1]), mode=L.X_NCHW)
    gx = conv_bwd(topo.first, x_op, dy0, need_dgrad=need_dx, y_mode=L.Y_NCHW)
    flush('final')
    if sink is not None:
        sink.backward_end()
    return grads, gx


SINK_BLOCKS = 4          # residual blocks per announced gradient bucket (16 blocks: tail + 3 + final = 5 buckets)


class GeneratorFunction(torch.autograd.Function):
    """autograd node for a whole generator forward: inputs (x, *parameters) -> image."""

    @staticmethod
    def forward(ctx, topo, training, sink, x, *params):
        out, sv = run_forward(topo, x, training)
        ctx.sv, ctx.params, ctx.sink = sv, params, sink
        return out

    @staticmethod
    def backward(ctx, grad_out):
        grads, gx = run_backward(ctx.sv, grad_out, ctx.needs_input_grad[3], sink=ctx.sink, params=ctx.params)
        ctx.sv = None
        return (None, None, None, gx) + tuple(grads.get(id(p)) if p.requires_grad else None for p in ctx.params)


def generator_apply(topo, module, x):
    params = [p for p in module.parameters()]
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
        # distributed.GradReducer.attach(module) leaves itself here: the backward schedule announces gradients to it
        sink = getattr(module, '_sisr_grad_sink', None)
        return GeneratorFunction.apply(topo, module.training, sink, x, *params)
    out, _ = run_forward(topo, x, module.training)
    return out
