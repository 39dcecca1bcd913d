"""Data-parallel gradient exchange: one process per GPU, ``torch.distributed`` (backend "nccl" is
RCCL on ROCm, over xGMI inside a node).

The reference's only parallelism is ``nn.DataParallel`` (config.py:114-118): batch scattered over
the GPUs, full replicas, per-replica BatchNorm statistics, gradients summed onto device 0.  Here
every rank owns its shard of patches and a full replica (the spectral-norm power iteration is
replica-deterministic given identical W, u, v) and the only exchange is a mean all-reduce of the
parameter gradients per optimizer step.  Equal shards => the mean of the per-replica mean-loss
gradients is the reference's global-batch gradient.

Overlap with backward.  A network's backward pass is ONE hand-written schedule here
(generator_engine.run_backward), so there are no per-parameter autograd hooks to hang the exchange on;
instead the schedule itself announces gradients as they become final, last layers first:
``GradReducer.attach(module)`` makes the module's backward call ``ready(pairs, tag)`` at its flush points
(after the output / upscale / trunk-end convs, after every few residual blocks, after the first conv).
Each announced bucket is reduced IN PLACE on a side HIP stream while the compute stream carries on with
the rest of the backward pass.  xGMI is point-to-point and ring collectives are per-link bound, so a
bucket travels as ONE message: its small tensors are packed into a flat buffer on the side stream,
tensors of ``inplace_bytes`` or more (the discriminator's 75-302 MB FC gradient) are reduced where they
lie -- no pack / unpack pass over them.  The compute stream waits for the side stream once, at the end of
the backward schedule, before autograd hands the gradients to the parameters.

Under HIP-graph replay (graph.GraphedStep) collectives are not captured: the schedule cuts the capture
at each flush point (graph.segment_boundary) and ``launch_bucket(tag)`` issues that bucket's reduction
between two replayed segments -- same overlap, the exchange stays ordinary stream work.

The discriminator's schedule (discriminator_engine.run_backward) announces too: the classifier head first -- its 75-302 MB weight
gradient is 94 % of D's bytes and the FIRST thing the backward pass produces, so its exchange (of the gradient's factors, see below)
has the whole conv stack's backward to hide behind -- then the conv layers deepest first.  D runs one backward pass per forward of the loss (real batch, fake
batch, train.py:132,156); every pass announces its own gradients and autograd adds the reduced passes (mean of a sum = sum of
means).  ``enabled = False`` mutes a reducer for a pass whose gradients are discarded (the G step's pass through D, train.py:174).

The classifier head's weight gradient is NOT reduced at all: it is a rank-16 product ``d1^T x`` of two tensors of 1.2-4.8 MB, so the ranks
all-gather the FACTORS (``gather()``: N x 1.2-4.8 MB instead of 2 (N - 1) / N x 75-302 MB per GPU and pass over per-link-bound xGMI) and
every rank forms the mean gradient itself from all N x 16 rows (``sisr_fc_wgrad_rows``, exact fp32, the same bits on every rank).

``all_reduce_mean()`` is the plain post-backward form (everything that has a ``.grad`` and was not
already reduced during the backward pass): the path for modules used without ``attach``.
"""
import torch
import torch.distributed as dist

from . import graph as G


FINAL = 'final'        # tag of the last bucket a backward schedule announces (nothing of the schedule follows it)


class GradReducer:
    def __init__(self, params, world_size=None, bucket_bytes=32 << 20, inplace_bytes=4 << 20, group=None, name=''):
        module = params if isinstance(params, torch.nn.Module) else None
        # `name` prefixes the tags this reducer hands to graph.segment_boundary(): two reducers captured into one GraphedStep (the
        # discriminator's and the generator's in a joint SRGAN iteration) must not collide on 'final'.  A network that runs several
        # backward passes per sweep (D on the real and on the fake batch, train.py:132,156) announces every pass: each gets its own
        # sequence number, and -- the mean being linear -- reducing the passes separately and letting autograd add them is the mean of
        # the sum (twice the bytes of reducing the sum; the passes are what overlaps with the schedule that produces them).
        self.name = name
        self.enabled = True           # False: ready() is a no-op (the G step's pass through D: those gradients are discarded)
        self._seq = 0
        plist = list(module.parameters()) if module is not None else list(params)
        self.params = [p for p in plist if p.requires_grad]
        self.group = group
        self.world = world_size if world_size is not None else dist.get_world_size(group)
        self.bucket_bytes, self.inplace_bytes = bucket_bytes, inplace_bytes
        # static buckets of the post-backward form, in REVERSE parameter order (the order gradients appear in)
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        self._side = None
        self._flat = {}               # bucket layout -> (flat message buffer, its per-tensor views)
        self._done = set()            # id(param) reduced during the current backward pass
        self._dirty = False           # side stream holds work the compute stream has not waited for
        self._capturing = False
        self._captured = {}           # tag -> [static gradient tensors] recorded while capturing
        self._launched = set()
        self.stats = {'early_buckets': 0, 'late_buckets': 0}
        if module is not None:
            self.attach(module)

    # ---- wiring ----------------------------------------------------------------------------------------------
    def attach(self, module):
        """the module's hand-written backward schedule will announce its gradients to this reducer"""
        module._sisr_grad_sink = self
        return self

    def capture_mode(self, on):
        self._capturing = bool(on)
        if on:
            self._captured, self._launched, self._seq = {}, set(), 0

    def _streams(self, device):
        if device.type != 'cuda':
            return None, None
        if self._side is None:
            self._side = torch.cuda.Stream(device=device)
        return torch.cuda.current_stream(device), self._side

    # ---- called from inside the backward schedule ---------------------------------------------------------------
    def ready(self, pairs, tag):
        """pairs: [(parameter, its final gradient tensor)] produced since the previous call; tag: schedule position"""
        pairs = [(p, g) for p, g in pairs if g is not None and p.requires_grad]
        if self.world <= 1 or not pairs or not self.enabled:
            return
        if self._capturing:
            utag = '%s%s#%d' % (self.name, tag, self._seq)      # unique per announcement (several passes use the same schedule tags)
            if G.segment_boundary(utag):                # the capture is cut here; launch_bucket(utag) runs at replay
                self._seq += 1
                self._captured[utag] = ([g for _, g in pairs], [id(p) for p, _ in pairs], tag == FINAL, 'reduce')
                return
            # (not inside a capture: a warm-up run of the function that is about to be captured -- ordinary eager exchange)
        self._launch([g for _, g in pairs])
        self._done.update(id(p) for p, _ in pairs)
        self.stats['early_buckets'] += 1

    def gather(self, pairs, tag, done=()):
        """pairs: [(local tensor [rows, ...], destination [world * rows, ...])]: all-gather every local tensor into its destination
        (rank-major) on the side stream -- the exchange of FACTORS of a gradient instead of the gradient (the discriminator's first
        Linear: its 75-302 MB weight gradient is a rank-16 product of two tensors of 1.2-4.8 MB; every rank then forms the mean itself
        from all ranks' rows).  Same protocol as ready(): issued at once in eager mode, between two replayed segments under graph
        replay.  done: the parameters whose gradient this exchange stands for (the post-backward pass must not reduce them again).
        The consumer calls wait() before it reads the destinations (eager) / reads them in the segment behind the schedule's LAST
        bucket, which joins the side stream (replay)."""
        if self.world <= 1 or not pairs or not self.enabled:
            return False
        if self._capturing:
            utag = '%s%s#%d' % (self.name, tag, self._seq)
            if G.segment_boundary(utag):
                self._seq += 1
                self._captured[utag] = (list(pairs), [id(p) for p in done], False, 'gather')
                return True
        self._launch_gather(list(pairs))
        self._done.update(id(p) for p in done)
        self.stats['early_buckets'] += 1
        return True

    def wait(self):
        """eager mode: the compute stream waits for everything issued on the side stream so far (no-op inside a capture: there the
        schedule's last bucket joins)"""
        if not torch.cuda.is_available() or not torch.cuda.is_current_stream_capturing():
            self._join()

    def _launch_gather(self, pairs, static=False):
        cur, side = self._streams(pairs[0][0].device)
        if side is None:
            for src, dst in pairs:
                dist.all_gather(list(dst.chunk(self.world, dim=0)), src.contiguous(), group=self.group)
            return
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for src, dst in pairs:
                dist.all_gather(list(dst.chunk(self.world, dim=0)), src.contiguous(), group=self.group)
                if not static:                               # (static = tensors of a graph's private pool: never freed)
                    src.record_stream(side)
                    dst.record_stream(side)
        self._dirty = True

    def backward_end(self):
        """end of the schedule: from here on autograd may read (clone / accumulate) the announced gradients"""
        if not self._capturing and self.enabled:
            self._join()

    def owns(self, tag):
        """the tag of a replayed segment boundary belongs to this reducer"""
        return tag in self._captured

    # ---- graph replay -----------------------------------------------------------------------------------------
    def launch_bucket(self, tag):
        """between two replayed graph segments: reduce the (static) gradients the finished segment produced"""
        grads, ids, last, kind = self._captured.get(tag, (None, (), False, 'reduce'))
        if grads and self.world > 1:
            if kind == 'gather':
                self._launch_gather(grads, static=True)
            else:
                self._launch(grads, static=True)
            self._launched.add(tag)
            self._done.update(ids)
            self.stats['early_buckets'] += 1
        if last:
            self._join()        # the segment that follows holds autograd's hand-over of ALL gradients to the parameters

    def launch_remaining(self):
        """after the last replayed segment: buckets whose boundary closed the capture (no segment followed them)"""
        for tag, (grads, ids, _, kind) in self._captured.items():
            if tag not in self._launched and self.world > 1:
                if kind == 'gather':
                    self._launch_gather(grads, static=True)
                else:
                    self._launch(grads, static=True)
                self._done.update(ids)
        self._launched = set()

    # ---- post-backward ------------------------------------------------------------------------------------------
    def finish(self):
        """reduce every gradient the backward pass did not announce, then make the compute stream wait for the
        whole exchange.  No host synchronisation."""
        self.all_reduce_mean()

    def all_reduce_mean(self):
        """grad <- mean over ranks for every parameter that has a gradient and was not reduced during the
        backward pass.  Returns after the compute stream has been made to wait for the exchange."""
        if self.world > 1:
            for bucket in self.buckets:
                grads = [p.grad for p in bucket if p.grad is not None and id(p) not in self._done]
                if grads:
                    self._launch(grads)
                    self.stats['late_buckets'] += 1
        self._done = set()
        self._join()

    # ---- the exchange ---------------------------------------------------------------------------------------------
    def _launch(self, grads, static=False):
        cur, side = self._streams(grads[0].device)
        if side is None:
            self._reduce(grads)
            return
        side.wait_stream(cur)                            # the gradients were produced on the compute stream
        with torch.cuda.stream(side):
            self._reduce(grads)
            if not static:                               # (static = tensors of a graph's private pool: never freed)
                for g in grads:
                    g.record_stream(side)
        self._dirty = True

    def _join(self):
        if self._dirty and self._side is not None:
            torch.cuda.current_stream(self._side.device).wait_stream(self._side)
        self._dirty = False

    def _allreduce_mean_(self, t):
        if dist.get_backend(self.group) == 'nccl':
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)      # RCCL averages in the collective
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t.mul_(1.0 / self.world)

    def _reduce(self, grads):
        big = [g for g in grads if g.numel() * g.element_size() >= self.inplace_bytes and g.is_contiguous()]
        small = [g for g in grads if not (g.numel() * g.element_size() >= self.inplace_bytes and g.is_contiguous())]
        for g in big:                                    # one message each, reduced where it lies
            self._allreduce_mean_(g)
        if len(small) == 1 and small[0].is_contiguous():
            self._allreduce_mean_(small[0])
        elif small:                                      # one message for all of them, through a PERSISTENT flat buffer
            key = tuple((g.numel(), g.dtype) for g in small)
            ent = self._flat.get(key)
            if ent is None:                              # (one per bucket layout: allocated once, not per step)
                flat = torch.empty(sum(g.numel() for g in small), dtype=small[0].dtype, device=small[0].device)
                ent = self._flat[key] = (flat, list(flat.split([g.numel() for g in small])))
            flat, views = ent
            torch._foreach_copy_(views, [g.reshape(-1) for g in small])
            self._allreduce_mean_(flat)
            torch._foreach_copy_([g.reshape(-1) for g in small], views)
