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

``all_reduce_mean()`` is the plain post-backward form (everything that has a ``.grad`` and was not
already reduced during the backward pass): the path for modules used without ``attach`` (e.g. the
discriminator, whose several forward calls per loss accumulate into one gradient).
"""
import torch
import torch.distributed as dist

from . import graph as G


FINAL = 'final'        # tag of the last bucket a backward schedule announces (nothing of the schedule follows it)


class GradReducer:
    def __init__(self, params, world_size=None, bucket_bytes=32 << 20, inplace_bytes=4 << 20, group=None):
        module = params if isinstance(params, torch.nn.Module) else None
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
            self._captured, self._launched = {}, set()

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
        if self.world <= 1 or not pairs:
            return
        if self._capturing:
            self._captured[tag] = ([g for _, g in pairs], [id(p) for p, _ in pairs])
            G.segment_boundary(tag)                     # the capture is cut here; launch_bucket(tag) runs at replay
            return
        self._launch([g for _, g in pairs])
        self._done.update(id(p) for p, _ in pairs)
        self.stats['early_buckets'] += 1

    def backward_end(self):
        """end of the schedule: from here on autograd may read (clone / accumulate) the announced gradients"""
        if not self._capturing:
            self._join()

    # ---- graph replay -----------------------------------------------------------------------------------------
    def launch_bucket(self, tag):
        """between two replayed graph segments: reduce the (static) gradients the finished segment produced"""
        grads, ids = self._captured.get(tag, (None, ()))
        if grads and self.world > 1:
            self._launch(grads, static=True)
            self._launched.add(tag)
            self._done.update(ids)
            self.stats['early_buckets'] += 1
        if tag == FINAL:
            self._join()        # the segment that follows holds autograd's hand-over of ALL gradients to the parameters

    def launch_remaining(self):
        """after the last replayed segment: buckets whose boundary closed the capture (no segment followed them)"""
        for tag, (grads, ids) in self._captured.items():
            if tag not in self._launched and self.world > 1:
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
        elif small:                                      # one message for all of them
            flat = torch.cat([g.reshape(-1) for g in small])
            self._allreduce_mean_(flat)
            torch._foreach_copy_(small, [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in small]), small)])
