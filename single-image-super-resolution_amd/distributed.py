"""Data-parallel gradient exchange: one process per GPU, ``torch.distributed`` (backend "nccl" is
RCCL on ROCm, over xGMI inside a node).

The reference's only parallelism is ``nn.DataParallel`` (config.py:114-118): batch scattered over
the GPUs, full replicas, per-replica BatchNorm statistics, gradients summed onto device 0.  Here
every rank owns its shard of patches and a full replica (the spectral-norm power iteration is
replica-deterministic given identical W, u, v) and the only exchange is one mean all-reduce of the
parameter gradients per optimizer step, in a few large flat buckets (xGMI is point-to-point: ring
collectives are per-link bound, so few large messages beat many small ones) issued on a side HIP
stream so that it overlaps whatever the compute stream does next.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, params, world_size=None, bucket_bytes=128 << 20, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = world_size if world_size is not None else dist.get_world_size(group)
        self.buckets, cur, size = [], [], 0
        for p in self.params:
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        self._side = None

    def _streams(self, device):
        if device.type != 'cuda':
            return None, None
        if self._side is None:
            self._side = torch.cuda.Stream(device=device)
        return torch.cuda.current_stream(device), self._side

    def all_reduce_mean(self):
        """grad <- mean over ranks, for every parameter that has a gradient.  Returns after the
        compute stream has been made to wait for the exchange (no host synchronisation)."""
        if self.world <= 1:
            return
        for bucket in self.buckets:
            grads = [p.grad for p in bucket if p.grad is not None]
            if not grads:
                continue
            cur, side = self._streams(grads[0].device)
            if side is not None:
                side.wait_stream(cur)                       # gradients are produced on `cur`
                with torch.cuda.stream(side):
                    self._reduce(grads)
                    for g in grads:
                        g.record_stream(side)
                cur.wait_stream(side)
            else:
                self._reduce(grads)

    def _reduce(self, grads):
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.div_(self.world)
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
