"""HIP-graph replay of a launch sequence (a forward+backward pass is a few hundred kernel launches; replayed from
a graph they cost no per-launch host time).  The path's launches are capturable by construction: descriptor tables
travel through pinned staging buffers that outlive the graph (engine._table_to_device), nothing synchronises with
the host, and every tensor a step produces lives in the graph's private pool -- so the values `fn` returns are
static tensors that each replay overwrites.  Optimizer steps stay outside (their scalars change every step)."""
import torch


class GraphedStep:
    """Capture `fn()` (no arguments; it reads static input tensors) once, replay it on every call.

    >>> step = GraphedStep(lambda: fwd_bwd())     # warms up on a side stream, then captures
    >>> loss = step()                              # replays; `loss` is the static tensor of the capture
    """

    def __init__(self, fn, warmup=2):
        dev = torch.cuda.current_device()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn()

    def __call__(self):
        self.graph.replay()
        return self.out
