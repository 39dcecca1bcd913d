"""HIP-graph replay of a launch sequence (a forward+backward pass is a few hundred kernel launches; replayed from
a graph they cost no per-launch host time).  The path's launches are capturable by construction: descriptor tables
travel through pinned staging buffers that outlive the graph (engine._table_to_device), nothing synchronises with
the host, and every tensor a step produces lives in the graph's private pool -- so the values `fn` returns are
static tensors that each replay overwrites.  Optimizer steps stay outside (their scalars change every step).

Thin by design: capture / replay themselves are ``torch.cuda.CUDAGraph`` (hipGraph underneath).  What this module
adds is (a) capture in SEGMENTS -- `fn` may call ``segment_boundary()`` to close the running graph and open the next
one, so that a caller can run ordinary stream work between two parts of one backward pass (the per-bucket gradient
all-reduce of distributed.GradReducer) -- and (b) a failure path that leaves the process usable: a capture that is
invalidated (a host synchronisation inside `fn`, an allocation the pool cannot serve, ...) is abandoned, the device
is synchronised, the HIP runtime's pending error is cleared and ``GraphCaptureError`` is raised with the reason, so
the caller can decide to run eagerly (bench.py does, and says so in its output) instead of dying on the next launch.
"""
import torch

from . import _lib as L


class GraphCaptureError(RuntimeError):
    pass


_ACTIVE = None          # the GraphedStep being captured (segment_boundary() talks to it)


def segment_boundary(tag=None):
    """Called from inside a function being captured by GraphedStep: ends the running graph segment and starts the
    next one.  No-op outside a capture (eager execution) -- returns False then."""
    if _ACTIVE is None:
        return False
    _ACTIVE._next_segment(tag)
    return True


class GraphedStep:
    """Capture `fn()` (no arguments; it reads static input tensors) once, replay it on every call.

    >>> step = GraphedStep(lambda: fwd_bwd())     # warms up on a side stream, then captures
    >>> loss = step()                              # replays; `loss` is the static tensor of the capture

    ``between(tag)`` (optional) is called on the host after each replayed segment except the last, with the tag the
    matching ``segment_boundary(tag)`` call passed: ordinary stream work issued there runs between the segments.
    """

    def __init__(self, fn, warmup=2, between=None):
        global _ACTIVE
        self.between = between
        dev = torch.cuda.current_device()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graphs, self.tags = [], []
        self._pool = torch.cuda.graph_pool_handle()
        self._stream = torch.cuda.Stream(device=dev)
        self._stream.wait_stream(torch.cuda.current_stream())
        _ACTIVE = self
        try:
            with torch.cuda.stream(self._stream):
                self._begin()
                try:
                    self.out = fn()
                    self._end()
                except BaseException:
                    self._abandon()
                    raise
        except Exception as e:                                   # noqa: BLE001
            _ACTIVE = None
            self._recover()
            raise GraphCaptureError('HIP graph capture failed: %s: %s' % (type(e).__name__, str(e).splitlines()[0])) from e
        finally:
            _ACTIVE = None
        torch.cuda.current_stream().wait_stream(self._stream)

    # ---- capture plumbing --------------------------------------------------------------------------------
    def _begin(self):
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=self._pool)
        self.graphs.append(g)

    def _end(self):
        self.graphs[-1].capture_end()

    def _next_segment(self, tag):
        self._end()
        self.tags.append(tag)
        self._begin()

    def _abandon(self):
        try:
            self.graphs[-1].capture_end()        # ends (and reports) an invalidated capture; the graph is dropped
        except Exception:                        # noqa: BLE001
            pass

    def _recover(self):
        self.graphs = []
        try:
            torch.cuda.synchronize()
        except Exception:                        # noqa: BLE001
            pass
        L.lib().sisr_clear_last_error()          # the invalidated capture's code must not reach the next launch

    # ---- replay ---------------------------------------------------------------------------------------------
    def __call__(self):
        last = len(self.graphs) - 1
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < last and self.between is not None:
                self.between(self.tags[i])
        return self.out
