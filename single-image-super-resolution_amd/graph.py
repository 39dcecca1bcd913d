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
import gc
import threading

import torch

from . import _lib as L


class GraphCaptureError(RuntimeError):
    pass


_ACTIVE = None          # the GraphedStep being captured (segment_boundary() talks to it)
_CAPTURE_STREAM = {}    # device index -> THE capture stream of this process


def _capture_stream(dev):
    """One capture stream per device for every GraphedStep of the process (as torch.cuda.graph does).  Autograd
    runs a parameter's AccumulateGrad node on the stream the node was created on, and nodes stay alive across
    captures (any retained loss tensor holds them): with a stream per capture, the second graph of an iteration
    (the G step after the D step: both accumulate into the discriminator's gradients) forks onto the first one's
    capture stream and the graph gets a cross-stream branch -- measured on ROCm 7.2: its replay faults
    (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in the first kernel behind the fork).  One stream = no forks: every
    captured graph is a single chain."""
    if dev not in _CAPTURE_STREAM:
        _CAPTURE_STREAM[dev] = torch.cuda.Stream(device=dev)
    return _CAPTURE_STREAM[dev]


_TRIGGER = {}           # device -> (leaf tensor, its gradient seed) for _Call


class _Call(torch.autograd.Function):
    """identity whose backward calls a host function: a way to run that function on autograd's device thread"""

    @staticmethod
    def forward(ctx, x, fn):
        ctx.fn = fn
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.fn()
        return None, None


def _detach(x):
    if isinstance(x, torch.Tensor):
        return x.detach()
    if isinstance(x, (tuple, list)):
        return type(x)(_detach(v) for v in x)
    if isinstance(x, dict):
        return {k: _detach(v) for k, v in x.items()}
    return x


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
        # A segment boundary inside a backward pass is reached on autograd's device worker thread, and HIP (ROCm 7.2)
        # only lets the thread that began a capture end it (hipErrorStreamCaptureWrongThread, also in relaxed mode).
        # Segmented captures therefore issue EVERY capture begin / end on that worker thread (_on_worker); plain
        # captures begin and end on the calling thread.
        self._via_worker = between is not None
        self._tid = None
        dev = self._dev = torch.cuda.current_device()
        if self._via_worker and dev not in _TRIGGER:
            _TRIGGER[dev] = (torch.zeros(1, device='cuda:%d' % dev, requires_grad=True), torch.ones(1, device='cuda:%d' % dev))
        from . import engine as E
        # warm-up on THE capture stream of the device (not on a stream of its own): a parameter's AccumulateGrad node runs
        # on the stream it was created on and lives as long as anything references the graph behind it, so warm-ups of
        # two GraphedSteps that share parameters (the D step and the G step both accumulate into the discriminator) on
        # two different side streams leave nodes behind whose stream is not the capture stream -- autograd then warns
        # ("AccumulateGrad node's stream does not match") and inserts cross-stream waits into the capture
        side = _capture_stream(dev)
        side.wait_stream(torch.cuda.current_stream())
        per_run = 0
        with torch.cuda.stream(side):
            for _ in range(warmup):
                before = E.table_bytes_staged()
                fn()
                per_run = max(per_run, E.table_bytes_staged() - before)
        # the captured run stages as many descriptor-table bytes as a warm-up run did: have them (twice over) in pinned
        # memory BEFORE the capture begins -- a pinned allocation inside a capture invalidates it
        E.reserve_capture_tables(2 * per_run + (64 << 10))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gc.collect()                       # no tensor of the warm-up runs may be freed in the middle of the capture
        torch.cuda.empty_cache()
        self.graphs, self.tags = [], []
        self._pool = torch.cuda.graph_pool_handle()
        self._stream = _capture_stream(dev)
        self._stream.wait_stream(torch.cuda.current_stream())
        _ACTIVE = self
        try:
            with torch.cuda.stream(self._stream):
                self._on_worker(self._begin)
                try:
                    self.out = _detach(fn())     # static result tensors; the autograd graph behind them is dropped
                    self._on_worker(self._end)
                except BaseException:
                    self._on_worker(self._abandon)
                    raise
        except Exception as e:                                   # noqa: BLE001
            _ACTIVE = None
            self._recover()
            raise GraphCaptureError('HIP graph capture failed: %s: %s' % (type(e).__name__, str(e).splitlines()[0])) from e
        finally:
            _ACTIVE = None
        torch.cuda.current_stream().wait_stream(self._stream)

    # ---- capture plumbing --------------------------------------------------------------------------------
    def _on_worker(self, f):
        """run f() on autograd's worker thread of this device (segmented captures), else right here"""
        if not self._via_worker or threading.get_ident() == self._tid:
            return f()
        x, g = _TRIGGER[self._dev]

        def on_thread():
            self._tid = threading.get_ident()
            f()
        # a backward pass whose only node is _Call: autograd executes the node on the device's worker thread, under
        # the stream that is current here (the capture stream); no kernel is launched by the pass itself
        torch.autograd.backward(_Call.apply(x, on_thread), g)

    def _begin(self):
        from . import engine as E
        E.invalidate_weight_caches()         # images packed before this point are another graph's memory or predate an optimizer step
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=self._pool)
        self.graphs.append(g)

    def _end(self):
        self.graphs[-1].capture_end()

    def _next_segment(self, tag):
        def cut():
            self._end()
            self.tags.append(tag)
            self._begin()
        self._on_worker(cut)

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
