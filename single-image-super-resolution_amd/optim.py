"""Fused Adam for the MI355X path.

Drop-in for the optimizer the reference builds, ``optim.Adam(net.parameters(), lr=lr, betas=(.9, 0.999))``
(config.py:292-294), stepped once per network per iteration (train.py:75,108) and driven by ``LambdaLR``
(config.py:170-180; train.py:121-122).  Subclasses ``torch.optim.Adam`` so the constructor, ``param_groups``,
``state_dict()`` / ``load_state_dict()`` (the reference checkpoints ``opti_g`` / ``opti_d``, utils.py:108-115) and
schedulers behave as before; only ``step()`` differs: ONE HIP launch (``sisr_adam_step``) updates every parameter
of a group instead of several elementwise passes per tensor list.  Device fp32 parameters only, no fallback."""
import ctypes as C
import math

import torch

from . import _lib as L
from . import engine as E


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        if amsgrad or kw.get('maximize') or kw.get('capturable') or kw.get('differentiable'):
            raise NotImplementedError('fused Adam: amsgrad / maximize / capturable / differentiable are not implemented '
                                      '(the reference uses none of them, config.py:292-294)')
        kw.pop('foreach', None)
        kw.pop('fused', None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, **kw)
        self._tables = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}                  # the cached descriptor tables point at the replaced moment tensors
        for st in self.state.values():     # 'step' stays a host scalar (torch moves it to the parameter's device)
            if isinstance(st.get('step'), torch.Tensor) and st['step'].is_cuda:
                st['step'] = st['step'].cpu()

    def __setstate__(self, state):
        super().__setstate__(state)
        self._tables = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.lib()
        E.invalidate_weight_caches()      # the kernel below rewrites parameters behind torch's version counters
        for gi, group in enumerate(self.param_groups):
            by_step = {}
            for p in group['params']:
                if p.grad is None:
                    continue
                E.require_gpu_tensor(p, 'fused Adam parameter')
                if p.grad.is_sparse or p.grad.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError('fused Adam: dense contiguous fp32 parameters and gradients expected')
                st = self.state[p]
                if len(st) == 0:          # same state layout as torch.optim.Adam
                    st['step'] = torch.tensor(0.0, dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1
                by_step.setdefault(int(st['step'].item()), []).append(p)     # 'step' lives on the host: no sync
            beta1, beta2 = group['betas']
            for t, plist in by_step.items():
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in plist]
                # the table holds raw pointers of parameters, gradients AND both moment tensors: load_state_dict()
                # replaces the moments (same parameter / gradient addresses), so they are part of the key
                key = (gi, tuple(p.data_ptr() for p in plist), tuple(g.data_ptr() for g in grads),
                       tuple(self.state[p]['exp_avg'].data_ptr() for p in plist),
                       tuple(self.state[p]['exp_avg_sq'].data_ptr() for p in plist))
                cached = self._tables.get(gi)
                if cached is None or cached[0] != key:
                    table = (L.AdamDesc * len(plist))()
                    blocks = 0
                    for d, p, g in zip(table, plist, grads):
                        st = self.state[p]
                        d.p, d.m, d.v, d.g = p.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), g.data_ptr()
                        d.numel, d.block_start = p.numel(), blocks
                        blocks += lib.sisr_adam_blocks(p.numel())
                    cached = (key, E._table_to_device(table, plist[0].device), blocks)
                    self._tables[gi] = cached
                L.check(lib.sisr_adam_step(cached[1].data_ptr(), len(plist), cached[2], float(group['lr']), beta1, beta2,
                                           group['eps'], group['weight_decay'], 1.0 - math.pow(beta1, t),
                                           1.0 - math.pow(beta2, t), E._stream()), 'sisr_adam_step')
                del grads
        return loss
