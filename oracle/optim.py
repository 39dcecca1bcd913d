"""ORACLE (test infrastructure only) -- numpy restatement of the optimizer step the reference takes.

The reference builds ``optim.Adam(net.parameters(), lr=lr, betas=(.9, 0.999))`` (config.py:292-294), steps it at
train.py:75 (D) and train.py:108 (G), and scales lr with ``LambdaLR(lr_lambda=lambda it: f ** it)``
(config.py:170-180; stepped at train.py:121-122).  torch.optim.Adam is a dependency of the reference, not part of
it; its documented update for amsgrad=False, maximize=False is restated here and pinned in the tests against
torch.optim.Adam itself (which IS importable in this container)."""
import numpy as np


def adam_step(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """one update, step count t (1-based) -> (p, m, v); float32 arithmetic like the device kernel"""
    f = np.float32
    p, g, m, v = (np.asarray(a, dtype=np.float32) for a in (p, g, m, v))
    g = g + f(weight_decay) * p
    m = m + (g - m) * f(1.0 - beta1)
    v = v * f(beta2) + f(1.0 - beta2) * g * g
    bc1, bc2 = 1.0 - beta1 ** t, 1.0 - beta2 ** t
    denom = np.sqrt(v) * f(1.0 / np.sqrt(bc2)) + f(eps)       # host scalars: double arithmetic, one rounding to fp32
    return p - f(lr / bc1) * (m / denom), m, v


def lambda_lr(base_lr, ratio, total_iters, it):
    """config.py:170-180: lr(it) = base_lr * f**it with f = ratio ** (1 / total_iters)"""
    return base_lr * (ratio ** (1.0 / total_iters)) ** it
