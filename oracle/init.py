"""Deterministic synthetic state for parity tests -- TEST INFRASTRUCTURE ONLY.

The reference initialises with torch's default init under a random seed (config.py:192-198) and
module construction consumes RNG in constructor order, so fixtures carry explicit states
instead: every tensor is drawn from a numpy ``RandomState`` keyed by (seed, key name), with
scales close to torch's default init so activations stay well conditioned.
"""
import zlib

import numpy as np
import torch


def synth_tensor(key, shape, seed=0):
    rs = np.random.RandomState((zlib.crc32(key.encode()) + 7919 * seed) % (2 ** 31))
    shape = tuple(shape)
    if key.endswith('num_batches_tracked'):
        return torch.zeros((), dtype=torch.int64)
    if key.endswith('running_mean'):
        return torch.from_numpy(rs.uniform(-0.2, 0.2, shape).astype(np.float32))
    if key.endswith('running_var'):
        return torch.from_numpy(rs.uniform(0.5, 1.5, shape).astype(np.float32))
    if key.endswith(('weight_u', 'weight_v')):
        v = rs.normal(size=shape).astype(np.float32)
        return torch.from_numpy(v / max(np.linalg.norm(v), 1e-12))
    if len(shape) == 1 and shape[0] == 1 and key.endswith('weight'):      # PReLU slope
        return torch.from_numpy(rs.uniform(0.1, 0.4, shape).astype(np.float32))
    if len(shape) == 1 and key.endswith('weight'):                          # BN gamma
        return torch.from_numpy(rs.uniform(0.5, 1.5, shape).astype(np.float32))
    if len(shape) == 1:                                                     # biases / BN beta
        return torch.from_numpy(rs.uniform(-0.1, 0.1, shape).astype(np.float32))
    fan_in = int(np.prod(shape[1:]))
    bound = (3.0 / fan_in) ** 0.5                                           # unit-gain uniform
    return torch.from_numpy(rs.uniform(-bound, bound, shape).astype(np.float32))


def synth_state(shapes, seed=0):
    """shapes: {key: shape} (e.g. from a module's state_dict) -> {key: tensor}."""
    return {k: synth_tensor(k, s, seed) for k, s in shapes.items()}


def synth_input(shape, seed=0):
    rs = np.random.RandomState(1000003 + seed)
    return torch.from_numpy(rs.uniform(-1.0, 1.0, tuple(shape)).astype(np.float32))
