"""Loss graph restatement -- TEST INFRASTRUCTURE ONLY.

Follows train.py:128-186 and config.py:107,124-166,184-189: BCELoss (mean reduction, log
clamped at -100 as torch.nn.BCELoss publishes) on D's sigmoid outputs, feature-MSE for the
content loss.  Labels: real 1.0, reduced real 0.9, fake 0.0 (config.py:186-188).
"""
import random

import numpy as np
import torch

REAL_LABEL, REAL_LABEL_REDUCED, FAKE_LABEL = 1.0, 0.9, 0.0
W_ADV_G, W_ADV_D, W_CONTENT, W_IDENTITY = 5e-2, 1.0, 1.0, 10.0      # config.py:136-164


def bce(p, target):
    """torch.nn.BCELoss(): mean(-(t*max(log p,-100) + (1-t)*max(log(1-p),-100)))."""
    t = torch.full_like(p, float(target))
    return -(t * torch.clamp(torch.log(p), min=-100.0)
             + (1.0 - t) * torch.clamp(torch.log(1.0 - p), min=-100.0)).mean()


def adversarial_loss_d(d_real, d_fakes):
    """train.py:128-168: BCE(D(real), 0.9) + sum_k BCE(D(fake_k), 0)."""
    err = bce(d_real.view(-1), REAL_LABEL_REDUCED)
    for d_fake in d_fakes:
        err = err + bce(d_fake.view(-1), FAKE_LABEL)
    return err


def adversarial_loss_g(d_fake):
    """train.py:171-181: BCE(D(fake), 1)."""
    return bce(d_fake.view(-1), REAL_LABEL)


def content_loss_g(feat_real, feat_fake):
    """train.py:183-186: mean((E(real) - E(fake))^2)."""
    return torch.mean(torch.pow(feat_real - feat_fake, 2))


def replay_sample_indices(n_old, ratio):
    """train.py:144-145: np.random.choice(list(range(len(old_fakes))), int(len(old_fakes) * ratio), replace=False)"""
    return [int(i) for i in np.random.choice(list(range(n_old)), int(n_old * ratio), replace=False)]


def replay_store(old_fakes, curr_fake, step, freq, max_len):
    """train.py:66-71 on a plain Python list: keep one batch every `freq` steps; once `max_len` entries are held,
    overwrite a random one (random.randint, both ends included)"""
    if step % freq == 0:
        if len(old_fakes) == max_len:
            old_fakes[random.randint(0, max_len - 1)] = curr_fake
        else:
            old_fakes.append(curr_fake)
    return old_fakes
