"""Experience replay of old generated batches, kept on the device.

The reference keeps a Python list ``dis_list_old`` of up to ``dis_list_old_len`` (1000) past fake batches,
by default on the CPU (config.py:49-54): every D step moves the current fake batch to the host
(train.py:60-61), presents 1 % of the stored batches to the discriminator after moving each back
(train.py:144-156) and stores / overwrites one entry (train.py:66-71).  With 288 GB of HBM per MI355X the list
lives where it is used: ``DeviceReplayList`` is a drop-in for that list -- ``len()``, indexing, assignment,
``append`` and iteration as train.py uses them -- whose entries are views into ONE preallocated device ring
(1000 batches of 16x3x96x96 fp32 = 1.8 GB; 7.1 GB at 192x192), so storing a batch is one device copy and
presenting it costs nothing.  Set ``dis_list_old_cpu = False`` (config.py:53) and build the list with
``gen_dis_list`` below instead of config.py:323-331; train.py itself is unchanged.

The sampling (train.py:144-145) and the overwrite policy (train.py:66-71) draw from the same host generators as the
reference (``numpy.random.choice`` / ``random.randint``), so a seeded run picks the same entries.
"""
import random

import numpy as np
import torch


class DeviceReplayList:
    """list of past fake batches backed by one device tensor ``[capacity, *batch_shape]``"""

    def __init__(self, capacity, device=None, dtype=torch.float32):
        self.capacity, self.device, self.dtype = int(capacity), device, dtype
        self._ring = None             # allocated at the first append (the batch shape is not known before)
        self._n = 0

    # ---- list protocol (what train.py:66-71,144-156 and utils.py:114 use) ------------------------------------------
    def __len__(self):
        return self._n

    def _slot(self, i):
        if not -self._n <= i < self._n:
            raise IndexError('replay list index out of range')
        return self._ring[i % self._n if i < 0 else i]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._slot(k) for k in range(*i.indices(self._n))]
        return self._slot(int(i))

    def __setitem__(self, i, batch):
        self._slot(int(i)).copy_(batch, non_blocking=True)

    def __iter__(self):
        return (self._ring[k] for k in range(self._n))

    def append(self, batch):
        if self._ring is None:
            dev = self.device if self.device is not None else batch.device
            self._ring = torch.empty((self.capacity,) + tuple(batch.shape), dtype=self.dtype, device=dev)
        if tuple(batch.shape) != tuple(self._ring.shape[1:]):
            raise ValueError('replay list holds batches of shape %s, got %s' % (tuple(self._ring.shape[1:]), tuple(batch.shape)))
        if self._n == self.capacity:
            raise IndexError('replay list is full (%d entries): overwrite an entry instead (train.py:68-69)' % self.capacity)
        self._ring[self._n].copy_(batch, non_blocking=True)
        self._n += 1

    # ---- the reference's policies ------------------------------------------------------------------------------------
    def sample(self, ratio):
        """train.py:144-145: ``int(len * ratio)`` distinct entries, drawn with numpy's global generator"""
        idx = np.random.choice(list(range(self._n)), int(self._n * ratio), replace=False)
        return [self._ring[int(i)] for i in idx]

    def store(self, batch, step, freq=1):
        """train.py:66-71: every ``freq`` steps keep the batch; once full, overwrite a random entry"""
        if step % freq != 0:
            return
        if self._n == self.capacity:
            self[random.randint(0, self.capacity - 1)] = batch
        else:
            self.append(batch)

    # ---- checkpoint interchange (utils.py:108-115 saves the list as 'dis_list') ---------------------------------------
    def to_list(self):
        return [self._ring[k].detach().cpu() for k in range(self._n)]

    @classmethod
    def from_list(cls, batches, capacity, device):
        out = cls(capacity, device)
        for b in batches[:capacity]:
            out.append(b.to(device))
        return out


def gen_dis_list(checkpoint, capacity, device, progressive_gan_suffix=0):
    """config.py:323-331: reuse the checkpoint's old fakes only when their size still fits the generator"""
    old = checkpoint.get('dis_list', []) if progressive_gan_suffix % 2 == 0 else []
    return DeviceReplayList.from_list(list(old), capacity, device)


def adversarial_loss_d(net_d, criterion, real, curr_fake, replay, real_label_reduced, fake_label, ratio=0.01):
    """The D-step loss of train.py:128-168 over ``[curr_fake] + sampled old fakes`` (every batch is its own D forward:
    its own BatchNorm statistics and spectral-norm iteration, as in the reference).  -> (D_G_z1, D_x, errD)"""
    d_real = net_d(real).view(-1)
    err = criterion(d_real, real_label_reduced)
    d_x = d_real.mean()
    d_g_z1 = 0.0
    for fake in [curr_fake] + replay.sample(ratio):
        d_fake = net_d(fake).view(-1)
        err = err + criterion(d_fake, fake_label)
        d_g_z1 = d_g_z1 + d_fake.mean()
    return d_g_z1, d_x, err
