"""Experience replay of old generated batches, kept on the device.

The reference keeps a Python list ``dis_list_old`` of up to ``dis_list_old_len`` (1000) past fake batches,
by default on the CPU (config.py:49-54): every D step moves the current fake batch to the host
(train.py:60-61), presents 1 % of the stored batches to the discriminator after moving each back
(train.py:144-156) and stores / overwrites one entry (train.py:66-71).  With 288 GB of HBM per MI355X the list
lives where it is used: ``DeviceReplayList`` is a drop-in for that list -- ``len()``, indexing, assignment,
``append`` and iteration as train.py uses them -- whose entries are views into ONE preallocated device ring
(1000 + 1 batches of 16x3x96x96 fp32 = 1.8 GB; 7.1 GB at 192x192), so storing a batch is one device copy and
presenting it costs nothing.  Set ``dis_list_old_cpu = False`` (config.py:53) and build the list with
``gen_dis_list`` below instead of config.py:323-331; train.py itself is unchanged.

The sampling (train.py:144-145) and the overwrite policy (train.py:66-71) draw from the same host generators as the
reference (``numpy.random.choice`` / ``random.randint``), so a seeded run picks the same entries.

Two properties of the reference's plain list that the ring must not lose:

* **Assignment rebinds, it never mutates -- and it snapshots.**  train.py runs the D forwards on the sampled entries
  (train.py:64), THEN assigns ``dis_list_old[randint] = curr_fake`` (train.py:68-69), THEN calls ``errD.backward()``
  (train.py:74).  The discriminator's backward reads its saved inputs, so writing into the ring row of the assigned slot
  would hand the weight gradient of the first conv the NEW batch whenever that slot was one of those just sampled.  And the
  assigned tensor may be a HIP graph's static output buffer (``graph.GraphedStep`` returns the same tensors on every replay), so
  merely keeping a reference until later would store the NEXT iteration's fake.  The ring therefore has one row more than its
  capacity: an assignment copies the batch into the spare row at once (stream-ordered behind the forwards already issued),
  points the slot at that row, and the row the slot had -- which a pending backward may still read -- is quarantined until the
  next read access of the list (sample / index / iterate / append / save: the next iteration) before it becomes the spare.
  A second assignment before that read access finds no spare row and keeps a private CLONE of the batch, copied in at the
  read access (``flush()`` forces both).
* **The checkpoint holds a plain list.**  utils.py:108-115 pickles the object it was given as ``'dis_list'``;
  ``__reduce__`` makes that a Python list of the held batches as CPU tensors -- the reference's own format, readable
  without this package (``torch.load(..., weights_only=True)`` additionally needs
  ``torch.serialization.add_safe_globals([list])``: a reduced list is a call of ``builtins.list``).
"""
import random

import numpy as np
import torch


class DeviceReplayList:
    """list of past fake batches backed by one device tensor ``[capacity, *batch_shape]``"""

    def __init__(self, capacity, device=None, dtype=torch.float32):
        self.capacity, self.device, self.dtype = int(capacity), device, dtype
        self._ring = None             # [capacity + 1, *batch_shape], allocated at the first append (shape unknown before)
        self._n = 0
        self._map = list(range(self.capacity))      # slot -> ring row
        self._spare = [self.capacity]               # free ring rows
        self._quarantine = []                       # rows released by an assignment since the last read access
        self._pending = {}            # slot -> private clone: assignments that found no spare row (see module docstring)

    def flush(self):
        """read-access point: rows released by earlier assignments become spare, cloned assignments are copied in"""
        if self._pending:
            todo, self._pending = self._pending, {}
            for i, batch in todo.items():
                self._ring[self._map[i]].copy_(batch, non_blocking=True)
        if self._quarantine:
            self._spare += self._quarantine
            self._quarantine = []

    def _row(self, k):
        return self._ring[self._map[k]]

    # ---- list protocol (what train.py:66-71,144-156 and utils.py:114 use) ------------------------------------------
    def __len__(self):
        return self._n

    def _index(self, i):
        i = int(i)
        if not -self._n <= i < self._n:
            raise IndexError('replay list index out of range')
        return i % self._n if i < 0 else i

    def __getitem__(self, i):
        self.flush()
        if isinstance(i, slice):
            return [self._row(k) for k in range(*i.indices(self._n))]
        return self._row(self._index(i))

    def __setitem__(self, i, batch):
        i = self._index(i)
        self._check_shape(batch)
        if self._spare:
            r = self._spare.pop()
            self._ring[r].copy_(batch.detach(), non_blocking=True)     # snapshot NOW: the caller's buffer may be a graph's static output
            self._quarantine.append(self._map[i])                       # the old row may still be read by a pending backward
            self._map[i] = r
            self._pending.pop(i, None)
        else:
            self._pending[i] = batch.detach().clone()                   # (a second assignment to the same slot replaces the first)

    def __iter__(self):
        self.flush()
        return (self._row(k) for k in range(self._n))

    def _check_shape(self, batch):
        if self._ring is not None and tuple(batch.shape) != tuple(self._ring.shape[1:]):
            raise ValueError('replay list holds batches of shape %s, got %s' % (tuple(self._ring.shape[1:]), tuple(batch.shape)))

    def append(self, batch):
        if self._ring is None:
            dev = self.device if self.device is not None else batch.device
            self._ring = torch.empty((self.capacity + 1,) + tuple(batch.shape), dtype=self.dtype, device=dev)
        self._check_shape(batch)
        if self._n == self.capacity:
            raise IndexError('replay list is full (%d entries): overwrite an entry instead (train.py:68-69)' % self.capacity)
        self.flush()
        self._row(self._n).copy_(batch, non_blocking=True)       # (a slot nobody has seen yet: nothing can be reading it)
        self._n += 1

    # ---- the reference's policies ------------------------------------------------------------------------------------
    def sample(self, ratio):
        """train.py:144-145: ``int(len * ratio)`` distinct entries, drawn with numpy's global generator"""
        self.flush()
        idx = np.random.choice(list(range(self._n)), int(self._n * ratio), replace=False)
        return [self._row(int(i)) for i in idx]

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
        self.flush()
        # (own storage per entry: a view of a host-resident ring would drag the whole ring into torch.save)
        return [self._row(k).detach().to('cpu', copy=True) for k in range(self._n)]

    def __reduce__(self):
        """pickled (torch.save of utils.py:108-115) as a plain list of the held batches on the CPU: the reference's own
        checkpoint format, not this class and not the preallocated ring"""
        return (list, (self.to_list(),))

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
