"""CPU: the device replay list (row f2) behaves like the Python list the reference keeps (config.py:49-54,
train.py:59-71,144-156): same entries after the same store / overwrite sequence and the same sampled indices as the
oracle restatement under the same host seeds; entries are views into one ring (no per-entry allocations)."""
import importlib
import random

import numpy as np
import torch

from oracle import losses as ol

R = importlib.import_module('single-image-super-resolution_amd.replay')


def _batches(n, shape=(2, 3, 4, 4)):
    g = torch.Generator().manual_seed(0)
    return [torch.rand(shape, generator=g) for _ in range(n)]


def test_store_and_overwrite_policy_matches_the_reference_list():
    bs = _batches(9)
    random.seed(5)
    ref = []
    for step, b in enumerate(bs):
        ol.replay_store(ref, b, step, freq=2, max_len=3)          # train.py:66-71
    random.seed(5)
    lst = R.DeviceReplayList(3, device='cpu')
    for step, b in enumerate(bs):
        lst.store(b, step, freq=2)
    assert len(lst) == len(ref) == 3
    assert all(torch.equal(a, b) for a, b in zip(lst, ref))
    base = lst._ring.data_ptr()
    assert [e.data_ptr() for e in lst] == [base + k * bs[0].numel() * 4 for k in range(3)]     # views of one ring
    lst[1] = bs[0]                                                   # list assignment = device copy into the slot
    assert torch.equal(lst[1], bs[0]) and lst[1].data_ptr() == base + bs[0].numel() * 4
    assert torch.equal(lst[-1], ref[-1])


def test_sampling_draws_the_reference_indices():
    bs = _batches(7)
    lst = R.DeviceReplayList(10, device='cpu')
    for b in bs:
        lst.append(b)
    for ratio in (0.01, 0.3, 0.6, 1.0):
        np.random.seed(11)
        want = ol.replay_sample_indices(len(bs), ratio)              # train.py:144-145
        np.random.seed(11)
        got = lst.sample(ratio)
        assert len(got) == int(7 * ratio) == len(want)
        assert all(torch.equal(g, bs[i]) for g, i in zip(got, want))


def test_checkpoint_round_trip_and_size_rule():
    bs = _batches(4)
    lst = R.DeviceReplayList.from_list(bs, 5, 'cpu')
    saved = {'dis_list': lst.to_list()}                              # utils.py:114
    again = R.gen_dis_list(saved, 5, 'cpu', progressive_gan_suffix=2)
    assert len(again) == 4 and all(torch.equal(a, b) for a, b in zip(again, bs))
    assert len(R.gen_dis_list(saved, 5, 'cpu', progressive_gan_suffix=1)) == 0       # config.py:325-330: size changed
