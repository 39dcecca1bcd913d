"""GPU: progressive growing through checkpoints (row f3; config.py:18-21,83-95, utils.py:108-115,
model_generator.py:65-84,117-141,191): a x2 generator trained and saved in the reference's checkpoint format is
loaded into the x4 architecture the way config.gen_modules does it, and trains on the HIP path."""
import io

import pytest
import torch

from gpu_helpers import pkg
from helpers import grads_close, load_case, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _checkpoint_x2():
    """what utils._save writes (utils.py:108-115) for a x2 SN generator: state taken from the golden fixture that
    was captured from the REFERENCE module, so the key set is the reference's"""
    z, cfg, state, grads, after = load_case('gen_x2_sn_w64')
    buf = io.BytesIO()
    torch.save({'epoch': 3, 'net_g': state, 'net_d': {}, 'opti_g': {}, 'opti_d': {}, 'dis_list': []}, buf)
    buf.seek(0)
    return torch.load(buf, map_location='cpu'), cfg, z


def test_x2_checkpoint_grows_into_x4_and_trains(capsys):
    from oracle import models as om
    mg, op = pkg('model_generator'), pkg('optim')
    ckpt, cfg, z = _checkpoint_x2()
    # progressive_gan_suffix = 1 (config.py:18-21): build x2, load the x2 checkpoint, THEN add the x2 suffix
    net = mg.Generator(cfg['n_blocks'], cfg['nf'], cfg['nl'], cfg['list_scales'], use_sn=cfg['use_sn'])
    net.load_state_dict(ckpt['net_g'], strict=False)                         # config.py:91
    assert capsys.readouterr().out == ''                                     # full coverage: nothing to report
    torch.manual_seed(1)
    net4 = mg.GeneratorSuffix(net)                                           # config.py:95
    sd4 = net4.state_dict()
    assert all(torch.equal(sd4['base.' + k], v) for k, v in ckpt['net_g'].items())
    assert {k for k in sd4 if not k.startswith('base.')} == {'upscale.0.bias', 'upscale.0.weight_orig', 'upscale.0.weight_u',
                                                             'upscale.0.weight_v', 'upscale.2.weight'}
    # one training step on the HIP path against the oracle on the same combined state
    x = torch.from_numpy(z['x'])
    r = torch.rand(x.shape[0], 3, x.shape[2] * 4, x.shape[3] * 4, generator=torch.Generator().manual_seed(2)) - 0.5
    st = {k: v.clone() for k, v in sd4.items()}
    for k in om.param_keys(st):
        st[k].requires_grad_(True)
    out_ref, _ = om.generator_forward(st, x, (2,), True, 1)
    (out_ref * r).sum().backward()
    net4 = net4.cuda().train()
    xd = x.cuda()
    out = net4(xd)
    assert tuple(out.shape) == tuple(out_ref.shape) and rel_err(out.detach().cpu(), out_ref.detach()) < TOL
    (out * r.cuda()).sum().backward()
    got = {k: p.grad.detach().cpu() for k, p in net4.named_parameters()}
    assert grads_close(got, {k: st[k].grad for k in got}, TOL) == []
    # ... and the x4 checkpoint it writes loads back with progressive_gan_suffix = 2 (wrap first, then load)
    saved = {k: v.detach().cpu().clone() for k, v in net4.state_dict().items()}
    fresh = mg.GeneratorSuffix(mg.Generator(cfg['n_blocks'], cfg['nf'], cfg['nl'], cfg['list_scales'], use_sn=cfg['use_sn']))
    fresh.load_state_dict(saved, strict=False)                               # config.py:84,91
    assert all(torch.equal(v, saved[k]) for k, v in fresh.state_dict().items())


def test_frozen_prefix_keeps_the_checkpoint_weights(capsys):
    """_test_gen2 of the reference (model_generator.py:161-184): with freeze_prefix the prefix parameters are unchanged
    by an optimizer step, the suffix parameters move"""
    mg, op = pkg('model_generator'), pkg('optim')
    ckpt, cfg, z = _checkpoint_x2()
    net = mg.Generator(cfg['n_blocks'], cfg['nf'], cfg['nl'], cfg['list_scales'], use_sn=cfg['use_sn'])
    net.load_state_dict(ckpt['net_g'], strict=False)
    g2 = mg.GeneratorSuffix(net, freeze_prefix=True, freeze_upscale=True, freeze_end=True).cuda().train()
    before = {k: p.detach().clone() for k, p in g2.named_parameters()}
    adam = op.Adam([p for p in g2.parameters() if p.requires_grad], lr=.1)
    x = torch.from_numpy(z['x']).cuda()
    loss = torch.sum(torch.pow(g2(x), 2))
    loss.backward()
    adam.step()
    for k, p in g2.named_parameters():
        if k.startswith('base.'):
            assert torch.equal(p, before[k]) and p.grad is None, k
        else:
            assert not torch.equal(p, before[k]), k


def test_sn_checkpoint_into_a_plain_architecture_reports_what_did_not_load(capsys):
    """the caveat of model_generator.py:191 / config.py:62: a checkpoint written with spectral norm holds
    weight_orig/_u/_v; loaded with strict=False into an architecture whose upscale / end convs are plain
    (use_sn=False) those tensors match no key -- the reference's loader reports them instead of failing"""
    mg = pkg('model_generator')
    ckpt, cfg, z = _checkpoint_x2()
    plain = mg.Generator(cfg['n_blocks'], cfg['nf'], cfg['nl'], cfg['list_scales'], use_sn=False)
    plain.load_state_dict(ckpt['net_g'], strict=False)
    rep = capsys.readouterr().out
    assert 'missing' in rep and 'upscale.0.0.weight' in rep and 'unused' in rep and 'upscale.0.0.weight_orig' in rep
    sd = plain.state_dict()
    for k, v in ckpt['net_g'].items():                    # everything that has a namesake was loaded (trunk is always SN)
        if k in sd:
            assert torch.equal(sd[k], v), k
    with pytest.raises(RuntimeError):
        plain.load_state_dict(ckpt['net_g'], strict=True)
