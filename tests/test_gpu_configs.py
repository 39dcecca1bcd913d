"""GPU: every BASELINE.json configuration at its FULL size on the HIP path (the CPU oracle would take minutes to
hours there), checked through size-independent properties -- parity itself is pinned at small sizes by the golden
tests (incl. the 16-block depth case in test_gpu_generator.py):

  cfg2  CelebA 96-crop x2 SRGAN: the JOINT iteration (train.py:45-108) -- x2 G at LR 48 -> 96, D on 3x96x96
        (fc_in 18,432), MaskedVGG(0b00010) "VGG22" content loss -- in the bf16 build, B=16: eager run vs
        HIP-graph replay of the same iteration, and run-to-run determinism;
  cfg3  CelebA 96-crop x4: GeneratorSuffix(Generator([2])) at LR 24 -> 96 with the "VGG54" content loss
        MaskedVGG(0b10000) at HR 96 (config.py:83-88,104);
  cfg4  Flickr 192-crop x4: the same x4 G at LR 48 -> 192 and Discriminator((3,192,192)) with fc_in 73,728
        (a 302 MB FC weight; config.py:81-82);
  cfg5  progressive x8: model_generator_progressive's 64 -> 16 -> 4 channel stack of three suffixes at B=16
        (model_generator_progressive.py:47-65).

Properties: bit-identical replay (every reduction on the path has a fixed order), exact linearity of the backward
pass in the incoming gradient (x2 -> exactly x2 on every gradient), output shapes and ranges (tanh / sigmoid),
finiteness.  fp32 parity build unless stated."""
import pytest
import torch

from gpu_helpers import pkg

pytestmark = pytest.mark.gpu
B = 16
FEATS, STRIDES = [64, 64, 128, 128, 256, 256, 512, 512], [1, 2, 1, 2, 1, 2, 1, 2]      # config.py:81-82


def _snapshot(net):
    return {k: v.clone() for k, v in net.state_dict().items()}


def _fwd_bwd(net, state, x, loss_fn, scale=1.0):
    """one training-mode forward+backward from a fixed spectral-norm / BatchNorm state"""
    net.load_state_dict(state)
    net.zero_grad(set_to_none=True)
    xx = x.clone().requires_grad_(True)
    out = net(xx)
    (loss_fn(out) * scale).backward()
    return out.detach(), xx.grad.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()
                                                    if p.grad is not None}


def _check_properties(net, x, loss_fn, out_shape, out_range):
    state = _snapshot(net)
    out, gx, grads = _fwd_bwd(net, state, x, loss_fn)
    assert tuple(out.shape) == out_shape
    lo, hi = out_range
    assert bool(torch.isfinite(out).all()) and float(out.min()) >= lo and float(out.max()) <= hi
    assert bool(torch.isfinite(gx).all()) and all(bool(torch.isfinite(v).all()) for v in grads.values())
    assert float(gx.abs().max()) > 0 and all(float(v.abs().max()) >= 0 for v in grads.values())
    out2, gx2, grads2 = _fwd_bwd(net, state, x, loss_fn)                    # determinism: bit-identical replay
    assert torch.equal(out, out2) and torch.equal(gx, gx2)
    assert all(torch.equal(grads[k], grads2[k]) for k in grads), [k for k in grads if not torch.equal(grads[k], grads2[k])][:5]
    _, gx3, grads3 = _fwd_bwd(net, state, x, loss_fn, scale=2.0)            # backward linearity (x2 is exact in fp32)
    assert torch.equal(gx3, 2.0 * gx)
    assert all(torch.equal(grads3[k], 2.0 * grads[k]) for k in grads), [k for k in grads if not torch.equal(grads3[k], 2.0 * grads[k])][:5]
    return out, grads


def _rand(shape, seed):
    return (torch.rand(shape, generator=torch.Generator().manual_seed(seed)) * 2 - 1).cuda()


def test_cfg3_x4_suffix_generator_with_vgg54_content_loss():
    mg, mce = pkg('model_generator'), pkg('model_content_extractor')
    torch.manual_seed(0)
    net = mg.GeneratorSuffix(mg.Generator(16, 64, 256, [2], use_sn=True)).cuda().train()      # config.py:79-80,84
    ext = mce.MaskedVGG(0b10000, pretrained=False).cuda()                                      # "VGG54"
    hr, lr = _rand((B, 3, 96, 96), 21), _rand((B, 3, 24, 24), 22)
    assert tuple(ext(hr).shape) == (B, mce.get_size(hr, 0b10000)) == (B, 6 * 6 * 512)
    with torch.no_grad():
        f_real = ext(hr)

    def content_loss(fake):                                                                    # train.py:183-186
        return torch.mean(torch.pow(f_real - ext(fake), 2))
    out, grads = _check_properties(net, lr, content_loss, (B, 3, 96, 96), (-1.0, 1.0))
    assert 'upscale.0.weight_orig' in grads and 'base.end.0.weight_orig' in grads             # suffix re-uses base.end


def test_cfg4_x4_generator_and_discriminator_at_hr192():
    mg, md = pkg('model_generator'), pkg('model_discriminator')
    torch.manual_seed(0)
    net_g = mg.GeneratorSuffix(mg.Generator(16, 64, 256, [2], use_sn=True)).cuda().train()
    lr, r = _rand((B, 3, 48, 48), 31), _rand((B, 3, 192, 192), 32)
    _check_properties(net_g, lr, lambda out: (out * r).sum(), (B, 3, 192, 192), (-1.0, 1.0))
    del net_g
    torch.cuda.empty_cache()
    net_d = md.Discriminator((3, 192, 192), FEATS, STRIDES).cuda().train()
    assert net_d.fc[0].weight.shape == (1024, 73728)                                           # 302 MB fp32
    hr, rd = _rand((B, 3, 192, 192), 33), _rand((B, 1), 34)
    out, grads = _check_properties(net_d, hr, lambda o: (o * rd).sum(), (B, 1), (0.0, 1.0))
    assert bool(((out > 0) & (out < 1)).all())                                                 # sigmoid range


def test_cfg5_progressive_x8_stack():
    mp = pkg('model_generator_progressive')
    torch.manual_seed(0)
    g1 = mp.GeneratorSuffix(mp.GeneratorProgresiveBase(16, 64), 64)                            # x2, 64 -> 16 channels
    g2 = mp.GeneratorSuffix(g1.beginning, 16)                                                  # x4, 16 -> 4
    g3 = mp.GeneratorSuffix(g2.beginning, 4)                                                   # x8, 4 -> 1 ... conv(1 -> 3)
    net = g3.cuda().train()
    lr, r = _rand((B, 3, 24, 24), 41), _rand((B, 3, 192, 192), 42)
    _check_properties(net, lr, lambda out: (out * r).sum(), (B, 3, 192, 192), (-1.0, 1.0))


def _cfg2_setup(seed=0):
    mg, md, mce, ut, op = (pkg('model_generator'), pkg('model_discriminator'), pkg('model_content_extractor'),
                           pkg('utils'), pkg('optim'))
    dev = torch.device('cuda')
    torch.manual_seed(seed)
    net_g = mg.Generator(16, 64, 256, [2], use_sn=True).to(dev).train()
    net_d = md.Discriminator((3, 96, 96), FEATS, STRIDES).to(dev).train()
    ext = mce.MaskedVGG(0b00010, pretrained=False).to(dev)                                     # "VGG22"
    og, od = op.Adam(net_g.parameters(), lr=1e-5), op.Adam(net_d.parameters(), lr=1e-5)
    crit = torch.nn.BCELoss()
    hr = _rand((B, 3, 96, 96), 51)
    ones, red, zeros = torch.ones(B, device=dev), torch.full((B,), .9, device=dev), torch.zeros(B, device=dev)

    def d_part():                                   # train.py:45-74
        lr = ut.lr_from_hr(hr, (48, 48), device=dev)
        fake = net_g(lr)
        net_d.zero_grad()
        err_d = crit(net_d(hr).view(-1), red) + crit(net_d(fake.detach()).view(-1), zeros)
        err_d.backward()
        return err_d

    def g_part():                                   # train.py:82-107 (D already stepped)
        lr = ut.lr_from_hr(hr, (48, 48), device=dev)
        fake = net_g(lr)
        net_g.zero_grad()
        err_g = crit(net_d(fake).view(-1), ones) * 5e-2 + torch.mean(torch.pow(ext(hr) - ext(fake), 2))
        err_g.backward()
        return err_g
    return net_g, net_d, og, od, d_part, g_part


def test_cfg2_joint_srgan_iteration_bf16_eager_vs_graph_replay():
    """the D step + G step of one SRGAN iteration at cfg2's full size in the bf16 build: two eager iterations from the
    same seed are bit-identical; the HIP-graph replay of the iteration (two captured graphs, optimizer steps between
    them) produces the same losses and the same updated parameters as the eager launches, iteration after iteration"""
    E, G = pkg('engine'), pkg('graph')
    E.set_precision('bf16')
    try:
        def eager_run(iters):
            net_g, net_d, og, od, d_part, g_part = _cfg2_setup()
            losses = []
            for _ in range(iters):
                ed = d_part(); od.step(); eg = g_part(); og.step()
                losses.append((float(ed), float(eg)))
            return losses, _snapshot(net_g), _snapshot(net_d)
        l1, g1, d1 = eager_run(3)
        l2, g2, d2 = eager_run(3)
        assert l1 == l2 and all(torch.equal(g1[k], g2[k]) for k in g1) and all(torch.equal(d1[k], d2[k]) for k in d1)
        assert all(0 < ld < 100 and 0 < lg < 100 for ld, lg in l1)                 # finite BCE / feature-MSE losses

        net_g, net_d, og, od, d_part, g_part = _cfg2_setup()
        state_g, state_d = _snapshot(net_g), _snapshot(net_d)
        d_graph, g_graph = G.GraphedStep(d_part), G.GraphedStep(g_part)            # warm-ups advance SN / BN state:
        net_g.load_state_dict(state_g); net_d.load_state_dict(state_d)             # ... restart from the seed state
        og.state.clear(); od.state.clear()
        losses = []
        for _ in range(3):
            ed = d_graph(); od.step(); eg = g_graph(); og.step()
            losses.append((float(ed), float(eg)))
        for (a, b), (c, d) in zip(losses, l1):
            assert abs(a - c) <= 1e-6 * max(1.0, abs(c)) and abs(b - d) <= 1e-6 * max(1.0, abs(d)), (losses, l1)
        gs, ds = _snapshot(net_g), _snapshot(net_d)
        for k in g1:
            if g1[k].is_floating_point():
                assert float((gs[k] - g1[k]).abs().max()) <= 1e-6 * max(1.0, float(g1[k].abs().max())), k
        for k in d1:
            if d1[k].is_floating_point():
                assert float((ds[k] - d1[k]).abs().max()) <= 1e-6 * max(1.0, float(d1[k].abs().max())), k
    finally:
        E.set_precision('fp32')


def test_cfg2_iteration_with_one_generator_forward_eager_vs_segmented_replay():
    """train.py:53,60 runs G ONCE per iteration and reuses `fake` (detached for the D step, with its graph for the G step).  The
    iteration as ONE GraphedStep cut at 'd_step' (the discriminator's Adam step runs between the two replayed segments, the G step
    differentiates through the autograd graph the first segment built) gives the losses and parameters of the eager iteration."""
    E, G = pkg('engine'), pkg('graph')
    mg, md, mce, ut, op = (pkg('model_generator'), pkg('model_discriminator'), pkg('model_content_extractor'),
                           pkg('utils'), pkg('optim'))
    E.set_precision('bf16')
    try:
        def setup():
            dev = torch.device('cuda')
            torch.manual_seed(0)
            net_g = mg.Generator(16, 64, 256, [2], use_sn=True).to(dev).train()
            net_d = md.Discriminator((3, 96, 96), FEATS, STRIDES).to(dev).train()
            ext = mce.MaskedVGG(0b00010, pretrained=False).to(dev)
            og, od = op.Adam(net_g.parameters(), lr=1e-5), op.Adam(net_d.parameters(), lr=1e-5)
            crit = torch.nn.BCELoss()
            hr = _rand((B, 3, 96, 96), 51)
            ones, red, zeros = torch.ones(B, device=dev), torch.full((B,), .9, device=dev), torch.zeros(B, device=dev)

            def both():
                lr = ut.lr_from_hr(hr, (48, 48), device=dev)
                fake = net_g(lr)
                net_d.zero_grad()
                err_d = crit(net_d(hr).view(-1), red) + crit(net_d(fake.detach()).view(-1), zeros)
                err_d.backward()
                if not G.segment_boundary('d_step'):
                    od.step()
                net_g.zero_grad()
                err_g = crit(net_d(fake).view(-1), ones) * 5e-2 + torch.mean(torch.pow(ext(hr) - ext(fake), 2))
                err_g.backward()
                return err_d, err_g
            return net_g, net_d, og, od, both
        net_g, net_d, og, od, both = setup()
        ref_losses = []
        for _ in range(3):
            ed, eg = both()
            og.step()
            ref_losses.append((float(ed), float(eg)))
        g1, d1 = _snapshot(net_g), _snapshot(net_d)
        assert all(0 < a < 100 and 0 < b < 100 for a, b in ref_losses)

        net_g, net_d, og, od, both = setup()
        state_g, state_d = _snapshot(net_g), _snapshot(net_d)
        step = G.GraphedStep(both, between=lambda tag: od.step() if tag == 'd_step' else None)
        assert len(step.graphs) == 2
        net_g.load_state_dict(state_g); net_d.load_state_dict(state_d)             # the warm-ups advanced SN / BN / Adam state
        og.state.clear(); od.state.clear()
        losses = []
        for _ in range(3):
            ed, eg = step()
            og.step()
            losses.append((float(ed), float(eg)))
        for (a, b), (c, d) in zip(losses, ref_losses):
            assert abs(a - c) <= 1e-6 * max(1.0, abs(c)) and abs(b - d) <= 1e-6 * max(1.0, abs(d)), (losses, ref_losses)
        gs, ds = _snapshot(net_g), _snapshot(net_d)
        for k in g1:
            if g1[k].is_floating_point():
                assert float((gs[k] - g1[k]).abs().max()) <= 1e-6 * max(1.0, float(g1[k].abs().max())), k
        for k in d1:
            if d1[k].is_floating_point():
                assert float((ds[k] - d1[k]).abs().max()) <= 1e-6 * max(1.0, float(d1[k].abs().max())), k
    finally:
        E.set_precision('fp32')
