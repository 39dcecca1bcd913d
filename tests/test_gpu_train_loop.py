"""GPU: several iterations shaped exactly like the reference's train_loop (train.py:45-122) on the drop-ins installed under
the reference's module names (``install(fused_adam=True)``): lr_from_hr, G forward, the D step with the experience-replay
list (adversarial_loss_d over the current and the sampled old fakes, store / overwrite policy), optimizerD.step(), the G
step (adversarial + VGG content loss), optimizerG.step(), the three ``.item()`` reads and both LambdaLR scheduler steps --
against the same loop on the CPU oracle with torch's own Adam / LambdaLR: every iteration's three losses and D outputs,
and the parameters after the last step."""
import importlib
import random
import sys

import numpy as np
import pytest
import torch

from gpu_helpers import PKG, pkg
from helpers import analytically_zero, rel_err

pytestmark = pytest.mark.gpu
FEATS, STRIDES = [16, 16, 32, 32], [1, 2, 1, 2]
ITERS, BS, HR, LR, MASK = 4, 8, 32, 16, 0b00011
LR0, RATIO, TOTAL = 1e-4, 0.1, 40        # config.py:38 uses 1e-5; a larger rate makes the parameter comparison meaningful
REPLAY_LEN, REPLAY_RATIO = 2, 0.5        # config.py:50-52 (1000, 0.01) scaled down so that sampling AND overwriting occur


def _batches():
    g = torch.Generator().manual_seed(13)
    return [torch.rand(BS, 3, HR, HR, generator=g) * 2 - 1 for _ in range(ITERS)]


def _oracle_loop(g_state, d_state, v_state):
    from oracle import models as om, ops as oo, losses as ol
    g_state = {k: v.clone() for k, v in g_state.items()}
    d_state = {k: v.clone() for k, v in d_state.items()}
    for st in (g_state, d_state):
        for k in om.param_keys(st):
            st[k].requires_grad_(True)
    adam = pkg('optim').Adam.__mro__[1]                 # torch's own Adam (install(fused_adam=True) has re-pointed torch.optim.Adam)
    og = adam([g_state[k] for k in om.param_keys(g_state)], lr=LR0, betas=(.9, .999))                    # config.py:293
    od = adam([d_state[k] for k in om.param_keys(d_state)], lr=LR0, betas=(.9, .999))                    # config.py:294
    f = RATIO ** (1.0 / TOTAL)
    sg = torch.optim.lr_scheduler.LambdaLR(og, lr_lambda=lambda it: f ** it)                              # config.py:170-180
    sd = torch.optim.lr_scheduler.LambdaLR(od, lr_lambda=lambda it: f ** it)
    old, log = [], []
    for i, hr in enumerate(_batches()):
        img_lr = oo.lr_from_hr(hr, (LR, LR))
        fake, new = om.generator_forward(g_state, img_lr, (2,), True, 0)
        with torch.no_grad():
            for k, v in new.items():
                g_state[k] = v
        od.zero_grad()
        curr = fake.detach()
        d_real, new = om.discriminator_forward(d_state, hr, STRIDES, True)
        d_state.update(new)
        d_fakes = []
        for fk in [curr] + [old[j] for j in ol.replay_sample_indices(len(old), REPLAY_RATIO)]:
            d_f, new = om.discriminator_forward(d_state, fk, STRIDES, True)
            d_state.update(new)
            d_fakes.append(d_f)
        err_d = ol.adversarial_loss_d(d_real, d_fakes) * ol.W_ADV_D
        ol.replay_store(old, curr, i, 1, REPLAY_LEN)
        err_d.backward()
        od.step()
        og.zero_grad()
        d_out, new = om.discriminator_forward(d_state, fake, STRIDES, True)
        d_state.update(new)
        err_adv = ol.adversarial_loss_g(d_out) * ol.W_ADV_G
        with torch.no_grad():
            f_real = om.masked_vgg_forward(v_state, hr, MASK)
        err_cont = ol.content_loss_g(f_real, om.masked_vgg_forward(v_state, fake, MASK)) * ol.W_CONTENT
        (err_adv + err_cont).backward()
        og.step()
        log.append((err_d.item(), err_adv.item(), err_cont.item(), float(d_real.mean()), float(d_out.mean())))
        sd.step()
        sg.step()
    return log, {k: v.detach() for k, v in g_state.items()}, {k: v.detach() for k, v in d_state.items()}


def test_train_loop_shaped_iterations_match_the_oracle():
    importlib.import_module(PKG).install(fused_adam=True)
    try:
        import model_generator, model_discriminator, model_content_extractor          # the reference's module names
        utils = sys.modules['utils'] if hasattr(sys.modules.get('utils'), 'lr_from_hr') else pkg('utils')
        rp = pkg('replay')
        assert torch.optim.Adam is pkg('optim').Adam                                  # config.py:293-294 now builds the fused step
        torch.manual_seed(0)
        net_g = model_generator.Generator(2, 16, 64, [2], use_sn=True)               # config.py:79-80 (narrow)
        net_d = model_discriminator.Discriminator((3, HR, HR), FEATS, STRIDES)      # config.py:81-82
        ext = model_content_extractor.MaskedVGG(MASK, width_div=4, pretrained=False)
        g0 = {k: v.detach().clone() for k, v in net_g.state_dict().items()}
        d0 = {k: v.detach().clone() for k, v in net_d.state_dict().items()}
        v0 = {k: v.detach().clone() for k, v in ext.state_dict().items()}
        random.seed(21)
        np.random.seed(22)
        want_log, want_g, want_d = _oracle_loop(g0, d0, v0)

        device = torch.device('cuda')
        net_g, net_d, ext = net_g.to(device).train(), net_d.to(device).train(), ext.to(device)
        criterion = torch.nn.BCELoss()                                               # config.py:107
        optimizerG = torch.optim.Adam(net_g.parameters(), lr=LR0, betas=(.9, 0.999))
        optimizerD = torch.optim.Adam(net_d.parameters(), lr=LR0, betas=(.9, 0.999))
        f = RATIO ** (1.0 / TOTAL)
        schedulerG = torch.optim.lr_scheduler.LambdaLR(optimizerG, lr_lambda=lambda it: f ** it)
        schedulerD = torch.optim.lr_scheduler.LambdaLR(optimizerD, lr_lambda=lambda it: f ** it)
        real_label = torch.full((BS,), 1.0, device=device)                           # config.py:184-189
        real_label_reduced = torch.full((BS,), .9, device=device)
        fake_label = torch.full((BS,), .0, device=device)
        dis_list_old = rp.gen_dis_list({}, REPLAY_LEN, device)                      # config.py:323-331 (device-resident list)
        random.seed(21)
        np.random.seed(22)
        G_losses, D_losses, cont_losses, extra = [], [], [], []
        for i, img_hr in enumerate(_batches()):
            img_hr = img_hr.to(device)                                               # train.py:45
            img_lr = utils.lr_from_hr(img_hr, (LR, LR), device=device)              # train.py:46
            fake = net_g(img_lr)                                                     # train.py:53
            net_d.zero_grad()                                                        # train.py:58
            curr_fake = fake.detach()
            D_G_z1, D_x, errD = rp.adversarial_loss_d(net_d, criterion, img_hr, curr_fake, dis_list_old,
                                                      real_label_reduced, fake_label, REPLAY_RATIO)      # train.py:64
            dis_list_old.store(curr_fake, i, 1)                                      # train.py:66-71
            errD = errD * 1.0
            errD.backward()                                                          # train.py:74
            optimizerD.step()
            net_g.zero_grad()                                                        # train.py:82
            d_out = net_d(fake).view(-1)
            errG_adv = criterion(d_out, real_label) * 5e-2                           # train.py:171-181, config.py:136-140
            errG_cont = torch.mean(torch.pow(ext(img_hr) - ext(fake), 2)) * 1.0      # train.py:183-186
            errG = errG_adv + errG_cont
            errG.backward()
            optimizerG.step()                                                        # train.py:108
            G_losses.append(errG_adv.item())                                         # train.py:117-119: three host reads
            D_losses.append(errD.item())
            cont_losses.append(errG_cont.item())
            extra.append((float(D_x), float(d_out.mean())))
            schedulerD.step()                                                        # train.py:121-122
            schedulerG.step()
        assert len(dis_list_old) == REPLAY_LEN
        for i, w in enumerate(want_log):
            for got, ref, what in ((D_losses[i], w[0], 'errD'), (G_losses[i], w[1], 'errG_adv'), (cont_losses[i], w[2], 'errG_cont'),
                                   (extra[i][0], w[3], 'D(x)'), (extra[i][1], w[4], 'D(G(z))')):
                # (Adam divides by sqrt(v): rounding differences in near-zero gradient entries become parameter differences of
                # the size of the step, so the two trajectories drift apart slowly -- 2e-3 while fresh, 1e-2 later)
                assert abs(got - ref) <= (2e-3 if i < 2 else 1e-2) * max(abs(ref), 1e-2), (i, what, got, ref)
        assert abs(optimizerG.param_groups[0]['lr'] - LR0 * f ** ITERS) < 1e-12
        # parameters after ITERS Adam steps: the UPDATE (p - p0) is compared, relative to its own size
        for name, net, p0, want in (('G', net_g, g0, want_g), ('D', net_d, d0, want_d)):
            sd = net.state_dict()
            keys = {k: p for k, p in net.named_parameters()}
            for k, p in keys.items():
                if analytically_zero(k, keys):        # bias in front of a BatchNorm: its gradient is rounding noise, and Adam
                    continue                          # turns noise into full-size steps of random sign on both sides
                upd_ref = (want[k] - p0[k]).double()
                upd_got = (sd[k].cpu() - p0[k]).double()
                scale = float(upd_ref.abs().max())
                diff = upd_got - upd_ref
                # (Adam's step is lr * m / sqrt(v): entries whose gradient is rounding-sized move by a full step of either
                # sign on both sides, so single entries of the big tensors differ by a fraction of a step -- the bound is on
                # the RMS of the update difference, with a loose cap on the worst entry)
                assert scale > 0 and float(diff.pow(2).mean().sqrt()) <= 3e-2 * float(upd_ref.pow(2).mean().sqrt()), (name, k)
                assert float(diff.abs().max()) <= 0.25 * scale, (name, k)
            for k in sd:
                if k.endswith(('weight_u', 'running_mean', 'running_var')):
                    assert rel_err(sd[k].cpu(), want[k]) < 2e-3, (name, k)
    finally:
        for m in ('model_generator', 'model_generator_progressive', 'model_discriminator', 'model_content_extractor', 'utils'):
            sys.modules.pop(m, None)
        torch.optim.Adam = pkg('optim').Adam.__mro__[1]                              # undo install(fused_adam=True)
