#!/usr/bin/env python3
"""Generate the golden fixtures in this directory.  Run ONLY in the authoring container:

    python tests/golden/make_golden.py

It imports the reference's own torch-only modules from /root/reference (model_generator,
model_discriminator, model_generator_progressive -- SURVEY.md 8c), loads a deterministic
synthetic state (oracle/init.py) into them, runs forward + backward on CPU (torch 2.10.0) and
stores inputs / expected outputs as small ``.npz`` files.  The reference does not travel to the
GPU box; these vectors do.  utils.py and model_content_extractor.py import torchvision (absent
here), so their fixtures come from the torch primitives the reference calls
(``F.interpolate(..., 'bicubic', align_corners=True)`` utils.py:17; a VGG19-``features``
stand-in built from nn.Conv2d / nn.ReLU(inplace=True) / nn.MaxPool2d with synthetic weights).
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

from oracle import init as oinit            # noqa: E402
from oracle import models as omodels        # noqa: E402

import model_generator as ref_g             # noqa: E402  (reference)
import model_discriminator as ref_d         # noqa: E402  (reference)
import model_generator_progressive as ref_p  # noqa: E402  (reference)


def load_synth(module, seed):
    sd = module.state_dict()
    state = oinit.synth_state({k: v.shape for k, v in sd.items()}, seed)
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(state[k])
    return state


def run_case(name, module, cfg, in_shape, seed):
    torch.manual_seed(0)
    state = load_synth(module, seed)
    x = oinit.synth_input(in_shape, seed).requires_grad_(True)
    module.train()
    out = module(x)
    r = oinit.synth_input(out.shape, seed + 1)
    (out * r).sum().backward()
    rec = {'cfg': np.array(json.dumps(cfg)), 'x': x.detach().numpy(), 'r': r.numpy(),
           'out': out.detach().numpy(), 'grad_x': x.grad.numpy()}
    for k, v in state.items():
        rec['state/' + k] = v.numpy()
    for k, p in module.named_parameters():
        rec['grad/' + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    after = module.state_dict()
    for k in after:
        if k.endswith(('weight_u', 'weight_v', 'running_mean', 'running_var', 'num_batches_tracked')):
            rec['after/' + k] = after[k].numpy().copy()
    with torch.no_grad():
        rec['out2'] = module(x).numpy()                  # second training-mode forward
        module.eval()
        rec['out_eval'] = module(x).numpy()              # eval forward on the advanced state
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **rec)
    print('%-28s out%s  %.2f MB' % (name, tuple(out.shape), os.path.getsize(path) / 1e6))


def run_sampled_case(name, module, cfg, in_shape, seed):
    """Full-depth networks: the state is NOT stored (it is regenerated from oracle/init.py's synth_state with
    `state_seed`); big gradients are stored as a fixed strided sample of 4096 entries + their sum."""
    torch.manual_seed(0)
    state = load_synth(module, seed)
    x = oinit.synth_input(in_shape, seed).requires_grad_(True)
    module.train()
    out = module(x)
    r = oinit.synth_input(out.shape, seed + 1)
    (out * r).sum().backward()
    rec = {'cfg': np.array(json.dumps(dict(cfg, state_seed=seed))), 'x': x.detach().numpy(), 'r': r.numpy(),
           'out': out.detach().numpy(), 'grad_x': x.grad.numpy()}
    for k, p in module.named_parameters():
        if p.numel() <= 4096:
            rec['grad/' + k] = p.grad.numpy()
        else:
            flat = p.grad.reshape(-1)
            rec['gradsample/' + k] = flat[:: max(1, flat.numel() // 4096)][:4096].numpy()
            rec['gradsum/' + k] = np.array(flat.double().sum().item())
    after = module.state_dict()
    for k in after:
        if k.endswith(('weight_u', 'weight_v', 'running_mean', 'running_var', 'num_batches_tracked')):
            rec['after/' + k] = after[k].numpy().copy()
    with torch.no_grad():
        rec['out2'] = module(x).numpy()
    del state
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **rec)
    print('%-28s out%s  %.2f MB' % (name, tuple(out.shape), os.path.getsize(path) / 1e6))


def deep_generator_cases():
    """the benchmark's own architecture at full depth (16 residual blocks = 34 stacked conv+BatchNorm layers,
    spectral norm everywhere; config.py:79-80) on a small batch"""
    cfg = dict(kind='generator', n_blocks=16, nf=64, nl=256, list_scales=[2], use_sn=True, n_suffix=0)
    run_sampled_case('gen_x2_sn_16blocks', ref_g.Generator(16, 64, 256, [2], use_sn=True), cfg, (2, 3, 16, 16), 14)


def generator_cases():
    cfg = dict(kind='generator', n_blocks=1, nf=64, nl=256, list_scales=[2], use_sn=True, n_suffix=0)
    run_case('gen_x2_sn_w64', ref_g.Generator(1, 64, 256, [2], use_sn=True), cfg, (2, 3, 12, 12), 1)

    cfg = dict(kind='generator', n_blocks=2, nf=16, nl=64, list_scales=[2], use_sn=False, n_suffix=0)
    run_case('gen_x2_nosn_w16', ref_g.Generator(2, 16, 64, [2], use_sn=False), cfg, (2, 3, 10, 14), 2)

    cfg = dict(kind='generator', n_blocks=1, nf=32, nl=128, list_scales=[2, 2], use_sn=True, n_suffix=0)
    run_case('gen_x4_scales22_w32', ref_g.Generator(1, 32, 128, [2, 2], use_sn=True), cfg, (1, 3, 9, 8), 3)

    cfg = dict(kind='generator', n_blocks=1, nf=32, nl=128, list_scales=[2], use_sn=True, n_suffix=1)
    g = ref_g.GeneratorSuffix(ref_g.Generator(1, 32, 128, [2], use_sn=True))
    run_case('gen_x4_suffix_w32', g, cfg, (2, 3, 8, 8), 4)

    cfg = dict(kind='generator', n_blocks=1, nf=16, nl=64, list_scales=[2], use_sn=False, n_suffix=2)
    g = ref_g.GeneratorSuffix(ref_g.GeneratorSuffix(ref_g.Generator(1, 16, 64, [2], use_sn=False)))
    run_case('gen_x8_suffix2_w16', g, cfg, (1, 3, 6, 6), 5)


def progressive_cases():
    g0 = ref_p.GeneratorProgresiveBase(1, n_features=16)
    g1 = ref_p.GeneratorSuffix(g0, n_features=16)
    cfg = dict(kind='progressive', n_blocks=1, nf=16, n_suffix=1)
    run_case('prog_x2_w16', g1, cfg, (2, 3, 8, 8), 6)
    g0 = ref_p.GeneratorProgresiveBase(1, n_features=64)
    g1 = ref_p.GeneratorSuffix(g0, n_features=64)
    g2 = ref_p.GeneratorSuffix(g1.beginning, n_features=16)
    g3 = ref_p.GeneratorSuffix(g2.beginning, n_features=4)
    cfg = dict(kind='progressive', n_blocks=1, nf=64, n_suffix=3)
    run_case('prog_x8_w64', g3, cfg, (1, 3, 8, 8), 7)


def discriminator_cases():
    feats, strides = [16, 16, 32, 32], [1, 2, 1, 2]
    cfg = dict(kind='discriminator', input_shape=[3, 16, 16], list_n_features=feats, list_stride=strides)
    run_case('dis_16px_w16', ref_d.Discriminator((3, 16, 16), feats, strides), cfg, (4, 3, 16, 16), 8)
    feats, strides = [64, 64, 128, 128, 256, 256, 512, 512], [1, 2, 1, 2, 1, 2, 1, 2]
    cfg = dict(kind='discriminator', input_shape=[3, 32, 32], list_n_features=feats, list_stride=strides)
    # full SRGAN feature/stride lists on 32x32 inputs (B=4 so the last BatchNorm still sees 16
    # samples per channel -- smaller is ill-conditioned): fc_in = 2048; state is large (6.8 M
    # params), so only outputs / input-grad and a few small grads are stored (see below)
    torch.manual_seed(0)
    m = ref_d.Discriminator((3, 32, 32), feats, strides)
    state = load_synth(m, 9)
    x = oinit.synth_input((4, 3, 32, 32), 9).requires_grad_(True)
    out = m(x)
    r = oinit.synth_input(out.shape, 10)
    (out * r).sum().backward()
    rec = {'cfg': np.array(json.dumps(dict(cfg, state_seed=9))), 'x': x.detach().numpy(), 'r': r.numpy(),
           'out': out.detach().numpy(), 'grad_x': x.grad.numpy()}
    for k, p in m.named_parameters():
        if p.numel() <= 4096:
            rec['grad/' + k] = p.grad.numpy()
        else:     # big tensors: a fixed strided sample of 4096 entries + the sum
            flat = p.grad.reshape(-1)
            rec['gradsample/' + k] = flat[:: max(1, flat.numel() // 4096)][:4096].numpy()
            rec['gradsum/' + k] = np.array(flat.double().sum().item())
    del state
    path = os.path.join(HERE, 'dis_32px_srgan.npz')
    np.savez_compressed(path, **rec)
    print('%-28s out%s  %.2f MB' % ('dis_32px_srgan', tuple(out.shape), os.path.getsize(path) / 1e6))


def bicubic_cases():
    rec = {}
    for i, (shape, size) in enumerate([((2, 3, 16, 16), (8, 8)), ((1, 3, 13, 10), (5, 7)),
                                       ((2, 1, 8, 8), (4, 4)), ((1, 3, 24, 24), (6, 6)),
                                       ((1, 2, 6, 6), (12, 9))]):
        x = oinit.synth_input(shape, 20 + i) * 1.2          # overshoot so the clamp acts
        x.requires_grad_(True)
        y = F.interpolate(x, size, mode='bicubic', align_corners=True)            # utils.py:17
        lr = torch.max(torch.min(y, torch.full((1,), 1.0)), torch.full((1,), -1.0))  # utils.py:20
        r = oinit.synth_input(lr.shape, 40 + i)
        (y * r).sum().backward()
        rec['x%d' % i], rec['size%d' % i] = x.detach().numpy(), np.array(size)
        rec['interp%d' % i], rec['lr%d' % i] = y.detach().numpy(), lr.detach().numpy()
        rec['r%d' % i], rec['grad_x%d' % i] = r.numpy(), x.grad.numpy()
    rec['n'] = np.array(5)
    np.savez_compressed(os.path.join(HERE, 'bicubic.npz'), **rec)
    print('bicubic ok')


def patch_pipeline_cases():
    """row f4: the dataset transform of config.py:225-231 applied with the libraries the reference calls -- torchvision's
    Resize on a PIL image is ``img.resize(size[::-1], PIL.Image.BILINEAR)``, ToTensor is uint8 HWC -> float CHW / 255,
    Normalize(.5, .5) is (t - .5) / .5 (torchvision itself is not installed here; Pillow, which does the arithmetic, is)
    -- followed by the F.interpolate call of utils.py:17 and the clamp of utils.py:20.  Sizes: CelebA (218 x 178) to the
    cfg1 / cfg2 crops, a Flickr-like landscape image to HR 192, MNIST-like grayscale 28 -> 14, an up-scaling case."""
    from PIL import Image
    rs = np.random.RandomState(7)
    rec, cases = {}, [((2, 218, 178, 3), (64, 64), (32, 32)), ((2, 218, 178, 3), (96, 96), (48, 48)),
                      ((1, 333, 500, 3), (192, 192), (48, 48)), ((2, 28, 28, 1), (14, 14), (7, 7)),
                      ((1, 37, 53, 3), (64, 64), (32, 32)), ((1, 64, 100, 3), (64, 64), (32, 32))]
    for i, (shape, hr, lr) in enumerate(cases):
        # smooth random images (pure noise would make every anti-aliased pixel ~ 128)
        base = rs.randint(0, 256, (shape[0], shape[1] // 8 + 2, shape[2] // 8 + 2, shape[3])).astype(np.uint8)
        imgs = np.stack([np.asarray(Image.fromarray(b.squeeze(-1) if shape[3] == 1 else b).resize((shape[2], shape[1]), Image.BICUBIC))
                         .reshape(shape[1:]) for b in base])
        imgs = np.clip(imgs.astype(int) + rs.randint(-20, 21, imgs.shape), 0, 255).astype(np.uint8)
        outs = []
        for im in imgs:
            pil = Image.fromarray(im.squeeze(-1) if shape[3] == 1 else im)
            res = np.asarray(pil.resize((hr[1], hr[0]), Image.BILINEAR)).reshape(hr + (shape[3],))     # transforms.Resize
            t = torch.from_numpy(res.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)       # ToTensor
            t = (t - 0.5) / 0.5                                                                          # Normalize
            outs.append(t)
        img_hr = torch.stack(outs)
        y = F.interpolate(img_hr, lr, mode='bicubic', align_corners=True)                                # utils.py:17
        img_lr = torch.max(torch.min(y, torch.full((1,), 1.0)), torch.full((1,), -1.0))                  # utils.py:20
        rec['imgs%d' % i], rec['hr_size%d' % i], rec['lr_size%d' % i] = imgs, np.array(hr), np.array(lr)
        rec['img_hr%d' % i], rec['img_lr%d' % i] = img_hr.numpy(), img_lr.numpy()
    rec['n'] = np.array(len(cases))
    path = os.path.join(HERE, 'patch_pipeline.npz')
    np.savez_compressed(path, **rec)
    print('patch pipeline ok  %.2f MB' % (os.path.getsize(path) / 1e6))


def vgg_cases():
    """VGG19-features stand-in (torch primitives) + the reference's tap loop semantics; weights
    synthetic (pretrained fetch impossible offline).  Channel widths divided by 8 to
    keep fixtures small; the topology (indices, in-place ReLUs, pools) is the real one."""
    layers, cin = [], 3
    for v in omodels.VGG19_CFG:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v // 8, 3, padding=1), nn.ReLU(inplace=True)]
            cin = v // 8
    feats = nn.Sequential(*layers)
    assert tuple(i for i, l in enumerate(feats) if isinstance(l, nn.MaxPool2d)) == omodels.MAXPOOL_INDEXES
    sd = feats.state_dict()
    state = oinit.synth_state({'layers.' + k: v.shape for k, v in sd.items()}, 11)
    with torch.no_grad():
        for k, v in sd.items():
            # gain sqrt(2) on weights keeps ReLU stacks from vanishing
            v.copy_(state['layers.' + k] * (2 ** 0.5 if k.endswith('weight') else 1.0))
            state['layers.' + k] = v.clone()
    rec = {'width_div': np.array(8)}
    for k, v in state.items():
        rec['state/' + k] = v.numpy()
    x0 = oinit.synth_input((2, 3, 32, 32), 12)
    rec['x'] = x0.numpy()
    for mask in (0b00010, 0b00011, 0b01111, 0b10000, 0b10101, 0b11111):
        kept = [omodels.MAXPOOL_INDEXES_BEFORE_ACT[i] for i in range(5) if mask & (1 << i)]
        sub = feats[:kept[-1]]
        x = x0.clone().requires_grad_(True)
        h, saved = x, []
        for i, l in enumerate(sub, 1):
            h = l(h)
            if i in kept:
                saved.append(h)
        out = torch.cat([e.view(e.shape[0], -1) for e in saved], dim=1)
        r = oinit.synth_input(out.shape, 13 + mask)
        (out * r).sum().backward()
        rec['out_%d' % mask], rec['r_%d' % mask], rec['grad_x_%d' % mask] = \
            out.detach().numpy(), r.numpy(), x.grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'vgg_standin.npz'), **rec)
    print('vgg stand-in ok  %.2f MB' % (os.path.getsize(os.path.join(HERE, 'vgg_standin.npz')) / 1e6))


if __name__ == '__main__':
    torch.set_num_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == 'patches':   # only the patch-pipeline fixture (needs Pillow, not the reference)
        patch_pipeline_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'deep':      # only the full-depth fixtures
        deep_generator_cases()
        sys.exit(0)
    generator_cases()
    deep_generator_cases()
    progressive_cases()
    discriminator_cases()
    bicubic_cases()
    patch_pipeline_cases()
    vgg_cases()
