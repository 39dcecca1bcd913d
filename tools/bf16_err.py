import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
from gpu_helpers import pkg
from helpers import load_case
from test_gpu_generator import build
E = pkg('engine')
def l2(a, b): a, b = a.double().reshape(-1), torch.as_tensor(b).double().reshape(-1); return float((a-b).norm()/b.norm())
def mx(a, b): a, b = a.double().reshape(-1), torch.as_tensor(b).double().reshape(-1); return float((a-b).abs().max()/b.abs().max())
for name in ['gen_x2_sn_w64', 'gen_x4_suffix_w32', 'prog_x8_w64', 'gen_x2_nosn_w16']:
    z, cfg, state, grads, after = load_case(name)
    for prec in ('fp32', 'bf16'):
        E.set_precision(prec)
        net = build(cfg); net.load_state_dict(state, strict=True); net = net.cuda().train()
        x = torch.from_numpy(z['x']).cuda().requires_grad_(True)
        out = net(x); (out * torch.from_numpy(z['r']).cuda()).sum().backward()
        pg = {k: p.grad.cpu() for k, p in net.named_parameters()}
        big = max(float(v.abs().max()) for v in grads.values())
        worst = max((float((pg[k]-grads[k]).abs().max())/max(float(grads[k].abs().max()), 0.05*big), k) for k in grads)
        wl2 = max((l2(pg[k], grads[k]) if float(grads[k].abs().max()) > 0.05*big else 0.0, k) for k in grads)
        print('%-20s %s out max %.2e l2 %.2e | gx max %.2e l2 %.2e | pgrad worst max %.2e (%s) l2 %.2e (%s)' % (name, prec, mx(out.detach().cpu(), z['out']), l2(out.detach().cpu(), z['out']), mx(x.grad.cpu(), z['grad_x']), l2(x.grad.cpu(), z['grad_x']), worst[0], worst[1][-30:], wl2[0], wl2[1][-30:]))

# full-depth generator: bf16 mode against the fp32 (parity) mode on the same seeded state / batch
mg = pkg('model_generator')
for (B, hw) in ((4, 48), (16, 96)):
    res = {}
    for prec in ('fp32', 'bf16'):
        E.set_precision(prec)
        torch.manual_seed(0)
        net = mg.Generator(16, 64, 256, [2], use_sn=True).cuda().train()
        x = (torch.rand(B, 3, hw, hw, generator=torch.Generator().manual_seed(1)) * 2 - 1).cuda().requires_grad_(True)
        tgt = (torch.rand(B, 3, 2 * hw, 2 * hw, generator=torch.Generator().manual_seed(2)) * 2 - 1).cuda()
        out = net(x); loss = 10 * torch.mean((out - tgt) ** 2); loss.backward()
        res[prec] = (out.detach().cpu(), x.grad.cpu(), {k: p.grad.cpu() for k, p in net.named_parameters()}, float(loss))
    a, b = res['bf16'], res['fp32']
    l2s = sorted(((l2(a[2][k], b[2][k]), k) for k in b[2] if float(b[2][k].abs().max()) > 1e-6), reverse=True)
    print('G16 B%d %dx%d: loss %.6f vs %.6f | out l2 %.2e | gx l2 %.2e | param-grad l2: worst %.2e (%s) median %.2e' % (
        B, hw, hw, a[3], b[3], l2(a[0], b[0]), l2(a[1], b[1]), l2s[0][0], l2s[0][1], l2s[len(l2s)//2][0]))
