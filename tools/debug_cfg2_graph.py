"""debug: where does the graphed cfg2 iteration go non-finite?  VARIANT=new|old  RELOAD=0|1"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
PKG = 'single-image-super-resolution_amd'
sub = lambda n: importlib.import_module(PKG + '.' + n)
sub('engine').set_precision(os.environ.get('SISR_PRECISION', 'bf16'))
mg, md, mce, ut, op, G = sub('model_generator'), sub('model_discriminator'), sub('model_content_extractor'), sub('utils'), sub('optim'), sub('graph')
dev = torch.device('cuda', 0)
B, HR, LR = 16, 96, 48
FEATS, STRIDES = [64, 64, 128, 128, 256, 256, 512, 512], [1, 2, 1, 2, 1, 2, 1, 2]
torch.manual_seed(0)
net_g = mg.Generator(16, 64, 256, [2], use_sn=True).to(dev).train()
net_d = md.Discriminator((3, HR, HR), FEATS, STRIDES).to(dev).train()
ext = mce.MaskedVGG(0b00010, pretrained=False).to(dev)
og, od = op.Adam(net_g.parameters(), lr=1e-5), op.Adam(net_d.parameters(), lr=1e-5)
hr = ((torch.rand(B, 3, HR, HR, generator=torch.Generator().manual_seed(51)) * 2 - 1)).to(dev)
ones, red, zeros = torch.ones(B, device=dev), torch.full((B,), .9, device=dev), torch.zeros(B, device=dev)


def bce(p, t):          # BCELoss without the device-side range assert
    p = p.clamp(1e-12, 1 - 1e-7)
    return -(t * torch.log(p) + (1 - t) * torch.log(1 - p)).mean()


def d_part():
    lr = ut.lr_from_hr(hr, (LR, LR), device=dev)
    fake = net_g(lr)
    net_d.zero_grad()
    dr, df = net_d(hr).view(-1), net_d(fake.detach()).view(-1)
    err_d = bce(dr, red) + bce(df, zeros)
    err_d.backward()
    return err_d, lr, fake.detach(), dr.detach(), df.detach()


def g_part():
    lr = ut.lr_from_hr(hr, (LR, LR), device=dev)
    fake = net_g(lr)
    net_g.zero_grad()
    dg = net_d(fake).view(-1)
    fr, ff = ext(hr), ext(fake)
    err_g = bce(dg, ones) * 5e-2 + torch.mean(torch.pow(fr - ff, 2))
    err_g.backward()
    return err_g, fake.detach(), dg.detach(), fr.detach(), ff.detach()


def fin(name, t):
    ok = bool(torch.isfinite(t).all())
    print('   %-8s finite=%s absmax=%.4g' % (name, ok, float(t.abs().max()) if ok else float('nan')), flush=True)
    return ok


class OldGraphed:
    def __init__(self, fn, warmup=2):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn()

    def __call__(self):
        self.graph.replay()
        return self.out


variant, reload_ = os.environ.get('VARIANT', 'new'), os.environ.get('RELOAD', '1') == '1'
print('=== variant', variant, 'reload', reload_, flush=True)
sg = {k: v.clone() for k, v in net_g.state_dict().items()}
sd = {k: v.clone() for k, v in net_d.state_dict().items()}
GS = G.GraphedStep if variant == 'new' else OldGraphed
dgr, ggr = GS(d_part), GS(g_part)
if reload_:
    net_g.load_state_dict(sg); net_d.load_state_dict(sd)
for it in range(3):
    out = dgr()
    torch.cuda.synchronize()
    print(' iter', it, 'D part', flush=True)
    for n, t in zip(('err_d', 'lr', 'fake', 'd_real', 'd_fake'), out):
        fin(n, t)
    for n, p in list(net_d.named_parameters())[:3] + list(net_d.named_parameters())[-4:]:
        fin('g:' + n[-18:], p.grad)
    od.step()
    torch.cuda.synchronize()
    fin('dparams', torch.cat([p.detach().reshape(-1) for p in net_d.parameters()]))
    out = ggr()
    torch.cuda.synchronize()
    print(' iter', it, 'G part', flush=True)
    for n, t in zip(('err_g', 'fake', 'd_g', 'f_real', 'f_fake'), out):
        fin(n, t)
    og.step()
    torch.cuda.synchronize()
    fin('gparams', torch.cat([p.detach().reshape(-1) for p in net_g.parameters()]))
print('done', flush=True)
