"""Timing of one FULL SRGAN training iteration (HR 96, x2, B16; D step + G step with the VGG22
content loss, train.py:45-108) on the drop-in modules -- a sanity measurement of rows a5/a6/a7 next to the headline
generator-only benchmark of bench.py (developer tool; not the driver's metric)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
PKG = 'single-image-super-resolution_amd'
sub = lambda n: importlib.import_module(PKG + '.' + n)
prec = os.environ.get('SISR_PRECISION', 'bf16')
sub('engine').set_precision(prec)
mg, md, mce, ut, op = sub('model_generator'), sub('model_discriminator'), sub('model_content_extractor'), sub('utils'), sub('optim')
dev = torch.device('cuda', 0)
B, HR, LR = 16, int(os.environ.get('HR', '96')), int(os.environ.get('HR', '96')) // 2
torch.manual_seed(0)
net_g = mg.Generator(16, 64, 256, [2], use_sn=True).to(dev).train()
net_d = md.Discriminator((3, HR, HR), [64, 64, 128, 128, 256, 256, 512, 512], [1, 2, 1, 2, 1, 2, 1, 2]).to(dev).train()
ext = mce.MaskedVGG(0b00010, pretrained=False).to(dev)
og, od = op.Adam(net_g.parameters(), lr=1e-5), op.Adam(net_d.parameters(), lr=1e-5)
crit = torch.nn.BCELoss()
hr = (torch.rand(B, 3, HR, HR, device=dev) * 2 - 1)
ones, red, zeros = torch.ones(B, device=dev), torch.full((B,), .9, device=dev), torch.zeros(B, device=dev)


def d_part():                       # train.py:45-74: G forward, D on real and on detached fake, backward
    lr = ut.lr_from_hr(hr, (LR, LR), device=dev)
    fake = net_g(lr)
    net_d.zero_grad()
    err_d = crit(net_d(hr).view(-1), red) + crit(net_d(fake.detach()).view(-1), zeros)
    err_d.backward()
    return err_d


def g_part():                       # train.py:82-107: D on fake with the UPDATED D, VGG content loss, backward
    lr = ut.lr_from_hr(hr, (LR, LR), device=dev)
    fake = net_g(lr)
    net_g.zero_grad()
    err_g = crit(net_d(fake).view(-1), ones) * 5e-2 + torch.mean(torch.pow(ext(hr) - ext(fake), 2))
    err_g.backward()
    return err_g


GRAPH = os.environ.get('GRAPH', '0') == '1'       # GRAPH=1: the two halves replayed from HIP graphs
if GRAPH:
    # the G step re-runs the generator forward (its graph must own the autograd state it differentiates), which the
    # eager sequence shares with the D step: the graphed iteration does ~0.4 ms more GPU work and no host work
    d_graph, g_graph = sub('graph').GraphedStep(d_part), sub('graph').GraphedStep(g_part)


def iteration():
    if GRAPH:
        err_d = d_graph()
        od.step()
        err_g = g_graph()
        og.step()
        return err_d, err_g
    lr = ut.lr_from_hr(hr, (LR, LR), device=dev)
    fake = net_g(lr)
    net_d.zero_grad()
    err_d = crit(net_d(hr).view(-1), red) + crit(net_d(fake.detach()).view(-1), zeros)
    err_d.backward()
    od.step()
    net_g.zero_grad()
    err_g = crit(net_d(fake).view(-1), ones) * 5e-2 + torch.mean(torch.pow(ext(hr) - ext(fake), 2))
    err_g.backward()
    og.step()
    return err_d, err_g


for _ in range(3):
    iteration()
torch.cuda.synchronize()
n = int(os.environ.get('ITERS', '10'))
t0 = time.perf_counter()
for _ in range(n):
    iteration()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('full SRGAN iteration (%s, %s mode, B%d HR%d): %.2f ms  = %.0f HR patches/s' % ('HIP-graph replay' if GRAPH else 'eager launches', prec, B, HR, dt * 1e3, B / dt))
