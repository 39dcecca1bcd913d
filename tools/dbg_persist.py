import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
from gpu_helpers import pkg, FakeConv, nhwc, nchw
E = pkg('engine'); L = pkg('_lib')
E.set_precision('bf16')
torch.manual_seed(0)
n, h, w = 2, 12, 12
def run(persist, cout=64, shuffle=False, pro='affine'):
    if persist: os.environ.pop('SISR_BF16_NO_PERSIST', None)
    else: os.environ['SISR_BF16_NO_PERSIST'] = '1'
    g = torch.Generator().manual_seed(1)
    wt = (torch.rand(cout, 64, 3, 3, generator=g) - 0.5) * 0.2
    b = (torch.rand(cout, generator=g) - 0.5) * 0.2
    x = (torch.rand(n, h, w, 64, generator=g) * 2 - 1).cuda()
    sc = (torch.rand(64, generator=g) + 0.5).cuda(); sh = (torch.rand(64, generator=g) - 0.5).cuda()
    slope = torch.full((1,), 0.25).cuda()
    geom = E.ConvGeom(64, cout, 3, 1, 1, shuffle2=shuffle)
    ref = FakeConv(wt.cuda(), b.cuda(), geom)
    preps, keep = E.prepare_weights([(ref, n, h, w)], training=True)
    p = preps[0]
    op = {'affine': E.Operand.affine_act(x, sc, sh, slope), 'act': E.Operand.act(x, slope), 'none': E.Operand.plain(x)}[pro]
    y, sp, cp = E.conv_forward(p, op, bias=ref.bias, stats=not shuffle)
    res = {'variant': p.plans[0].plan.variant, 'y': y.cpu()}
    if not shuffle:
        bn = torch.nn.BatchNorm2d(cout).cuda()
        k = E.bn_finalize(sp, cp, bn)
        res['k'] = k.cpu(); res['cnt'] = cp.cpu()
    return res
for cfg in [dict(pro='none'), dict(pro='act'), dict(pro='affine'), dict(pro='none', cout=256, shuffle=True)]:
    a, b = run(True, **cfg), run(False, **cfg)
    print(cfg, 'variants', a['variant'], b['variant'], 'y maxdiff %.3e (max %.2f)' % (float((a['y'] - b['y']).abs().max()), float(b['y'].abs().max())),
          ('k maxdiff %.3e cnt %s vs %s' % (float((a['k'] - b['k']).abs().max()), a['cnt'].sum().item(), b['cnt'].sum().item())) if 'k' in a else '')
