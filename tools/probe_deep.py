"""launch times of the conv_deep.hip layers at the shapes of BASELINE's configs (B16): forward (ACT prologue, statistics) and data
gradient (BNACT_BWD prologue, fused reductions), HIP events on the launch stream.  Environment knobs are read per process:
SISR_DEEP=0 (generic kernel), SISR_DEEP_BN=64, SISR_DEEP_TARGET=<workgroups>, SISR_DEEP_MINCPS=<chunks>.
SISR_WGRAD_DEEP=0 (generic weight-gradient kernel), SISR_WGRAD_DEEP_PB=<pixel blocks>.
usage: python tools/probe_deep.py [hr96|hr192|all]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
import torch
from gpu_helpers import FakeConv, pkg
E, L = pkg('engine'), pkg('_lib')
E.set_precision('bf16')

D = [(64, 64, 2), (64, 128, 1), (128, 128, 2), (128, 256, 1), (256, 256, 2), (256, 512, 1), (512, 512, 2)]
VGG = [(64, 128, 1, 2), (128, 128, 1, 2), (128, 256, 1, 4), (256, 256, 1, 4), (256, 512, 1, 8), (512, 512, 1, 8), (512, 512, 1, 16)]


def layers(hr):
    out, r = [], hr
    for cin, cout, st in D:
        out.append(('D', cin, cout, st, r, r))
        r //= st
    for cin, cout, st, div in VGG:
        out.append(('V', cin, cout, st, hr // div, hr // div))
    out.append(('G', 64, 64, 1, hr // 4, hr // 4))
    return out


def timeit(fn, iters=20, reps=5):
    """GPU time per launch: `iters` launches captured into one HIP graph (no host time between them), replayed `reps` times"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    keep = []
    with torch.cuda.graph(g):
        for _ in range(iters):
            keep.append(fn())
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e3


which = sys.argv[1] if len(sys.argv) > 1 else 'hr96'
for hr in ([96] if which == 'hr96' else [192] if which == 'hr192' else [96, 192]):
    print('--- B16, HR %d   (knobs: %s)' % (hr, {k: v for k, v in os.environ.items() if k.startswith('SISR_DEEP') or k.startswith('SISR_WGRAD')}))
    for net, cin, cout, st, h, w in layers(hr):
        n = 16
        wt = (torch.rand(cout, cin, 3, 3, device='cuda') - 0.5) * 0.05
        ref = FakeConv(wt, torch.zeros(cout, device='cuda'), E.ConvGeom(cin, cout, 3, st, 1, deep_dgrad=(net == 'V')))
        E.reserve_capture_tables(4 << 20)
        p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
        ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
        x = (torch.rand(n, h, w, cin, device='cuda') - 0.5).to(torch.bfloat16)
        dy = (torch.rand(n, ho, wo, cout, device='cuda') - 0.5).to(torch.bfloat16)
        c = (torch.rand(n, ho, wo, cout, device='cuda') - 0.5).to(torch.bfloat16)
        k4 = torch.rand(4, cin, device='cuda') + 0.5
        q = torch.rand(5, cout, device='cuda') + 0.5
        xb = (torch.rand(n, h, w, cin, device='cuda') - 0.5).to(torch.bfloat16)
        flops = 2.0 * n * ho * wo * cin * cout * 9
        t_f = t_d = float('nan')
        info = ''
        if p.kinds[0]:
            t_f = timeit(lambda: E.conv_forward(p, E.Operand.act(x, 0.01), bias=ref.bias, stats=True))
            dp = p.plans[0].deep
            if p.kinds[0] == 2:
                info = 'BN %d tiles %dx%d split %d' % (dp.BN, dp.tiles_q * dp.tiles_x, dp.n_ntiles, dp.split)
        op = E.Operand(dy, tuple(dy.shape), pro=L.PRO_BNACT_BWD, x2=c, pa=q[0], pb=q[1], pd=q[2], ps=q[3], pt=q[4], slope=0.01)
        fuse = E.can_fuse_bn_backward(p)
        t_d = timeit(lambda: E.conv_dgrad(p, op, bnb=(xb, k4, 0.01) if fuse else None))
        t_w = float('nan')
        if net != 'V':                                  # (the VGG extractor has no weight gradient)
            xop = E.Operand(x, tuple(x.shape), pro=L.PRO_AFFINE_ACT, pa=k4[0], pd=k4[1], slope=0.01)
            t_w = timeit(lambda: E.conv_wgrad(p, xop, op))        # kernel + its slab reduction
            wp = p.plans[2].deep
            info += '  wgrad: %s' % ('deep tiles %d pb %d wgs %d' % (wp.n_tiles, wp.n_pb, wp.n_pb * wp.n_cib * wp.n_cob) if wp.enabled else 'generic')
        print('%s %3d->%3d s%d %3dx%-3d  fwd %7.1f us %6.0f TF   dgrad %7.1f us %6.0f TF   wgrad %7.1f us %6.0f TF   %.1f GF  %s' %
              (net, cin, cout, st, h, w, t_f, flops / t_f / 1e6, t_d, flops / t_d / 1e6, t_w, flops / t_w / 1e6, flops / 1e9, info))
