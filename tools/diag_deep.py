"""conv_deep: dump the first accumulator registers of tile 0 / wave 0 and what the store stage sees, statistics on and off"""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
os.environ['SISR_LIB'] = os.path.join(ROOT, 'single-image-super-resolution_amd', 'csrc', 'libsisr_hip_dump.so')
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from gpu_helpers import FakeConv, nchw, nhwc, pkg
E, L = pkg('engine'), pkg('_lib')
E.set_precision('bf16')


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


bf = lambda t: t.to(torch.bfloat16).float()
n, cin, cout, h, w = 1, 64, 64, 5, 24
wt = rnd((cout, cin, 3, 3), 2, 0.07)
ref = FakeConv(wt.cuda(), torch.zeros(cout).cuda(), E.ConvGeom(cin, cout, 3, 1, 1))
p = E.prepare_weights([(ref, n, h, w)], training=True)[0][0]
x = bf(rnd((n, cin, h, w), 31))
xd = nhwc(x).cuda().to(torch.bfloat16)
r = F.conv2d(x.double(), bf(wt).double(), padding=1)
print('ref pixel (0,0) ch 0..5', ['%.4f' % v for v in r[0, :6, 0, 0].tolist()], ' pixel (0,1) ch0 %.4f pixel (0,4) ch0 %.4f' % (float(r[0, 0, 0, 1]), float(r[0, 0, 0, 4])))
for stats in (False, True, False):
    E.DEEP_DEBUG = torch.zeros(1024, device='cuda')
    y, _, _ = E.conv_forward(p, E.Operand.plain(xd), bias=None, stats=stats)
    torch.cuda.synchronize()
    dbg = E.DEEP_DEBUG.cpu()
    got = nchw(y.float()).cpu().double()
    print('stats', stats, 'out pixel (0,0) ch 0..5', ['%.4f' % v for v in got[0, :6, 0, 0].tolist()])
    print('   acc[0][0][0] lanes 0..5 (row 0, ch 0..5):', ['%.4f' % v for v in dbg[0:6].tolist()], ' lanes 32..34 (row 4):', ['%.4f' % v for v in dbg[32:35].tolist()])
    print('   acc[0][0][1] lanes 0..2 (row 1):', ['%.4f' % v for v in dbg[64:67].tolist()])
    print('   store stage ps=0: ro lanes 0..3', dbg[256:260].tolist(), 'vo', dbg[320:324].tolist(), 'lo[0]', ['%.4f' % v for v in dbg[384:388].tolist()],
          'lo[1]', ['%.4f' % v for v in dbg[448:452].tolist()], 'acc000 now', ['%.4f' % v for v in dbg[512:516].tolist()])
