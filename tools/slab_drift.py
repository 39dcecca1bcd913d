"""Does rounding the weight-gradient slabs to bf16 (wgrad_trunk.hip, wgrad_deep.hip; SISR_SLAB_BF16) make a training trajectory drift?

  run:      python tools/slab_drift.py run <name> <precision> <iters>     (one process per variant: the knob is read when plans are made)
            -- BASELINE's cfg2 iteration exactly as bench.py builds it (G + D + VGG22, both fused Adam steps, B16, fixed seeds),
               launched eagerly for <iters> iterations; writes the parameters before and after to $SISR_DRIFT_DIR/drift_<name>.pt (default /tmp/sisr_drift)
  compare:  python tools/slab_drift.py compare ref a b ...
            -- for every variant: |theta - theta_ref| / |theta_ref - theta_0| per network (distance from the reference trajectory
               relative to the distance travelled), and the same between the variants themselves.
The question is answered by the DIFFERENCE between `bf16 slabs` and `fp32 slabs` against the same fp32-build reference: both carry
the bf16 build's operand rounding; only one carries the slab rounding.  (tools/r4_slab_drift.sh runs the three variants.)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.environ.get('SISR_DRIFT_DIR', '/tmp/sisr_drift')          # (the parameter dumps are 2 x 94 MB per variant: not under gpurun_out/)
os.makedirs(OUT, exist_ok=True)


def flat(net):
    import torch
    return torch.cat([p.detach().float().reshape(-1).cpu() for p in net.parameters()])


def run(name, precision, iters):
    import torch
    import bench
    E = bench.sub('engine')
    orig = E.set_precision
    E.set_precision = lambda p: orig(precision)                 # make_config_iteration asks for bf16: build what was asked for HERE
    dev = torch.device('cuda:0')
    it, info, keep = bench.make_config_iteration('cfg2', dev, 0, 1, False, print)
    net_g, net_d, og, od = keep[0], keep[1], keep[3], keep[4]
    for o in (og, od):
        for grp in o.param_groups:
            grp['lr'] = 1e-4                                       # (bench's 1e-5 moves the weights too little to see anything in 50 steps)
    before = {'g': flat(net_g), 'd': flat(net_d)}
    losses = []
    for i in range(iters):
        err_d, err_g = it()
        if i % 10 == 9:
            losses.append((float(err_d), float(err_g)))
    torch.cuda.synchronize()
    torch.save({'before': before, 'after': {'g': flat(net_g), 'd': flat(net_d)}, 'losses': losses, 'precision': precision,
                'slab_bf16': os.environ.get('SISR_SLAB_BF16', '1'), 'iters': iters}, os.path.join(OUT, 'drift_%s.pt' % name))
    print(name, precision, 'slab_bf16 =', os.environ.get('SISR_SLAB_BF16', '1'), 'losses', losses)


def compare(names):
    import torch
    recs = {n: torch.load(os.path.join(OUT, 'drift_%s.pt' % n), weights_only=True) for n in names}
    ref = recs[names[0]]
    print('trajectories of %d cfg2 iterations (Adam lr 1e-4, B16, eager); reference = %s (%s build)' % (ref['iters'], names[0], ref['precision']))
    for net in ('g', 'd'):
        travelled = (ref['after'][net] - ref['before'][net]).double().norm()
        print('network %s: |theta_ref - theta_0| = %.4e  (|theta_0| = %.4e)' % (net.upper(), travelled, ref['before'][net].double().norm()))
        for n in names[1:]:
            r = recs[n]
            assert torch.equal(r['before'][net], ref['before'][net]), 'variants must start from the same parameters'
            d = (r['after'][net] - ref['after'][net]).double().norm()
            print('   %-14s (%s build, SISR_SLAB_BF16=%s): |theta - theta_ref| / travelled = %.4f' % (n, r['precision'], r['slab_bf16'], d / travelled))
        for i in range(1, len(names)):
            for j in range(i + 1, len(names)):
                d = (recs[names[i]]['after'][net] - recs[names[j]]['after'][net]).double().norm()
                print('   %s vs %s: %.4f of the distance travelled' % (names[i], names[j], d / travelled))
    for n in names:
        print('losses every 10 iterations (errD, errG)', n, [(round(a, 4), round(b, 4)) for a, b in recs[n]['losses']])


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    else:
        compare(sys.argv[2:])
