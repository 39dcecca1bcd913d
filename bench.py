#!/usr/bin/env python3
"""Benchmark of the SRGAN hot path on MI355X (driver contract: see the task statement).

Workload (BASELINE.json north_star headline): the x2 generator
``Generator(16, 64, 256, [2], use_sn=True)`` trained on synthetic U(-1,1) HR patches,
per GPU B=16 patches of 192x192 (LR 96x96).  One step = one pass of the hot path over one batch:

    LR = lr_from_hr(HR)  ->  fake = G(LR)  ->  loss = 10 * mean((fake - HR)^2)   (identity
    extractor, config.py:158-162)  ->  backward (dgrad + wgrad of every layer)  ->  gradient
    all-reduce over RCCL when N > 1  ->  Adam step (lr 1e-5, config.py:38,293; the fused multi-tensor
    step of optim.py).

Inputs are resident in HBM before the timed region.  value = HR patches/s over all ranks.
Also reported on the same JSON line: the roofline of the dominant kernel (the 3x3 64->64 trunk
convolution, measured live with HIP events on the launch stream) and a CPU baseline (the oracle
timed on this host's cores on a bounded sample; rank 0, N=1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = 'single-image-super-resolution_amd'

B, HR, LR = 16, 192, 96
PEAK_F32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0                 # same guide: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)


def sub(name):
    return importlib.import_module(PKG + '.' + name)


def make_step(device, rank, world):
    mg, utils = sub('model_generator'), sub('utils')
    torch.manual_seed(0)                                      # identical replicas on every rank
    net = mg.Generator(16, 64, 256, [2], use_sn=True).to(device).train()
    opt = sub('optim').Adam(net.parameters(), lr=1e-5, betas=(0.9, 0.999))     # fused multi-tensor step (row f1)
    g = torch.Generator().manual_seed(rank)                   # a different shard of patches per rank
    hr = (torch.rand((B, 3, HR, HR), generator=g) * 2 - 1).to(device)
    reducer = sub('distributed').GradReducer(list(net.parameters()), world) if world > 1 else None

    def fwd_bwd():
        lr = utils.lr_from_hr(hr, (LR, LR), device=device)
        fake = net(lr)
        loss = 10.0 * torch.mean(torch.pow(hr - fake, 2))
        net.zero_grad(set_to_none=True)
        loss.backward()
        return loss

    state = {'graphed': None}

    def capture():
        """Record the forward+backward launch sequence (~350 kernels) once into a HIP graph (graph.GraphedStep);
        afterwards a step replays it -- same kernels, same work, no per-launch host cost.  Gradients are static
        tensors of the graph's pool; the gradient all-reduce and the Adam step stay ordinary stream work."""
        state['graphed'] = sub('graph').GraphedStep(fwd_bwd)

    def step():
        loss = state['graphed']() if state['graphed'] is not None else fwd_bwd()
        if reducer is not None:
            reducer.all_reduce_mean()
        opt.step()
        return loss

    return step, net, capture


def dominant_kernel_roofline(device, precision, iters=40):
    """Live measurement of the dominant kernel: the trunk convolution 3x3, 64->64 at (16, 96, 96),
    launched exactly as inside the step (BatchNorm-apply + PReLU prologue, BatchNorm-statistics
    epilogue).  Algorithmic work per launch = 2*N*H*W*Cout*Cin*9 flops (10.87 GFLOP)."""
    E = sub('engine')
    torch.manual_seed(1)
    w = (torch.rand(64, 64, 3, 3, device=device) - 0.5) * 0.1
    bias = torch.zeros(64, device=device)

    class Ref:
        pass
    ref = Ref()
    ref.weight, ref.bias, ref.u, ref.v, ref.geom = w, bias, None, None, E.ConvGeom(64, 64, 3, 1, 1)
    preps, keep = E.prepare_weights([(ref, B, LR, LR)], training=True)
    x = torch.rand(B, LR, LR, 64, device=device) * 2 - 1
    sc = torch.rand(64, device=device) + 0.5
    sh = torch.rand(64, device=device) - 0.5
    slope = torch.full((1,), 0.25, device=device)
    op = E.Operand.affine_act(x, sc, sh, slope)
    out = torch.empty_like(x)
    for _ in range(5):
        E.conv_forward(preps[0], op, bias=bias, stats=True, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()                       # events on the current stream = the stream the kernel runs on
    for _ in range(iters):
        E.conv_forward(preps[0], op, bias=bias, stats=True, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * B * LR * LR * 64 * 64 * 9
    if precision == 'bf16':
        # bf16 matrix cores: the layer is HBM-bound.  Algorithmic bytes per launch (the input
        # and the output tensor cross HBM once each; this round both are stored fp32): 2 * B*h*w*64 * 4 B
        nbytes = 2.0 * B * LR * LR * 64 * 4
        achieved = nbytes / (ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters (FETCH_SIZE, WRITE_SIZE): bench.py cannot run rocprofv3 on
        # itself, so the figure collected for this kernel with separate --pmc passes (tools/prof_conv.py,
        # gfx950 corrections applied) is committed under profiles/ and reported here; null if absent
        traffic = None
        try:
            with open(os.path.join(ROOT, 'profiles', 'r01_traffic_trunk_conv.json')) as fh:
                traffic = json.load(fh)['traffic_bytes_per_launch']
        except (OSError, KeyError, ValueError):
            pass
        return {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                'frac': round(achieved / PEAK_HBM_GBS, 4), 'traffic': traffic,
                'kernel': 'conv_mfma_bf16_kernel<*,2,1> (3x3 64->64 trunk conv, fp32 tensors in HBM)',
                'launch_ms': round(ms, 4), 'alg_bytes_per_launch': nbytes,
                'mfma_tflops': round(flops / (ms * 1e-3) / 1e12, 1)}
    achieved = flops / (ms * 1e-3) / 1e12
    return {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4), 'traffic': None,
            'kernel': 'conv_mfma_f32_kernel<*,2,1> (3x3 64->64 trunk conv)', 'launch_ms': round(ms, 4),
            'alg_flops_per_launch': flops}


def cpu_baseline(max_seconds=30.0):
    """The oracle (CPU restatement, torch-CPU fp32) on the same graph, bounded sample."""
    from oracle import init as oinit, models as omodels, ops as oops          # checker/baseline only
    mg = sub('model_generator')
    torch.manual_seed(0)
    net = mg.Generator(16, 64, 256, [2], use_sn=True)              # parameter container only (CPU)
    state = {k: v.detach().clone() for k, v in net.state_dict().items()}
    pk = omodels.param_keys(state)
    for k in pk:
        state[k].requires_grad_(True)
    opt = torch.optim.Adam([state[k] for k in pk], lr=1e-5, betas=(0.9, 0.999))
    hr = oinit.synth_input((B, 3, HR, HR), 0)
    cores = torch.get_num_threads()

    def step():
        lr = oops.lr_from_hr(hr, (LR, LR))
        fake, new = omodels.generator_forward(state, lr, (2,), True, 0)
        loss = 10.0 * torch.mean(torch.pow(hr - fake, 2))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        with torch.no_grad():
            for k, v in new.items():
                state[k] = v
    t0 = time.time()
    step()                                                        # warm-up
    warm = time.time() - t0
    n, t0 = 0, time.time()
    while n < 1 or (time.time() - t0 + warm * 1.2 < max_seconds and n < 4):
        step()
        n += 1
    dt = (time.time() - t0) / n
    return {'value': round(B / dt, 3), 'unit': 'HR patches/s', 'cores': cores, 'kind': 'port',
            'sample': '%d timed step(s) of the same B=%d HR=%d graph after 1 warm-up, oracle/ on torch-CPU fp32'
                      % (n, B, HR)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo only for 1-GPU rehearsals)')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly instead of replaying a HIP graph')
    ap.add_argument('--precision', choices=['bf16', 'fp32'], default=os.environ.get('SISR_PRECISION', 'bf16'),
                    help='bf16: bf16 matrix cores with fp32 accumulate (BASELINE config 1); fp32: exact-fp32 parity build')
    args = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    local = local % max(1, torch.cuda.device_count())           # rehearsal: several ranks may share one GPU (gloo)
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        import torch.distributed as dist
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)     # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(args.backend)
    sub('engine').set_precision(args.precision)
    step, net, capture = make_step(device, rank, world)
    graphed = False
    if not args.no_graph:
        try:
            capture()
            graphed = True
        except Exception as e:                                   # noqa: BLE001
            print('HIP graph capture failed (%s: %s); running eagerly' % (type(e).__name__, e), file=sys.stderr)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        roof = dominant_kernel_roofline(device, args.precision)
        rec = {
            'metric': 'HR patches/sec (x2 generator fwd+bwd, LR 96x96 -> SR 192x192)',
            'value': round(world * B * args.steps / dt, 2), 'unit': 'HR patches/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16' if args.precision == 'bf16' else 'f32', 'data': 'synthetic',
            'config': {'workload': 'SRGAN x2 generator (16 blocks, 64 features, spectral norm) fwd+bwd+Adam with '
                                   'bicubic LR degradation and pixel-MSE x10, per-GPU batch 16 HR 192x192 patches',
                       'per_gpu_batch': B, 'hr': HR, 'lr': LR, 'parallelism': 'dp%d' % world, 'hip_graph': graphed,
                       'final_loss': round(float(loss.item()), 6)},
            'roofline': roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            rec['cpu_baseline'] = cpu_baseline()
        print(json.dumps(rec))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
