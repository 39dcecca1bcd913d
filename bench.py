#!/usr/bin/env python3
"""Benchmark of the SRGAN hot path on MI355X (driver contract: see the task statement).

Workload (BASELINE.json north_star headline): the x2 generator
``Generator(16, 64, 256, [2], use_sn=True)`` trained on synthetic U(-1,1) HR patches,
per GPU B=16 patches of 192x192 (LR 96x96).  One step = one pass of the hot path over one batch:

    LR = lr_from_hr(HR)  ->  fake = G(LR)  ->  loss = 10 * mean((fake - HR)^2)   (identity
    extractor, config.py:158-162)  ->  backward (dgrad + wgrad of every layer), the gradient
    all-reduce over RCCL launched per bucket from inside the backward pass when N > 1  ->  Adam step
    (lr 1e-5, config.py:38,293; the fused multi-tensor step of optim.py).

Inputs are resident in HBM before the timed region.  value = HR patches/s over all ranks.

ONE invocation measures BOTH arithmetic builds, each over exactly --steps timed steps after --warmup:
  * top level of the JSON line: the **fp32 parity build** -- exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), the
    reference's own precision and the build every 1e-3 golden test runs; MFMA-bound (peak 157.3 TFLOP/s);
  * ``perf_build``: the **bf16 matrix-core build** (SURVEY 8d / BASELINE.json config 2 name bf16 for the perf
    configs; fp32 accumulate and statistics) -- HBM-bound, the regime the north star's 60 % target is stated in.
Each record carries the roofline of the kernel with the LARGEST total time per step (trunk forward conv, trunk
data-gradient conv and trunk weight gradient are probed live with HIP events on the launch stream) and a
whole-step fraction (SURVEY 8d algorithmic flops / bytes over the measured step time).  ``cpu_baseline``: the
oracle (CPU restatement, torch-CPU fp32) timed on this host's cores on the same batch (1 warm-up + best of 2 timed steps,
~6 s each on a 16-core share); rank 0, N=1 only.

``configs`` (``--configs none`` skips it): BASELINE.json's configs 2-5 as iteration rates over all N ranks -- a full SRGAN
iteration (train.py:45-108: ONE G forward, D step on real + detached fake, G step through D and the VGG content
extractor, both fused Adam steps; with N > 1 the gradient buckets of BOTH networks all-reduced from inside their backward
schedules) of each config's networks at its sizes in the bf16 build, replayed from one segmented HIP graph, with the
algorithmic flops / bytes of one iteration and the fractions of the MI355X peaks they amount to.  N = 1 times cfg2-cfg5; N > 1
times the configs BASELINE.json calls "8xMI355X DP" (cfg3, cfg4, cfg5).  BASELINE.json's metric string reads "96^2 crops, x4":
that workload is ``configs.cfg3``; the top-level value is the north star's headline (x2 generator, LR 96 -> SR 192).

``--gpus N`` with N > 1 starts the N rank processes itself (``torch.distributed.run``, one per GPU, rendezvous
on 127.0.0.1) BEFORE anything touches the GPU, unless it already runs under such a launcher (WORLD_SIZE set).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = 'single-image-super-resolution_amd'

B, HR, LR = 16, 192, 96
PEAK_F32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0                 # same guide: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
T_ELEMS = B * LR * LR * 64            # T = B*h*w*64 elements
STEP_FLOPS = 3 * 25.555e9 * B         # 76.7 GFLOP per sample at LR 96^2 (fwd + dgrad + wgrad)
STEP_T_UNITS = 364.7                  # fwd 127 T + bwd 237 T (+ 14 t3) -> 364.7 T of tensor traffic


def sub(name):
    return importlib.import_module(PKG + '.' + name)


# ------------------------------------------------------------------------------------------------------------
# N-rank launch
# ------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """Re-run this script as N fresh rank processes (nothing in this process has touched the GPU yet: no torch.cuda
    call, no HIP call -- a requirement of the pool, and why this is a child launch and not an exec)."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------------------------
# the step
# ------------------------------------------------------------------------------------------------------------
def make_step(device, rank, world, use_graph, log):
    import torch
    mg, utils, G = sub('model_generator'), sub('utils'), sub('graph')
    torch.manual_seed(0)                                      # identical replicas on every rank
    net = mg.Generator(16, 64, 256, [2], use_sn=True).to(device).train()
    opt = sub('optim').Adam(net.parameters(), lr=1e-5, betas=(0.9, 0.999))     # fused multi-tensor step (row f1)
    g = torch.Generator().manual_seed(rank)                   # a different shard of patches per rank
    hr = (torch.rand((B, 3, HR, HR), generator=g) * 2 - 1).to(device)
    reducer = sub('distributed').GradReducer(net, world) if world > 1 else None

    def fwd_bwd():
        lr = utils.lr_from_hr(hr, (LR, LR), device=device)
        fake = net(lr)
        loss = 10.0 * torch.mean(torch.pow(hr - fake, 2))
        net.zero_grad(set_to_none=True)
        loss.backward()          # with a reducer: every finished bucket is announced from inside the backward pass
        return loss

    graphed = None
    if use_graph:
        # Record the forward+backward launch sequence once into HIP graphs (graph.GraphedStep); afterwards a step
        # replays them -- same kernels, same work, no per-launch host cost.  With a reducer the capture is cut at
        # every bucket boundary so the bucket's all-reduce is issued on the side stream between two segments.
        try:
            if reducer is not None:
                reducer.capture_mode(True)
            graphed = G.GraphedStep(fwd_bwd, between=reducer.launch_bucket if reducer is not None else None)
        except G.GraphCaptureError as e:
            log('%s -- launching eagerly instead' % e)
            graphed = None
        finally:
            if reducer is not None:
                reducer.capture_mode(False)

    def step():
        if graphed is not None:
            loss = graphed()
            if reducer is not None:
                reducer.launch_remaining()
        else:
            loss = fwd_bwd()
        if reducer is not None:
            reducer.finish()      # the compute stream waits for the side stream's reductions; no host sync
        opt.step()
        return loss

    return step, net, graphed is not None


def timed_run(device, rank, world, precision, steps, warmup, use_graph, log):
    import torch
    import torch.distributed as dist
    sub('engine').set_precision(precision)
    step, net, graphed = make_step(device, rank, world, use_graph, log)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())
    del step, net
    torch.cuda.empty_cache()
    return dt, graphed, final_loss


# ------------------------------------------------------------------------------------------------------------
# live per-kernel probes (HIP events on the stream the kernels are launched on)
# ------------------------------------------------------------------------------------------------------------
def _time_launches(fn, iters):
    import torch
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()                       # torch's current stream = the stream engine.py launches on
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def _recorded_traffic(kernel_key):
    """HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, gfx950
    corrections applied): bench.py cannot run rocprofv3 on itself, so the figure is the one RECORDED under profiles/
    for this kernel by tools/pmc_collect.sh (newest round first), with the commit it was collected at; null when
    absent."""
    for name in ('r04_traffic.json', 'r03_traffic.json', 'r02_traffic.json'):
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as fh:
                rec = json.load(fh)[kernel_key]
            return rec['traffic_bytes_per_launch'], 'recorded: profiles/%s @ %s' % (name, rec.get('commit', '?'))
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None, None


def kernel_rooflines(device, precision, iters=40, only=None):
    """The three trunk kernels (3x3, 64->64 at (16, 96, 96)) launched exactly as inside the step:
       fwd   -- BatchNorm-apply + PReLU prologue, BatchNorm-statistics epilogue           (33 per step)
       dgrad -- BatchNorm-backward prologue, residual add, fused BatchNorm-backward sums  (33 per step)
       wgrad -- weight + bias gradient incl. its deterministic partial-slab reduction     (34 per step)
    Algorithmic work per launch: 2*N*H*W*64*64*9 = 10.87 GFLOP; bytes = the tensors a launch must
    move once at the storage type in use (fwd: in + out = 2T; dgrad: dy, BN input, residual, out, next BN input
    = 5T; wgrad: x, dy, BN input = 3T)."""
    import torch
    E, L = sub('engine'), sub('_lib')
    E.set_precision(precision)
    torch.manual_seed(1)
    w = (torch.rand(64, 64, 3, 3, device=device) - 0.5) * 0.1
    bias = torch.zeros(64, device=device)

    class Ref:
        pass
    ref = Ref()
    ref.weight, ref.bias, ref.u, ref.v, ref.geom = w, bias, None, None, E.ConvGeom(64, 64, 3, 1, 1)
    preps, keep = E.prepare_weights([(ref, B, LR, LR)], training=True)
    p = preps[0]
    act_dtype = E.act_dtype() if hasattr(E, 'act_dtype') else torch.float32
    elt = 2 if act_dtype == torch.bfloat16 else 4

    def rnd(*shape):
        return (torch.rand(*shape, device=device) * 2 - 1).to(act_dtype)
    x, x2, gy, res = rnd(B, LR, LR, 64), rnd(B, LR, LR, 64), rnd(B, LR, LR, 64), rnd(B, LR, LR, 64)
    sc, sh = torch.rand(64, device=device) + 0.5, torch.rand(64, device=device) - 0.5
    consts = torch.stack([sc, sh, torch.rand(64, device=device) - 0.5, torch.rand(64, device=device) + 0.5])
    q = torch.rand(3, 64, device=device) - 0.5
    slope = torch.full((1,), 0.25, device=device)
    fwd_op = E.Operand.affine_act(x, sc, sh, slope)
    out = torch.empty_like(x)
    dy_op = E.Operand(gy, tuple(gy.shape), pro=L.PRO_BNBWD, x2=x2, pa=q[0], pb=q[1], pd=q[2])
    fused = E.can_fuse_bn_backward(p)
    flops = 2.0 * B * LR * LR * 64 * 64 * 9
    T = T_ELEMS * elt
    roles = {
        'fwd': (lambda: E.conv_forward(p, fwd_op, bias=bias, stats=True, out=out), 33, 2 * T),
        'dgrad': (lambda: E.conv_dgrad(p, dy_op, res=res, bnb=(x, consts, slope) if fused else None), 33, 5 * T),
        'wgrad': (lambda: E.conv_wgrad(p, fwd_op, dy_op), 34, 3 * T),
    }
    fam = 'bf16' if precision == 'bf16' else 'f32'
    names = {'fwd': 'conv_mfma_%s_kernel (trunk 3x3 64->64, forward role)' % fam,
             'dgrad': 'conv_mfma_%s_kernel (trunk 3x3 64->64, data-gradient role)' % fam,
             'wgrad': 'wgrad_mfma_%s_kernel + slab_reduce_kernel (trunk 3x3 64->64)' % fam}
    if precision == 'bf16' and elt == 2 and os.environ.get('SISR_TRUNK', '1') != '0':
        # bf16 tensors: the persistent trunk kernels (conv_trunk.hip, wgrad_trunk.hip) take this geometry
        names = {'fwd': 'conv_trunk_fwd_kernel (trunk 3x3 64->64, forward role)',
                 'dgrad': 'conv_trunk_bwd_kernel (trunk 3x3 64->64, data-gradient role)',
                 'wgrad': 'wgrad_trunk_kernel + slab_reduce_kernel (trunk 3x3 64->64)'}
        if os.environ.get('SISR_TRUNK_WGRAD', '1') == '0':
            names['wgrad'] = 'wgrad_mfma_bf16_kernel + slab_reduce_kernel (trunk 3x3 64->64)'
    if precision in ('fp32', 'bf16x3') and os.environ.get('SISR_TRUNK', '1') != '0':
        # fp32 tensors: the persistent fp32-tensor kernels (conv_trunk_f32.hip, wgrad_trunk_f32.hip)
        if os.environ.get('SISR_TRUNK_F32CONV', '1') != '0':
            names['fwd'] = 'conv_trunk_f32_kernel (trunk 3x3 64->64, forward role)'
            names['dgrad'] = 'conv_trunk_f32_kernel (trunk 3x3 64->64, data-gradient role)'
        if os.environ.get('SISR_TRUNK_WGRAD', '1') != '0':
            names['wgrad'] = 'wgrad_trunk_f32_kernel + slab_reduce_kernel (trunk 3x3 64->64)'
        if precision == 'bf16x3':
            names = {r: n.replace('_f32_kernel', '_f32_kernel<SPLIT>') for r, n in names.items()}
    out_rec = {}
    if only is not None:                         # developer tools (tools/trace_conv.py, tools/prof_conv.py): one role
        roles = {r: v for r, v in roles.items() if r in only}
    for role, (fn, per_step, nbytes) in roles.items():
        ms = _time_launches(fn, iters)
        if precision in ('bf16', 'bf16x3'):
            achieved = nbytes / (ms * 1e-3) / 1e9
            traffic, src = _recorded_traffic('%s_%s' % ('split' if precision == 'bf16x3' else fam, role))
            rec = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                   'frac': round(achieved / PEAK_HBM_GBS, 4), 'traffic': traffic, 'traffic_source': src,
                   'alg_bytes_per_launch': nbytes, 'mfma_tflops': round(flops / (ms * 1e-3) / 1e12, 1)}
        else:
            achieved = flops / (ms * 1e-3) / 1e12
            traffic, src = _recorded_traffic('%s_%s' % ('split' if precision == 'bf16x3' else fam, role))
            rec = {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                   'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4), 'traffic': traffic, 'traffic_source': src,
                   'alg_flops_per_launch': flops, 'alg_bytes_per_launch': nbytes}
        rec.update({'kernel': names[role], 'launch_ms': round(ms, 4), 'launches_per_step': per_step,
                    'ms_per_step': round(ms * per_step, 3), 'storage': 'bf16' if elt == 2 else 'f32'})
        out_rec[role] = rec
    top = max(out_rec, key=lambda r: out_rec[r]['ms_per_step'])
    main = dict(out_rec[top])
    main['others'] = {r: {k: v[k] for k in ('kernel', 'launch_ms', 'ms_per_step', 'frac', 'achieved', 'unit')}
                      for r, v in out_rec.items() if r != top}
    return main, elt


def whole_step(precision, elt, ms_per_step):
    """SURVEY 8d whole-step figures over the measured step time: flops 1,227 GFLOP; tensor traffic 364.7 T."""
    sec = ms_per_step * 1e-3
    stored = STEP_T_UNITS * T_ELEMS * elt
    north = STEP_T_UNITS * T_ELEMS * 2                                   # the north star counts bf16 tensors (6.9 GB)
    rec = {'alg_flops': STEP_FLOPS, 'tflops': round(STEP_FLOPS / sec / 1e12, 1),
           'alg_bytes_at_storage_type': stored, 'gbs_at_storage_type': round(stored / sec / 1e9, 1),
           'frac_hbm_at_storage_type': round(stored / sec / 1e9 / PEAK_HBM_GBS, 4),
           'frac_hbm_north_star_bf16_bytes': round(north / sec / 1e9 / PEAK_HBM_GBS, 4)}
    if precision in ('fp32', 'bf16x3'):
        rec['frac_mfma_f32'] = round(STEP_FLOPS / sec / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
    return rec


# ------------------------------------------------------------------------------------------------------------
# CPU baseline
# ------------------------------------------------------------------------------------------------------------
def _host_cpus():
    """CPUs this process may actually use: the cgroup quota when there is one (a GPU box gives a container a share of
    the host's cores), else the affinity mask, else os.cpu_count()"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            with open(path) as fh:
                parts = fh.read().split()
            if path.endswith('cpu.max'):
                if parts[0] != 'max':
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as fh2:
                        n = min(n, max(1, q // int(fh2.read().split()[0])))
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    return n


def cpu_baseline(sample_b=B, repeats=2):
    """The oracle (CPU restatement, torch-CPU fp32) on the SAME graph and the SAME batch as a GPU step: all B = 16
    patches of HR 192 (BatchNorm sees the batch the GPU step sees), best of `repeats` timed steps after one warm-up,
    on all host threads torch uses.  ~10 s of CPU per step on the GPU node's host."""
    import torch
    from oracle import init as oinit, models as omodels, ops as oops          # checker/baseline only
    mg = sub('model_generator')
    torch.manual_seed(0)
    net = mg.Generator(16, 64, 256, [2], use_sn=True)              # parameter container only (CPU)
    state = {k: v.detach().clone() for k, v in net.state_dict().items()}
    pk = omodels.param_keys(state)
    for k in pk:
        state[k].requires_grad_(True)
    opt = torch.optim.Adam([state[k] for k in pk], lr=1e-5, betas=(0.9, 0.999))
    hr = oinit.synth_input((sample_b, 3, HR, HR), 0)
    cores = _host_cpus()
    torch.set_num_threads(cores)          # (os.cpu_count() of a container is the host's: 128 threads on a 16-CPU share thrash)

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
    step()                                                        # warm-up
    best = float('inf')
    for _ in range(repeats):
        t0 = time.time()
        step()
        best = min(best, time.time() - t0)
    return {'value': round(sample_b / best, 3), 'unit': 'HR patches/s', 'cores': cores, 'kind': 'port',
            'sample': 'best of %d timed steps (after 1 warm-up) of the same graph on the full batch of %d HR-%d patches '
                      '(%.1f s per step), oracle/ on torch-CPU fp32' % (repeats, sample_b, HR, best)}


# ------------------------------------------------------------------------------------------------------------
# BASELINE.json configs 2-5: one-GPU iteration rates
# ------------------------------------------------------------------------------------------------------------
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16
FEATS, STRIDES = [64, 64, 128, 128, 256, 256, 512, 512], [1, 2, 1, 2, 1, 2, 1, 2]      # config.py:81-82
VGG_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M')
CONFIGS = {
    # name: (BASELINE.json config string, HR, LR, generator kind, VGG mask)
    'cfg2': ('CelebA 96-crop x2 SRGAN (G+D+VGG22 content loss) bf16', 96, 48, 'x2', 0b00010),
    'cfg3': ('CelebA 96-crop x4 SRGAN, VGG54 content loss', 96, 24, 'x4_suffix', 0b10000),
    'cfg4': ('Flickr8k 192-crop x4 SRGAN, spectral-norm D', 192, 48, 'x4_suffix', 0b10000),
    'cfg5': ('Progressive x8 (model_generator_progressive, x2->x4->x8 stacked upsample)', 192, 24, 'progressive_x8', 0b10000),
}


def _conv_layers(kind, hr, lr, mask):
    """[(net, cin, cout, k, h_in, w_in, h_out, w_out)] of every conv of a config's G, D and VGG stack, and D's two Linear
    layers as (net, 'fc', in, out): the shapes the flop / byte model below is evaluated on"""
    out = []
    if kind == 'progressive_x8':
        out.append(('g', 3, 64, 9, lr, lr, lr, lr))
        out += [('g', 64, 64, 3, lr, lr, lr, lr)] * 33
        c, r = 64, lr
        for _ in range(3):
            out.append(('g', c, c, 3, r, r, r, r))             # conv(n -> n) + PixelShuffle(2): n / 4 channels at 2 r
            c, r = c // 4, 2 * r
        out.append(('g', c, 3, 3, r, r, r, r))
    else:
        out.append(('g', 3, 64, 9, lr, lr, lr, lr))
        out += [('g', 64, 64, 3, lr, lr, lr, lr)] * 33
        r = lr
        for _ in range(1 if kind == 'x2' else 2):
            out.append(('g', 64, 256, 3, r, r, r, r))
            r *= 2
        out.append(('g', 64, 3, 3, r, r, r, r))
    c, r = 3, hr
    for f, st in zip(FEATS, STRIDES):
        out.append(('d', c, f, 3, r, r, r // st, r // st))
        c, r = f, r // st
    out.append(('d', 'fc', c * r * r, 1024))
    out.append(('d', 'fc', 1024, 1))
    kept = max(i for i in range(5) if mask & (1 << i))
    n_layers = (3, 8, 17, 26, 35)[kept]                       # features[:k] (model_content_extractor.py:41-43)
    c, r, idx = 3, hr, 0
    for v in VGG_CFG:
        if idx >= n_layers:
            break
        if v == 'M':
            r //= 2
            idx += 1
        else:
            out.append(('v', c, v, 3, r, r, r, r))
            c = v
            idx += 2
    return out


def config_work(kind, hr, lr, mask):
    """Algorithmic work of ONE SRGAN iteration (train.py:45-108, empty replay list), SURVEY 8d conventions:
    flops = 2 MACs; G: forward once + data and weight gradients (3 F_G); D: forward on real, on the detached fake and on
    the fake of the G step, data + weight gradients in the D step, data gradient in the G step (8 F_D); frozen VGG:
    forward on real and fake + data gradient (3 F_V).  bytes: every conv reads its input and writes its output once per
    pass (backward: dy + the saved activation + dx for the data gradient, x + dy for the weight gradient), bf16 for
    multi-channel tensors and fp32 for 3-channel images; the Linear layers stream their fp32 weights once per pass."""
    passes = {'g': (1, 1, 1), 'd': (3, 3, 2), 'v': (2, 1, 0)}     # forward, data-gradient, weight-gradient passes
    flops = nbytes = 0.0
    for lay in _conv_layers(kind, hr, lr, mask):
        net = lay[0]
        fw, dg, wg = passes[net]
        if lay[1] == 'fc':
            _, _, k, n = lay
            flops += 2.0 * B * k * n * (fw + dg + wg)
            nbytes += 4.0 * k * n * (fw + dg + wg)
            continue
        _, cin, cout, ks, hi, wi, ho, wo = lay
        flops += 2.0 * B * ho * wo * cin * cout * ks * ks * (fw + dg + wg)
        ei, eo = (4 if cin <= 4 else 2), (4 if cout <= 4 else 2)
        i_b, o_b = B * hi * wi * cin * ei, B * ho * wo * cout * eo
        nbytes += fw * (i_b + o_b) + dg * (2 * o_b + i_b) + wg * (i_b + o_b)
    return flops, nbytes


def _fresh_allocator():
    """every build / config of one bench process starts from an empty caching allocator: where the previous workload left its
    blocks decides where this one's tensors land, and a replayed graph has been seen 15 % slower for it (cfg5 after the three
    builds: 21 ms instead of 18 ms; not inside any timed region)"""
    import gc
    import torch
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def make_config_iteration(name, device, rank, world, use_graph, log):
    """One full SRGAN iteration of a BASELINE.json config (train.py:45-108, empty replay list) as a callable:
        fake = G(lr)                      ONCE (train.py:53), its autograd graph kept for the G step
        D step: D(real), D(fake.detach()), BCE, backward, [gradient all-reduce], Adam           (train.py:58-75)
        G step: D(fake) * 5e-2 + feature-MSE through the VGG extractor, backward, [all-reduce], Adam   (train.py:82-108)
    Under HIP-graph replay the whole iteration is ONE GraphedStep cut into segments: at 'd_step' (the discriminator's optimizer
    step runs between two replayed segments, so the G step sees the stepped D exactly as train.py does) and, with N > 1 ranks, at
    every gradient bucket either network announces from inside its backward schedule (its all-reduce is issued on the side
    stream between the segments).  -> (iteration(), info dict)"""
    import torch
    desc, hr_sz, lr_sz, kind, mask = CONFIGS[name]
    E, G = sub('engine'), sub('graph')
    mg, md, mce, ut, op = (sub('model_generator'), sub('model_discriminator'), sub('model_content_extractor'), sub('utils'),
                           sub('optim'))
    E.set_precision('bf16')
    _fresh_allocator()
    torch.manual_seed(0)                                         # identical replicas on every rank
    if kind == 'progressive_x8':
        mp = sub('model_generator_progressive')
        g1 = mp.GeneratorSuffix(mp.GeneratorProgresiveBase(16, 64), 64)
        g2 = mp.GeneratorSuffix(g1.beginning, 16)
        net_g = mp.GeneratorSuffix(g2.beginning, 4)
    else:
        net_g = mg.Generator(16, 64, 256, [2], use_sn=True)
        if kind == 'x4_suffix':
            net_g = mg.GeneratorSuffix(net_g)
    net_g = net_g.to(device).train()
    net_d = md.Discriminator((3, hr_sz, hr_sz), FEATS, STRIDES).to(device).train()
    ext = mce.MaskedVGG(mask, pretrained=False).to(device)       # synthetic weights: the pretrained ones need a remote fetch
    og, od = op.Adam(net_g.parameters(), lr=1e-5), op.Adam(net_d.parameters(), lr=1e-5)
    crit = torch.nn.BCELoss()
    gen = torch.Generator().manual_seed(rank)                    # a different shard of patches per rank
    hr = (torch.rand(B, 3, hr_sz, hr_sz, generator=gen) * 2 - 1).to(device)
    ones, red, zeros = torch.ones(B, device=device), torch.full((B,), .9, device=device), torch.zeros(B, device=device)
    red_g = red_d = None
    if world > 1:
        dist_m = sub('distributed')
        red_g, red_d = dist_m.GradReducer(net_g, world, name='G:'), dist_m.GradReducer(net_d, world, name='D:')

    def both():
        lr = ut.lr_from_hr(hr, (lr_sz, lr_sz), device=device)
        fake = net_g(lr)                                         # train.py:53 -- once
        net_d.zero_grad()
        err_d = crit(net_d(hr).view(-1), red) + crit(net_d(fake.detach()).view(-1), zeros)
        err_d.backward()
        if not G.segment_boundary('d_step'):                     # replay: the host runs d_step() between the two segments
            d_step()                                             # eager: right here
        if red_d is not None:
            red_d.enabled = False                                # the G step's pass through D: those gradients are discarded
        net_g.zero_grad()
        err_g = crit(net_d(fake).view(-1), ones) * 5e-2 + torch.mean(torch.pow(ext(hr) - ext(fake), 2))
        err_g.backward()
        if red_d is not None:
            red_d.enabled = True
        return err_d, err_g

    def d_step():
        if red_d is not None:
            red_d.finish()               # whatever the backward schedule did not announce; the compute stream waits for the exchange
        od.step()

    def between(tag):
        if tag == 'd_step':
            d_step()
        elif red_d is not None and red_d.owns(tag):
            red_d.launch_bucket(tag)
        elif red_g is not None and red_g.owns(tag):
            red_g.launch_bucket(tag)

    graphed = None
    if use_graph:
        try:
            for r in (red_g, red_d):
                if r is not None:
                    r.capture_mode(True)
            graphed = G.GraphedStep(both, between=between)
        except G.GraphCaptureError as e:
            log('%s: %s -- launching eagerly instead' % (name, e))
            graphed = None
        finally:
            for r in (red_g, red_d):
                if r is not None:
                    r.capture_mode(False)

    def iteration():
        out = graphed() if graphed is not None else both()
        if red_g is not None:
            if graphed is not None:
                red_g.launch_remaining()
            red_g.finish()
        og.step()
        return out
    info = {'config': desc, 'hr': hr_sz, 'lr': lr_sz, 'generator': kind, 'vgg_mask': mask, 'per_gpu_batch': B, 'dtype': 'bf16',
            'hip_graph': graphed is not None, 'g_forward_runs': 1,
            'graph_segments': len(graphed.graphs) if graphed is not None else 0}
    keep = (net_g, net_d, ext, og, od, graphed, red_g, red_d)
    return iteration, info, keep


def bench_config(name, device, iters, log, rank=0, world=1, use_graph=True):
    """-> record of one config: ms per iteration, HR patches/s (all ranks), its algorithmic work and roofline fractions"""
    import torch
    import torch.distributed as dist
    desc, hr_sz, lr_sz, kind, mask = CONFIGS[name]
    iteration, info, keep = make_config_iteration(name, device, rank, world, use_graph, log)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(3):
        iteration()
    barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        iteration()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    sec = dt / iters
    flops, nbytes = config_work(kind, hr_sz, lr_sz, mask)
    traffic, src = _recorded_traffic('%s_iteration' % name)
    rec = dict(info)
    rec.update({'iters': iters, 'ms_per_iteration': round(sec * 1e3, 3), 'value': round(world * B / sec, 1), 'unit': 'HR patches/s',
                'n_gpus': world, 'alg_flops': flops, 'alg_bytes': nbytes,
                'roofline': {'bound': 'hbm', 'achieved': round(nbytes / sec / 1e9, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                             'frac': round(nbytes / sec / 1e9 / PEAK_HBM_GBS, 4), 'traffic': traffic, 'traffic_source': src,
                             'mfma_tflops': round(flops / sec / 1e12, 1),
                             'frac_mfma_bf16': round(flops / sec / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)}})
    del iteration, keep
    torch.cuda.empty_cache()
    return rec


# ------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo for rehearsals)')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly instead of replaying HIP graphs')
    ap.add_argument('--precision', choices=['both', 'fp32', 'bf16x3', 'bf16'], default=os.environ.get('SISR_BENCH_PRECISION', 'both'),
                    help='both (default): fp32 parity build at top level + bf16 build as perf_build; or one build only')
    ap.add_argument('--configs', default='all', help="BASELINE.json configs timed as full SRGAN iteration rates over all ranks: "
                    "'all' (N = 1: cfg2-cfg5; N > 1: cfg3-cfg5), 'none' or a comma list of cfg2,cfg3,cfg4,cfg5")
    ap.add_argument('--config-iters', type=int, default=10)
    ap.add_argument('--dry-run-ranks', action='store_true',
                    help='launcher self-test: start the ranks, all-reduce a 1 over them and print the count (no GPU work)')
    args = ap.parse_args()

    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))           # BEFORE any torch.cuda / HIP call
    world = int(env_world) if env_world is not None else 1
    if world != args.gpus:
        print('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))

    import torch
    import torch.distributed as dist

    def log(msg):
        print('[bench rank %d] %s' % (rank, msg), file=sys.stderr, flush=True)

    if args.dry_run_ranks:
        if world > 1:
            dist.init_process_group('gloo' if args.backend != 'nccl' or not torch.cuda.is_available() else args.backend)
            t = torch.ones(1)
            if dist.get_backend() == 'nccl':
                torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
                t = t.cuda()
            dist.all_reduce(t)
            n = int(t.item())
            dist.destroy_process_group()
        else:
            n = 1
        if rank == 0:
            print(json.dumps({'dry_run': True, 'n_gpus': n, 'world': world}))
        return

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    local = local % max(1, torch.cuda.device_count())           # rehearsal: several ranks may share one GPU (gloo)
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)     # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(args.backend)

    builds = ['fp32', 'bf16x3', 'bf16'] if args.precision == 'both' else [args.precision]
    records = {}
    for prec in builds:
        _fresh_allocator()
        dt, graphed, final_loss = timed_run(device, rank, world, prec, args.steps, args.warmup, not args.no_graph, log)
        if rank == 0:
            ms = dt / args.steps * 1e3
            roof, elt = kernel_rooflines(device, prec)
            records[prec] = {
                'value': round(world * B * args.steps / dt, 2), 'unit': 'HR patches/s', 'ms_per_step': round(ms, 3),
                'dtype': {'fp32': 'f32', 'bf16x3': 'bf16x3 (fp32 tensors; hi/lo bf16 MFMA operands, fp32 accumulate)', 'bf16': 'bf16'}[prec], 'steps': args.steps, 'warmup': args.warmup,
                'hip_graph': graphed, 'final_loss': round(final_loss, 6), 'roofline': roof,
                'whole_step': whole_step(prec, elt, ms)}
    if rank == 0:
        head = records[builds[0]]
        workload = ('SRGAN x2 generator (16 blocks, 64 features, spectral norm) fwd+bwd+Adam with bicubic LR degradation '
                    'and pixel-MSE x10, per-GPU batch 16 HR 192x192 patches (LR 96x96 -> SR 192x192): the north star\'s headline. '
                    'BASELINE.json\'s metric string "96^2 crops, x4" is configs.cfg3 of this line. ')
        if len(builds) == 3:
            workload += ('Top level = fp32 parity build (exact-fp32 MFMA, the reference\'s precision, the build that meets every '
                         '1e-3 golden vector); split_build = the same fp32 tensors with the trunk contractions on the bf16 matrix '
                         'instruction over hi / lo bf16 pairs of the fp32 operands (2^-17 operands: 3e-5 on the output of the 34-layer '
                         'generator, forward inside 1e-3 everywhere; NOT a 1e-3 parity build for gradients -- its extra forward noise '
                         'flips more activation masks: 1.5e-3 on one golden gradient, 2e-3 .. 2e-2 max-norm at full size, DESIGN.md section 3); '
                         'perf_build = bf16 tensors in HBM (the precision SURVEY 8d / BASELINE.json config 2 '
                         'sanction for the perf configs).  Same workload, same step count.')
        else:
            workload += 'Single build: %s.' % builds[0]
        rec = {
            'metric': 'HR patches/sec (x2 generator fwd+bwd, LR 96x96 -> SR 192x192)',
            'value': head['value'], 'unit': 'HR patches/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': head['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': head['dtype'], 'data': 'synthetic',
            'config': {'workload': workload, 'per_gpu_batch': B, 'hr': HR, 'lr': LR, 'parallelism': 'dp%d' % world,
                       'hip_graph': head['hip_graph'], 'final_loss': head['final_loss']},
            'roofline': head['roofline'], 'whole_step': head['whole_step'],
        }
        if len(builds) == 3:
            rec['split_build'] = records['bf16x3']
            rec['perf_build'] = records['bf16']
    if args.configs != 'none':
        # every rank takes part (the iterations all-reduce both networks' gradients when N > 1); rank 0 keeps the records
        default = list(CONFIGS) if world == 1 else ['cfg3', 'cfg4', 'cfg5']
        names = default if args.configs == 'all' else [c for c in args.configs.split(',') if c in CONFIGS]
        cfg_recs = {}
        for name in names:
            try:
                cfg_recs[name] = bench_config(name, device, args.config_iters, log, rank, world, not args.no_graph)
            except Exception as e:                              # noqa: BLE001  (the headline line must still be printed)
                if world > 1:
                    raise                                       # (a rank that drops out of a collective would hang the others)
                cfg_recs[name] = {'error': '%s: %s' % (type(e).__name__, str(e).splitlines()[0] if str(e) else '')}
        sub('engine').set_precision('fp32')
        if rank == 0:
            rec['configs'] = cfg_recs
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            rec['cpu_baseline'] = cpu_baseline()
        # the numbers a truncated tail of this (long) line must still show, last
        rec['summary'] = {'value': rec['value'], 'unit': rec['unit'], 'ms_per_step': rec['ms_per_step'], 'dtype': rec['dtype'], 'n_gpus': world,
                          'perf_build_value': rec.get('perf_build', {}).get('value'), 'perf_build_ms_per_step': rec.get('perf_build', {}).get('ms_per_step'),
                          'configs_ms_per_iteration': {k: v.get('ms_per_iteration') for k, v in rec.get('configs', {}).items()},
                          'roofline_frac': rec['roofline'].get('frac')}
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
