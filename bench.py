"""bench.py — training throughput of the LVAE hot path on MI355X (driver contract).

Workload (BASELINE.json metric): one ELBO training step = forward + backward + Adamax [+ gradient all-reduce] of the
CIFAR10-shaped 15-layer Ladder VAE (BASELINE configs[2]: fp32, batch 256 per GPU, DMoL likelihood), synthetic images,
default init under torch.manual_seed(42). Inputs are resident in HBM before the timed region. Weak scaling: every
rank processes its own 256-image shard; `value` = images of all ranks / max-over-ranks time.

Extra objects in the JSON line:
  roofline     — the dominant kernel (the convolution shape with the largest total time in an instrumented eager step, timed
                 with HIP events on the launch stream; not part of `value`): `frac` = matrix FLOPs the kernel ISSUES per launch (Winograd
                 F(2x2,3x3): 16/36 of the direct multiplies, each as six bf16-piece products) / average launch time / the dense peak of
                 the unit that issues them (bf16 MFMA); `effective_tflops` keeps the algorithmic direct-convolution rate, `step` holds the
                 whole-step fractions of both roofs and the PMC-measured traffic.
  bf16_shard   — the same step with compute_dtype = 'bf16' (BASELINE configs[3] per-GPU shard: bf16 matrix-core operands, bf16 storage
                 of the residual-block internals, fp32 accumulation / statistics / residual stream), with its HBM roofline.
  other_configs — BASELINE configs[1] (static-MNIST 12-layer, bf16, batch 256) and configs[4]'s per-GPU shard (64x64 20-layer, bf16,
                 batch 128) through the same TrainStep: 5 untimed + 10 timed steps each, ms/step, images/s and the step's HBM fraction.
  cpu_baseline — the CPU oracle (oracle/lvae_ref.py, a port of the reference) timed on this box's host cores on a bounded
                 sample (rank 0, N=1 only): BASELINE configs[0] at batch 64 and CIFAR-15 at batch 32, 2 warm-up + 5 timed steps.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import lvae_amd  # noqa: E402,F401
from lvae_amd.configs import CELEBA20, CIFAR15, MNIST3, MNIST12, synthetic_images  # noqa: E402  (the dicts the full-size parity tests build their models from)

PEAK_MFMA_F32 = 157.3   # TFLOP/s, MI355X_MICROARCH.md
PEAK_HBM = 8000.0       # GB/s, MI355X_MICROARCH.md (spec; 6.3 TB/s is what a streaming copy achieves)
F_ALG_PER_IMAGE = 10.563e9     # FLOP, fwd + dgrad + wgrad of every convolution
B_ALG_PER_IMAGE = 115.93e6     # bytes, fp32 activations in/out of every convolution, fwd + dgrad + wgrad
B_ALG_PER_IMAGE_BF16 = 57.96e6  # bytes, the same tensors stored in bf16 (BASELINE.md §4 counts 2 B per element)
PEAK_MFMA_BF16 = 2500.0        # TFLOP/s dense, MI355X_MICROARCH.md
# committed rocprofv3 summaries of this command (newest round first; tools/r05_final.sh writes them)
ROCPROF_SUMMARIES = ('profiles/r05_kernel_by_grid.txt', 'profiles/r04_kernel_by_grid.txt', 'profiles/r03_kernel_by_grid.txt', 'profiles/r02_kernel_by_grid.txt')
PMC_SUMMARIES = ('profiles/r05_pmc/hbm_traffic.json', 'profiles/r04_pmc/hbm_traffic.json', 'profiles/r03_pmc/hbm_traffic.json', 'profiles/r02_pmc/hbm_traffic.json', 'profiles/r01_pmc2/hbm_traffic.json')
PMC_SUMMARIES_BF16 = ('profiles/r05_pmc/hbm_traffic_bf16.json', 'profiles/r04_pmc/hbm_traffic_bf16.json', 'profiles/r03_pmc/hbm_traffic_bf16.json')
ROCPROF_SUMMARIES_BF16 = ('profiles/r05_bf16_kernel_by_grid.txt', 'profiles/r04_bf16_kernel_by_grid.txt', 'profiles/r03_bf16_kernel_by_grid.txt', 'profiles/r02_bf16_kernel_by_grid.txt')


def first_existing(paths):
    for rel in paths:
        if os.path.exists(os.path.join(ROOT, rel)):
            return rel
    return None


def host_cores():
    """Cores this process may actually use: the affinity mask, capped at the GPU box's per-GPU share (16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def log(msg):
    print('[bench] ' + msg, file=sys.stderr, flush=True)


def synth_batches(n, batch, seed):
    g = torch.Generator().manual_seed(seed)
    return [synthetic_images(CIFAR15, batch, g) for _ in range(n)]


def conv_roofline(model, x):
    """Instrumented eager step: HIP events (recorded on the launch stream) around every matrix-core conv launch
    (lvae_conv2d_f32 = forward and dgrad, lvae_conv1x1_gate_f32). Launches are grouped by shape; the group with the
    largest total time is the dominant kernel and gets the per-launch roofline record."""
    from lvae_amd import kernels as K
    from lvae_amd.engine import forward_pass
    rec = []
    variants = {}
    orig = K.call

    def timed_call(name, *args):
        if name not in ('lvae_conv2d_f32', 'lvae_conv1x1_gate_f32', 'lvae_resblock_conv_f32'):
            return orig(name, *args)
        d = args[0]._obj
        flops = 2.0 * d.N * d.OH * d.OW * d.Cout * (d.C1 + d.C2) * d.KH * d.KW
        if d.gather == 1 and d.stride > 1:
            flops /= d.stride * d.stride  # taps that hit no input pixel are not algorithmic work
        # algorithmic bytes of the launch: input + output tensors in their STORAGE type (bf16-stored residual-block tensors: 2 B), fp32 weights
        nbytes = ((2.0 if d.x_dtype else 4.0) * d.N * d.H * d.W * (d.C1 + d.C2) + (2.0 if d.y_dtype else 4.0) * d.N * d.OH * d.OW * d.Cout +
                  4.0 * d.KH * d.KW * (d.C1 + d.C2) * d.Cout)
        key = 'conv %dx%d s%d %d->%d @%dx%dx%d' % (d.KH, d.KW, d.stride, d.C1 + d.C2, d.Cout, d.N, d.OH, d.OW)
        if name == 'lvae_resblock_conv_f32':
            # fused residual-block launches (resblock_img.hip, or the Winograd kernel with the gate behind it): a group of their own per
            # prologue / epilogue; the 1x1 gate GEMM's work and tensors are counted with the launch that contains it
            e = args[1]._obj if args[1] is not None else None
            pro, epi = (e.prologue, e.epilogue) if e is not None else (0, 0)
            px = float(d.N) * d.OH * d.OW
            if epi == 1 or pro == 2:
                flops += 2.0 * px * 64 * 128
                nbytes += 4.0 * px * (128 + 64 + (64 if epi == 1 else 128))
            if pro == 1:
                nbytes += 4.0 * px * 64 * 2
            key = 'fused ' + ('gate-bwd + ' if pro == 2 else 'bn-apply + ' if pro == 1 else '') + key + (' + gate' if epi == 1 else '')
        if name == 'lvae_conv2d_f32':
            variants[key] = K._C.load().lvae_conv2d_variant(args[0])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(name, *args)
        e1.record()
        rec.append((key, flops, nbytes, e0, e1))

    K.call = timed_call
    tracker, model.grad_tracker = getattr(model, 'grad_tracker', None), None   # this instrumented step is local to rank 0: no gradient exchange
    try:
        model.zero_grad()
        K.prepared.prepare_all()  # as in the training step: the per-launch records below are the convolution kernels alone
        out = forward_pass(model, x)
        out['loss'].backward()
        torch.cuda.synchronize()
    finally:
        K.call = orig
        model.grad_tracker = tracker
    groups = {}
    for key, f, nb, e0, e1 in rec:
        g = groups.setdefault(key, [0, 0.0, 0.0, 0.0])
        g[0] += 1
        g[1] += f
        g[2] += nb
        g[3] += e0.elapsed_time(e1)
    dom = max(groups.items(), key=lambda kv: kv[1][3])
    fam_f = sum(g[1] for g in groups.values())
    fam_ms = sum(g[3] for g in groups.values())
    return dom, fam_f, fam_ms, len(rec), variants.get(dom[0], 0)


def pmc_traffic(dom_key, kname, wgs, paths=PMC_SUMMARIES):
    """HBM bytes per launch of the dominant kernel from a committed PMC run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    passes, (2*FETCH_SIZE + WRITE_SIZE)*1024 with the gfx950 half-count correction). Returns (bytes | None, source)."""
    for rel in paths:
        path = os.path.join(ROOT, rel)
        if dom_key != 'conv 3x3 s1 64->64 @256x16x16' or not os.path.exists(path):
            continue
        meta = json.load(open(path))
        prefix = kname if kname.endswith('<false>') else kname.split('<')[0]
        vals = [v['hbm_bytes_per_launch'] for k, v in meta['kernels'].items() if k.startswith(prefix) and k.endswith('@%d workgroups' % wgs)]
        if not vals and prefix != kname.split('<')[0]:   # summaries of earlier rounds: the kernel was not a template yet
            prefix = kname.split('<')[0]
            vals = [v['hbm_bytes_per_launch'] for k, v in meta['kernels'].items() if k.startswith(prefix) and k.endswith('@%d workgroups' % wgs)]
        if vals:
            return sum(vals) / len(vals), '%s (%s)' % (rel, meta.get('scope', 'single-layer micro-benchmark tools/conv_bench.py, not the whole step'))
    return None, None


def pmc_step(paths=PMC_SUMMARIES):
    """(HBM bytes of one whole training step measured by the committed PMC run, its file) or (None, None)."""
    rel = first_existing(paths)
    if rel is None:
        return None, None
    st = json.load(open(os.path.join(ROOT, rel))).get('step')
    return (st['hbm_bytes_per_step'], rel) if st else (None, None)


def rocprof_avg_us(kernel_prefix, wgs, paths=None):
    """Average duration of `kernel_prefix` at `wgs` workgroups from the committed rocprofv3 --kernel-trace summary (call-weighted over the
    template instances that match)."""
    rel = first_existing(paths or ROCPROF_SUMMARIES)
    if rel is None:
        return None
    calls = tot = 0.0
    for line in open(os.path.join(ROOT, rel)):
        if line.startswith(kernel_prefix) and ('wgs=%6d' % wgs) in line and 'avg=' in line:
            n = float(line.split('calls=')[1].split()[0])
            calls += n
            tot += n * float(line.split('avg=')[1].split()[0])
    return tot / calls if calls else None


def oracle_images_per_s(cfg, batch, warm, steps):
    from oracle import lvae_ref as R
    from lvae_amd.models.lvae import LadderVAE
    torch.manual_seed(42)
    sd = {k: v.clone() for k, v in LadderVAE(**cfg).state_dict().items()}
    pkeys = [k for k in sd if R.is_parameter_key(k)]
    for k in pkeys:
        sd[k].requires_grad_(k != 'top_down_layers.%d.top_prior_params' % (len(cfg['z_dims']) - 1) or cfg['learn_top_prior'])
    tk = [k for k in pkeys if sd[k].requires_grad]
    m = [torch.zeros_like(sd[k]) for k in tk]
    u = [torch.zeros_like(sd[k]) for k in tk]
    g = torch.Generator().manual_seed(99)
    xs = [synthetic_images(cfg, batch, g) for _ in range(2)]
    gen = torch.Generator().manual_seed(1)
    times = []
    for i in range(warm + steps):
        t0 = time.time()
        for k in tk:
            sd[k].grad = None
        fp, _ = R.forward_pass(sd, cfg, xs[i % 2], R.Tape(gen=gen), param_keys=pkeys)
        fp['loss'].backward()
        with torch.no_grad():
            R.adamax_step([sd[k] for k in tk], [sd[k].grad for k in tk], m, u, i + 1)
        times.append(time.time() - t0)
    t = sum(times[warm:]) / steps
    return batch / t, t


def cpu_baseline():
    """The oracle (CPU port of the reference path) on this box's host cores: forward + backward + Adamax."""
    cores = host_cores()
    torch.set_num_threads(cores)
    v3, t3 = oracle_images_per_s(CIFAR15, 32, 2, 5)
    v1, t1 = oracle_images_per_s(MNIST3, 64, 2, 5)
    return {'value': v3, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': 'CIFAR-15 fp32 batch 32, fwd+bwd+Adamax, 2 warm-up + 5 timed steps (%.2f s/step); BASELINE configs[0] '
                      '(static-MNIST 3-layer, batch 64) on the same cores: %.1f images/s (%.2f s/step)' % (t3, v1, t1),
            'cfg1_mnist3_b64_images_per_s': v1}


# BASELINE configs[1] and configs[4] (per-GPU shard): (key, constructor dict, images per GPU, algorithmic FLOPs and bf16 bytes per image
# of SURVEY.md §8d / BASELINE.md §4)
OTHER_CONFIGS = (
    ('cfg2_mnist12_bf16_b256', 'BASELINE configs[1]: static-MNIST 12-layer LVAE, Bernoulli likelihood, bf16, batch 256', MNIST12, 256, 8.337e9, 46.73e6),
    ('cfg5_celeba20_bf16_b128', 'BASELINE configs[4] per-GPU shard: 64x64 20-layer LVAE, DMoL-10, bf16, batch 128 (1024 over 8 GPUs)', CELEBA20, 128, 42.472e9, 233.10e6),
)


def time_other_config(desc, cfg, batch, f_alg_img, b_alg_img, dev, steps=10, warmup=5):
    """One of the other BASELINE configurations through the same TrainStep (hipGraph) as the headline: `warmup` untimed + `steps` timed
    steps on a ring of 4 resident synthetic batches, compute_dtype bf16. The model and its graph are dropped afterwards."""
    import gc
    from lvae_amd import kernels as K
    from lvae_amd.engine import TrainStep
    from lvae_amd.models.lvae import LadderVAE
    from lvae_amd.noise import PhiloxNoise
    from lvae_amd.optim import Adamax
    torch.manual_seed(42)
    model = LadderVAE(**cfg).to(dev)
    model.train()
    model.compute_dtype = 'bf16'
    model.noise = PhiloxNoise(seed=42, rank=0)
    model.pack()
    step = TrainStep(model, Adamax(model, lr=3e-4), use_graph=True)
    gen = torch.Generator().manual_seed(1234)
    ring = [synthetic_images(cfg, batch, gen).to(dev) for _ in range(4)]
    for i in range(max(warmup, 3)):
        step(ring[i % 4])
    dt, out = time_steps(step, ring, steps, 1, dev)
    s = dt / steps
    rec = {'config': desc, 'value': batch / s, 'unit': 'images/s', 'ms_per_step': s * 1e3, 'steps': steps, 'warmup': warmup, 'dtype': 'bf16',
           'batch_per_gpu': batch, 'hip_graph': step.use_graph, 'neg_elbo': -float(out['elbo']),
           'step': {'algorithmic_gb_per_s': b_alg_img * batch / s / 1e9, 'hbm_frac': b_alg_img * batch / s / 1e9 / PEAK_HBM,
                    'hbm_floor_ms': b_alg_img * batch / (PEAK_HBM * 1e9) * 1e3, 'algorithmic_tflops': f_alg_img * batch / s / 1e12,
                    'bf16_mfma_frac': f_alg_img * batch / s / 1e12 / PEAK_MFMA_BF16,
                    'note': 'algorithmic bytes = BASELINE.md §4 (2 B per activation element, fwd + dgrad + wgrad of every convolution)'}}
    del step, model, ring, out
    gc.collect()
    K.prepared.evict_dead()
    K.set_precision('f32')
    torch.cuda.empty_cache()
    return rec


def time_steps(step, ring, n, world, dev):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        out = step(ring[i % len(ring)])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt, out


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks as CHILD processes through torch.distributed.run
    — before this process has made any GPU call, and never by exec — relay rank 0's JSON line and exit non-zero if any child does."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC only on this platform (RCCL across processes)
    env['LVAE_BENCH_SELF_LAUNCHED'] = '1'
    env.setdefault('OMP_NUM_THREADS', str(max(1, host_cores() // max(1, args.gpus))))
    log('no WORLD_SIZE in the environment: starting %d ranks as child processes (%s)' % (args.gpus, ' '.join(cmd[1:8])))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        out = out.rstrip('\n')
        if out.startswith('{') and '"metric"' in out:
            line = out
        elif out:
            print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc != 0 or line is None:
        raise SystemExit(rc if rc != 0 else 4)
    raise SystemExit(0)


def dominant_kernel_record(dkey, dn, dflops, dbytes, dms, kind, dtype):
    """Roofline record of the dominant kernel. `frac` = matrix FLOPs the kernel ISSUES per launch / average launch time / the dense
    peak of the unit that issues them; the algorithmic direct-convolution rate is kept as `effective_tflops`."""
    from lvae_amd import kernels as K
    t = dms * 1e-3 / dn                       # seconds per launch (HIP events on the launch stream)
    f_alg, b_alg = dflops / dn, dbytes / dn
    V = K._C
    if kind == V.VARIANT_WINO_SIX:   # Winograd F(2x2,3x3): 16/36 of the direct multiplies, each as six bf16-piece products on the bf16 MFMA
        issued, peak, unit = f_alg * 16.0 / 36.0 * 6.0, PEAK_MFMA_BF16, 'bf16 MFMA (v_mfma_f32_32x32x16_bf16), six exact bf16-piece products per fp32 product'
        kname = 'conv3x3_wino2_kernel<false>'
    elif kind == V.VARIANT_WINO_F32:
        issued, peak, unit, kname = f_alg * 16.0 / 36.0, PEAK_MFMA_F32, 'fp32 MFMA (v_mfma_f32_32x32x2_f32)', 'conv3x3_wino_kernel<64, 2, 1, false>'
    elif kind == V.VARIANT_BF16_DIRECT:
        issued, peak, unit, kname = f_alg, PEAK_MFMA_BF16, 'bf16 MFMA (v_mfma_f32_32x32x16_bf16), bf16 operands', 'conv3x3_bf16_kernel<1, *>'
    elif kind == V.VARIANT_SIX_DIRECT:
        issued, peak, unit, kname = 6 * f_alg, PEAK_MFMA_BF16, 'bf16 MFMA, six-product direct form', 'conv3x3_bf16_kernel<3, *>'
    else:
        issued, peak, unit, kname = f_alg, PEAK_MFMA_F32, 'fp32 MFMA (v_mfma_f32_32x32x2_f32)', 'conv3x3_pos_kernel / conv3x3_halo_kernel / conv1x1_kernel / conv_igemm_kernel'
    ach = issued / t / 1e12
    hbm = b_alg / t / 1e9
    # workgroups of the dominant shape (256x16x16): 256-pixel tiles for the 8-wave Winograd kernel, 128-pixel tiles otherwise
    wgs = 256 if kind == V.VARIANT_WINO_SIX else 512
    traffic, traffic_src = pmc_traffic(dkey, kname, wgs, PMC_SUMMARIES if dtype == 'f32' else PMC_SUMMARIES_BF16)
    rec = {
        'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
        'traffic': traffic, 'traffic_source': traffic_src,
        'kernel': '%s (forward + dgrad launches of: %s)' % (kname, dkey),
        'mfma_unit': unit, 'issued_flops_per_launch': issued, 'algorithmic_flops_per_launch': f_alg,
        'effective_tflops': f_alg / t / 1e12,
        'effective_frac_of_fp32_mfma_peak': f_alg / t / 1e12 / PEAK_MFMA_F32,
        'algorithmic_bytes_per_launch': b_alg, 'hbm_gb_per_s': hbm, 'hbm_frac_of_peak': hbm / PEAK_HBM,
        'launches_per_step': dn, 'avg_launch_us': t * 1e6,
        'avg_launch_us_rocprof': (rocprof_avg_us(kname, wgs, ROCPROF_SUMMARIES if dtype == 'f32' else ROCPROF_SUMMARIES_BF16) if kname.endswith('<false>') else None) or
                                 rocprof_avg_us(kname.split('<')[0], wgs, ROCPROF_SUMMARIES if dtype == 'f32' else ROCPROF_SUMMARIES_BF16),
        'rocprof_summary': first_existing(ROCPROF_SUMMARIES if dtype == 'f32' else ROCPROF_SUMMARIES_BF16),
        'note': 'frac = FLOPs the matrix unit ISSUES per launch / avg launch time / that unit\'s dense peak. effective_tflops counts the '
                'ALGORITHMIC direct-convolution FLOPs (2*N*OH*OW*Cout*Cin*9) the launch replaces. The kernel is bound by neither pipe peak: '
                'see the SQ counters named in DESIGN.md §5 (vector instructions per MFMA, parked waves).',
    }
    if kind == V.VARIANT_BF16_DIRECT:
        rec['bound'] = 'hbm' if hbm / PEAK_HBM > ach / peak else 'mfma'
        if rec['bound'] == 'hbm':
            rec.update({'achieved': hbm, 'peak': PEAK_HBM, 'unit': 'GB/s', 'frac': hbm / PEAK_HBM, 'mfma_frac': ach / peak})
    return rec


def step_record(step_s, batch, dtype):
    """Whole-step fractions: algorithmic conv FLOPs / bytes over the step time, and — from the committed PMC run —
    the HBM bytes the step really moves."""
    b_alg = (B_ALG_PER_IMAGE_BF16 if dtype == 'bf16' else B_ALG_PER_IMAGE) * batch
    f_alg = F_ALG_PER_IMAGE * batch
    rec = {'algorithmic_tflops': f_alg / step_s / 1e12, 'algorithmic_gb_per_s': b_alg / step_s / 1e9,
           'hbm_frac': b_alg / step_s / 1e9 / PEAK_HBM,
           'hbm_floor_ms': b_alg / (PEAK_HBM * 1e9) * 1e3, 'target_ms_at_half_hbm_roof': 2 * b_alg / (PEAK_HBM * 1e9) * 1e3,
           'mfma_f32_frac': f_alg / step_s / 1e12 / PEAK_MFMA_F32,
           'compute_floor_ms': {'fp32_mfma_direct': f_alg / (PEAK_MFMA_F32 * 1e12) * 1e3,
                                'bf16_mfma_six_product_direct': 6 * f_alg / (PEAK_MFMA_BF16 * 1e12) * 1e3,
                                'bf16_mfma_six_product_winograd': 6 * f_alg * 16 / 36 / (PEAK_MFMA_BF16 * 1e12) * 1e3,
                                'bf16_mfma_bf16_operands': f_alg / (PEAK_MFMA_BF16 * 1e12) * 1e3}}
    measured, src = pmc_step(PMC_SUMMARIES_BF16 if dtype == 'bf16' else PMC_SUMMARIES)
    if measured is not None:
        rec.update({'measured_hbm_bytes': measured, 'measured_hbm_frac': measured / step_s / 1e9 / PEAK_HBM,
                    'traffic_ratio': measured / b_alg, 'measured_hbm_source': src})
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256, help='images per GPU')
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='f32', help='f32 = BASELINE configs[2] (the headline); bf16 = configs[3] per-GPU shard')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend ('nccl' = RCCL; 'gloo' to rehearse N ranks on one GPU)")
    ap.add_argument('--async-wgrad', action='store_true',
                    help='weight-gradient kernels on side streams (measured slower than one stream since the Winograd kernels: 57.0 vs 55.1 ms)')
    ap.add_argument('--wgrad-group-rows', type=int, default=int(os.environ.get('LVAE_WGRAD_GROUP_ROWS', '16384')), help='weight gradients of layers with at most this many pixels (N*H*W) are launched in groups (0 = off)')
    ap.add_argument('--wgrad-streams', type=int, default=1, help='side streams the weight-gradient kernels are spread over')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-bf16-line', action='store_true', help='skip the nested bf16_shard measurement of the default fp32 run')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the nested measurements of BASELINE configs[1] and configs[4] (per-GPU shard)')
    ap.add_argument('--launch-check', action='store_true',
                    help='rendezvous rehearsal without a GPU: every rank joins the process group (use --backend gloo), all-reduces one host '
                         'scalar and rank 0 prints a JSON line; exercises the self-launch path of --gpus N (tests/test_dist_cpu.py)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and os.environ.get('LVAE_FORCE_DIST') != '1':
        self_launch(args, sys.argv[1:])   # never returns

    # Native libraries write to the process's stdout too (RCCL prints a five-line version banner there when its first communicator
    # is created): keep file descriptor 1 pointed at stderr for the whole run and give it back only for the ONE JSON line at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from lvae_amd import dist as ldist
    from lvae_amd.models.lvae import LadderVAE
    from lvae_amd.noise import PhiloxNoise
    from lvae_amd.optim import Adamax
    from lvae_amd.engine import TrainStep

    rank, world, local = ldist.init_from_env(args.backend)
    if args.launch_check:
        t = torch.tensor([float(rank + 1)])
        if world > 1:
            dist.all_reduce(t)
        if rank == 0:
            os.write(json_fd, (json.dumps({'metric': 'launch-check', 'value': float(t.item()), 'n_gpus': world, 'backend': args.backend,
                                           'self_launched': os.environ.get('LVAE_BENCH_SELF_LAUNCHED') == '1'}) + '\n').encode())
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        if os.environ.get('LVAE_LAUNCH_CHECK_FAIL_RANK') == str(rank):
            raise SystemExit(7)   # rehearses "one child fails": the parent must exit non-zero
        return
    if world > 1:
        # A multi-rank run that stops making progress (a collective one rank never enters) must not sit on the node until the caller's
        # limit: after LVAE_BENCH_WATCHDOG_S seconds (default 900) the rank says where it was and exits non-zero, which ends the job.
        import threading
        limit = float(os.environ.get('LVAE_BENCH_WATCHDOG_S', '900'))

        def _expired():
            log('rank %d: no result after %.0f s (LVAE_BENCH_WATCHDOG_S); LVAE_DDP_MODE=split keeps the gradient exchange outside '
                'the step graph if the captured exchange is what hangs' % (rank, limit))
            os._exit(3)

        watchdog = threading.Timer(limit, _expired)
        watchdog.daemon = True
        watchdog.start()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" %
                         (args.gpus, world, args.gpus))
    local = local % max(1, torch.cuda.device_count())  # rehearsal: several ranks may share one device (gloo only)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    torch.manual_seed(42)  # identical default init on every rank (README's seed)
    model = LadderVAE(**CIFAR15).to(dev)
    model.train()
    model.compute_dtype = args.dtype
    model.noise = PhiloxNoise(seed=42, rank=rank)
    arena = model.pack()
    ldist.broadcast_flat(arena.params)
    opt = Adamax(model, lr=3e-4)
    multi = world > 1 or os.environ.get('LVAE_FORCE_DIST') == '1'
    allreduce = None
    if world > max(1, torch.cuda.device_count()) and not args.no_graph and os.environ.get('LVAE_ALLOW_GLOO_GRAPH') != '1':
        # rehearsal with several ranks on ONE device: two processes replaying multi-thousand-node graphs on one GPU time-slice
        # (2.6x per rank measured); launch eagerly instead
        log('ranks share a device: hipGraph replay disabled')
        args.no_graph = True

    def make_step():
        kw = dict(use_graph=not args.no_graph, async_wgrad=args.async_wgrad, wgrad_streams=args.wgrad_streams,
                  wgrad_group_rows=args.wgrad_group_rows or None)
        if multi:
            # more than one rank: both forms of the gradient exchange are built and the faster one ON THIS WORLD SIZE is kept (a few trial
            # steps of each, all of them real training steps; LVAE_DDP_MODE=split|overlap skips the trial) — engine.AutoExchangeStep
            from lvae_amd.engine import AutoExchangeStep
            return AutoExchangeStep(model, opt, arena.grads, arena.segments, trial_steps=3, **kw)
        return TrainStep(model, opt, **kw)

    step = make_step()
    if multi:
        allreduce = step.allreduce
    torch.set_num_threads(host_cores())
    ring = [b.to(dev) for b in synth_batches(8, args.batch, 1234 + rank)]
    if rank == 0:
        log('model built (%d params), warming up' % sum(p.numel() for p in model.parameters()))
    for i in range(max(args.warmup, 3)):  # >= 3: two eager steps + the capture replay
        step(ring[i % 8])
    extra = 0
    while multi and not step.ready:   # the exchange-form trial (untimed, like the warm-up): 2 x (2 eager + capture + 3 timed) steps at most
        step(ring[extra % 8])
        extra += 1
        if extra > 64:
            raise SystemExit('the gradient-exchange trial did not finish')
    if multi:
        allreduce = step.allreduce
    dt, out = time_steps(step, ring, args.steps, world, dev)
    loss, elbo = float(out['loss']), float(out['elbo'])
    if rank == 0:
        log('timed %d steps: %.2f ms/step' % (args.steps, dt / args.steps * 1e3))

    line = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        line = {
            'metric': 'training images/sec (ELBO step: fwd+bwd+Adamax' + ('+grad all-reduce' if world > 1 else '') + ')',
            'value': args.batch * world * args.steps / dt, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'CIFAR10-shaped 15-layer LVAE (BASELINE configs[%d]): 32x32x3, zdims 32x15, 4 blocks/layer, '
                                   '64 filters, gated+skip, DMoL-10, dropout 0.2, free bits 1.0' % (2 if args.dtype == 'f32' else 3),
                       'batch_per_gpu': args.batch, 'global_batch': args.batch * world,
                       'parallelism': 'dp%d' % world, 'hip_graph': step.use_graph},
            'neg_elbo': -elbo, 'loss': loss,
        }
        if allreduce is not None and allreduce.active:
            line['config']['grad_exchange'] = step.exchange_description()
            line['config']['grad_exchange_ab'] = {'chosen': getattr(step, 'chosen', None), 'ms_per_step_max_over_ranks': getattr(step, 'timings_ms', None),
                                                  'trial_steps_untimed_in_this_line': extra}
            line['config']['allreduce_bytes_per_step'] = 4 * arena.grads.numel()
            line['config']['allreduce_buckets'] = len(allreduce.buckets)
            line['config']['allreduce_bucket_bytes'] = [4 * (hi - lo) for lo, hi, _ in allreduce.buckets]
            line['config']['backend'] = dist.get_backend() if dist.is_initialized() else None
    step_s = dt / args.steps
    if rank == 0 and not args.no_roofline:
        (dkey, (dn, dflops, dbytes, dms)), fam_f, fam_ms, n, kind = conv_roofline(model, ring[0])
        line['roofline'] = dominant_kernel_record(dkey, dn, dflops, dbytes, dms, kind, args.dtype)
        line['roofline']['all_conv_fwd_dgrad'] = {'launches_per_step': n, 'algorithmic_flops_per_step': fam_f, 'ms_per_step': fam_ms,
                                                  'effective_tflops': fam_f / (fam_ms * 1e-3) / 1e12}
        line['roofline']['step'] = step_record(step_s, args.batch, args.dtype)
    if rank == 0 and world == 1 and args.dtype == 'f32' and not args.no_bf16_line:
        # BASELINE configs[3] per-GPU shard: same model, same batch, compute_dtype bf16
        from lvae_amd import kernels as K
        model.compute_dtype = 'bf16'
        step16 = make_step()
        for i in range(5):
            step16(ring[i % 8])
        n16 = max(5, args.steps // 2)
        dt16, out16 = time_steps(step16, ring, n16, 1, dev)
        s16 = dt16 / n16
        (k16, (n_l, f16, b16, ms16)), _, _, _, kind16 = conv_roofline(model, ring[0])
        model.compute_dtype = 'f32'
        K.set_precision('f32')
        r16 = dominant_kernel_record(k16, n_l, f16, b16, ms16, kind16, 'bf16')
        r16['step'] = step_record(s16, args.batch, 'bf16')
        r16['step_hbm_frac'] = r16['step']['hbm_frac']
        line['bf16_shard'] = {
            'config': 'BASELINE configs[3] per-GPU shard: CIFAR10 15-layer, batch %d, compute_dtype bf16 (%s)' % (args.batch, K.BF16_MODE_NOTE),
            'value': args.batch / s16, 'unit': 'images/s', 'ms_per_step': s16 * 1e3, 'steps': n16, 'dtype': 'bf16',
            'neg_elbo': -float(out16['elbo']), 'roofline': r16}
        log('bf16 shard: %.2f ms/step' % (s16 * 1e3))
    if rank == 0 and world == 1 and args.dtype == 'f32' and not args.no_other_configs:
        line['other_configs'] = {}
        for key, desc, cfg, batch, f_img, b_img in OTHER_CONFIGS:
            line['other_configs'][key] = time_other_config(desc, cfg, batch, f_img, b_img, dev)
            log('%s: %.2f ms/step' % (key, line['other_configs'][key]['ms_per_step']))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log('cpu baseline on %d host cores ...' % host_cores())
        line['cpu_baseline'] = cpu_baseline()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + '\n').encode())
    if multi:
        torch.cuda.synchronize()
        step.close()   # lvae_allreduce_destroy: the private RCCL communicator goes before the process group does
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
