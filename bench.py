"""bench.py — training throughput of the LVAE hot path on MI355X (driver contract).

Workload (BASELINE.json metric): one ELBO training step = forward + backward + Adamax [+ gradient all-reduce] of the
CIFAR10-shaped 15-layer Ladder VAE (BASELINE configs[2]: fp32, batch 256 per GPU, DMoL likelihood), synthetic images,
default init under torch.manual_seed(42). Inputs are resident in HBM before the timed region. Weak scaling: every
rank processes its own 256-image shard; `value` = images of all ranks / max-over-ranks time.

Extra objects in the JSON line:
  roofline     — fp32-MFMA roofline of the dominant kernel (the forward + dgrad launches of the most expensive convolution
                 shape), from an instrumented eager step timed with HIP events on the launch stream (not part of `value`);
                 `achieved` counts the ALGORITHMIC (direct-convolution) FLOPs, `mfma_util` the MFMA FLOPs the kernel really
                 issues (Winograd F(2x2,3x3) issues 16/36 of them);
  cpu_baseline — the CPU oracle (oracle/lvae_ref.py, a port of the reference) timed on this box's host cores on a
                 bounded sample (rank 0, N=1 only).
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
from lvae_amd.configs import CIFAR15, MNIST3  # noqa: E402  (the same dicts the full-size parity tests build their models from)

PEAK_MFMA_F32 = 157.3  # TFLOP/s, MI355X_MICROARCH.md


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
    return [torch.floor(256 * torch.rand(batch, 3, 32, 32, generator=g)) / 255 for _ in range(n)]


def conv_roofline(model, x):
    """Instrumented eager step: HIP events (recorded on the launch stream) around every fp32-MFMA conv launch
    (lvae_conv2d_f32 = forward and dgrad, lvae_conv1x1_gate_f32). Launches are grouped by shape; the group with the
    largest total time is the dominant kernel and gets the per-launch roofline record."""
    from lvae_amd import kernels as K
    from lvae_amd.engine import forward_pass
    rec = []
    wino = set()
    orig = K.call

    def timed_call(name, *args):
        if name not in ('lvae_conv2d_f32', 'lvae_conv1x1_gate_f32'):
            return orig(name, *args)
        d = args[0]._obj
        flops = 2.0 * d.N * d.OH * d.OW * d.Cout * (d.C1 + d.C2) * d.KH * d.KW
        if d.gather == 1 and d.stride > 1:
            flops /= d.stride * d.stride  # taps that hit no input pixel are not algorithmic work
        nbytes = 4.0 * (d.N * d.H * d.W * (d.C1 + d.C2) + d.N * d.OH * d.OW * d.Cout + d.KH * d.KW * (d.C1 + d.C2) * d.Cout)
        key = 'conv %dx%d s%d %d->%d @%dx%dx%d' % (d.KH, d.KW, d.stride, d.C1 + d.C2, d.Cout, d.N, d.OH, d.OW)
        if name == 'lvae_conv2d_f32' and K._C.load().lvae_conv2d_workspace(args[0]) > 0:
            wino.add(key)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(name, *args)
        e1.record()
        rec.append((key, flops, nbytes, e0, e1))

    K.call = timed_call
    try:
        model.zero_grad()
        K.prepared.prepare_all()  # as in the training step: the per-launch records below are the convolution kernels alone
        out = forward_pass(model, x)
        out['loss'].backward()
        torch.cuda.synchronize()
    finally:
        K.call = orig
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
    return dom, fam_f, fam_ms, len(rec), dom[0] in wino


def pmc_traffic(dom_key, is_wino):
    """HBM bytes per launch of the dominant kernel from the committed PMC run (profiles/r01_pmc2/hbm_traffic.json);
    only the shape that was actually profiled (3x3 64->64 @256x16x16 on 512 workgroups)."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc2', 'hbm_traffic.json')
    if dom_key != 'conv 3x3 s1 64->64 @256x16x16' or not os.path.exists(path):
        return None
    ks = json.load(open(path))['kernels']
    prefix = 'conv3x3_wino_kernel' if is_wino else 'conv3x3_halo_kernel'
    vals = [v['hbm_bytes_per_launch'] for k, v in ks.items() if k.startswith(prefix) and k.endswith('@512 workgroups')]
    return sum(vals) / len(vals) if vals else None


def cpu_baseline(cfg, batch, steps):
    """The oracle (CPU port of the reference path) on this box's host cores: forward + backward + Adamax."""
    from oracle import lvae_ref as R
    from lvae_amd.models.lvae import LadderVAE
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(42)
    sd = {k: v.clone() for k, v in LadderVAE(**cfg).state_dict().items()}
    pkeys = [k for k in sd if R.is_parameter_key(k)]
    for k in pkeys:
        sd[k].requires_grad_(k != 'top_down_layers.%d.top_prior_params' % (len(cfg['z_dims']) - 1) or cfg['learn_top_prior'])
    tk = [k for k in pkeys if sd[k].requires_grad]
    m = [torch.zeros_like(sd[k]) for k in tk]
    u = [torch.zeros_like(sd[k]) for k in tk]
    xs = synth_batches(2, batch, 99)
    gen = torch.Generator().manual_seed(1)
    times = []
    for i in range(steps + 1):
        t0 = time.time()
        for k in tk:
            sd[k].grad = None
        fp, _ = R.forward_pass(sd, cfg, xs[i % 2], R.Tape(gen=gen), param_keys=pkeys)
        fp['loss'].backward()
        with torch.no_grad():
            R.adamax_step([sd[k] for k in tk], [sd[k].grad for k in tk], m, u, i + 1)
        times.append(time.time() - t0)
    t = sum(times[1:]) / steps
    return {'value': batch / t, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': 'CIFAR-15 fp32 batch %d, fwd+bwd+Adamax, 1 warm-up + %d timed steps (%.1f s/step)' % (batch, steps, t)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256, help='images per GPU')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend ('nccl' = RCCL; 'gloo' to rehearse N ranks on one GPU)")
    ap.add_argument('--async-wgrad', action='store_true',
                    help='weight-gradient kernels on side streams (measured slower than one stream since the Winograd kernels: 57.0 vs 55.1 ms)')
    ap.add_argument('--wgrad-group-rows', type=int, default=16384, help='weight gradients of layers with at most this many pixels (N*H*W) are launched in groups (0 = off)')
    ap.add_argument('--wgrad-streams', type=int, default=1, help='side streams the weight-gradient kernels are spread over')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    args = ap.parse_args()

    import lvae_amd  # noqa: F401
    from lvae_amd import dist as ldist
    from lvae_amd.models.lvae import LadderVAE
    from lvae_amd.noise import PhiloxNoise
    from lvae_amd.optim import Adamax
    from lvae_amd.engine import TrainStep

    rank, world, local = ldist.init_from_env(args.backend)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" %
                         (args.gpus, world, args.gpus))
    local = local % max(1, torch.cuda.device_count())  # rehearsal: several ranks may share one device (gloo only)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    torch.manual_seed(42)  # identical default init on every rank (README's seed)
    model = LadderVAE(**CIFAR15).to(dev)
    model.train()
    model.noise = PhiloxNoise(seed=42, rank=rank)
    arena = model.pack()
    ldist.broadcast_flat(arena.params)
    opt = Adamax(model, lr=3e-4)
    allreduce = (ldist.GradAllReduce(arena.grads, segments=arena.segments)
                 if (world > 1 or os.environ.get('LVAE_FORCE_DIST') == '1') else None)
    if world > max(1, torch.cuda.device_count()) and not args.no_graph and os.environ.get('LVAE_ALLOW_GLOO_GRAPH') != '1':
        # rehearsal with several ranks on ONE device: two processes replaying multi-thousand-node graphs on one GPU time-slice
        # through compute-wave save/restore (seconds per step, gpurun_out/ddp2g.log of round 1); launch eagerly instead
        log('ranks share a device: hipGraph replay disabled')
        args.no_graph = True
    step = TrainStep(model, opt, use_graph=not args.no_graph, allreduce=allreduce, async_wgrad=args.async_wgrad,
                     wgrad_streams=args.wgrad_streams, wgrad_group_rows=args.wgrad_group_rows or None)

    torch.set_num_threads(host_cores())
    ring = [b.to(dev) for b in synth_batches(8, args.batch, 1234 + rank)]
    if rank == 0:
        log('model built (%d params), warming up' % sum(p.numel() for p in model.parameters()))
    for i in range(max(args.warmup, 3)):  # >= 3: two eager steps + the capture replay
        step(ring[i % 8])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(ring[i % 8])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
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
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'CIFAR10-shaped 15-layer LVAE (BASELINE configs[2]): 32x32x3, zdims 32x15, 4 blocks/layer, '
                                   '64 filters, gated+skip, DMoL-10, dropout 0.2, free bits 1.0',
                       'batch_per_gpu': args.batch, 'global_batch': args.batch * world,
                       'parallelism': 'dp%d' % world, 'hip_graph': not args.no_graph},
            'neg_elbo': -elbo, 'loss': loss,
        }
    if rank == 0 and not args.no_roofline:
        (dkey, (dn, dflops, dbytes, dms)), fam_f, fam_ms, n, is_wino = conv_roofline(model, ring[0])
        ach = dflops / (dms * 1e-3) / 1e12
        issued = 16.0 / 36.0 if is_wino else 1.0
        line['roofline'] = {
            'bound': 'mfma', 'achieved': ach, 'peak': PEAK_MFMA_F32, 'unit': 'TFLOP/s', 'frac': ach / PEAK_MFMA_F32,
            'traffic': pmc_traffic(dkey, is_wino),
            'kernel': '%s (forward + dgrad launches of: %s)' % ('conv3x3_wino_kernel' if is_wino else 'conv3x3_halo_kernel', dkey),
            'flops_counted': 'algorithmic: direct convolution, 2*N*OH*OW*Cout*Cin*KH*KW per launch',
            'mfma_flops_issued_fraction': issued, 'mfma_util': ach * issued / PEAK_MFMA_F32,
            'launches_per_step': dn, 'avg_launch_us': dms * 1e3 / dn, 'flops_per_launch': dflops / dn,
            'algorithmic_bytes_per_launch': dbytes / dn,
            'all_conv_fwd_dgrad': {'launches_per_step': n, 'flops_per_step': fam_f, 'ms_per_step': fam_ms,
                                   'achieved': fam_f / (fam_ms * 1e-3) / 1e12, 'frac': fam_f / (fam_ms * 1e-3) / 1e12 / PEAK_MFMA_F32},
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log('cpu baseline on %d host cores ...' % host_cores())
        line['cpu_baseline'] = cpu_baseline(CIFAR15, 32, 3)
    if rank == 0:
        print(json.dumps(line))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
