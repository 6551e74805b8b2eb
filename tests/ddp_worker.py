"""Worker process of the data-parallel GPU tests (tests/test_ddp_gpu.py starts it; not collected by pytest).

  single  one rank, no process group: TrainStep (hipGraph), the reference point
  rccl1_split  one rank over RCCL with LVAE_FORCE_DIST=1, the DEFAULT form of a multi-rank GPU run: one message after backward, outside the
          graphs (fwd+bwd graph | exchange on the side stream | Adamax graph)
  rccl1   the same with LVAE_DDP_MODE=overlap: completion-ordered buckets exchanged on the side stream while backward runs, all inside the
          captured graph
  rccl1_fallback  as rccl1, but the all-reduce refuses to be captured: TrainStep must fall back to the split form in the same process
  gloo2   rank RANK of 2 over gloo, both ranks on the one GPU of the test box, eager launches (gloo cannot be captured)
  gloo2a  the same with every weight-gradient kernel on one of two side streams (async_wgrad): a bucket's exchange must wait for them
  emul2   one process playing both ranks of gloo2 one after the other: per-shard forward/backward with per-rank BatchNorm
          statistics and noise, gradients summed, Adamax with the 1/2 scale — what gloo2 must reproduce
usage: python tests/ddp_worker.py MODE OUT.pt STEPS
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CFG = dict(color_ch=3, z_dims=[8, 8, 8], blocks_per_layer=2, downsample=[0, 1, 1], nonlin='elu', merge_type='residual',
           batchnorm=True, stochastic_skip=True, n_filters=64, dropout=0.2, free_bits=1.0, learn_top_prior=True,
           img_shape=(32, 32), likelihood_form='discr_log_mix', res_block_type='bacdbacd', gated=True,
           no_initial_downscaling=False, analytical_kl=False)
PER_RANK = 64


def batches(steps, world):
    g = torch.Generator().manual_seed(77)
    return [torch.floor(256 * torch.rand(PER_RANK * world, 3, 32, 32, generator=g)) / 255 for _ in range(steps)]


def build(rank):
    import lvae_amd  # noqa: F401
    from lvae_amd.models.lvae import LadderVAE
    from lvae_amd.noise import PhiloxNoise
    from lvae_amd.optim import Adamax
    torch.manual_seed(42)
    m = LadderVAE(**CFG).cuda().train()
    m.noise = PhiloxNoise(seed=5, rank=rank)
    return m, Adamax(m, lr=1e-3)


def dump(path, model, extra=None):
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    torch.save({'sd': sd, 'extra': extra or {}}, path)


def main():
    mode, out, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    import lvae_amd  # noqa: F401
    from lvae_amd import dist as ldist
    from lvae_amd.engine import TrainStep, forward_pass
    if mode == 'single':
        m, opt = build(0)
        step = TrainStep(m, opt, use_graph=True)
        losses = [float(step(x.cuda())['loss']) for x in batches(steps, 1)]
        torch.cuda.synchronize()
        dump(out, m, {'losses': losses})
    elif mode == 'rccl1_split':
        # the default form of a multi-rank GPU run: ONE message after backward, outside the graphs (fwd+bwd graph | exchange | Adamax graph)
        os.environ['LVAE_FORCE_DIST'] = '1'
        rank, world, _ = ldist.init_from_env('nccl')
        m, opt = build(0)
        arena = m.pack()
        ldist.broadcast_flat(arena.params)
        ar = ldist.GradAllReduce(arena.grads, segments=arena.segments)
        assert ar.comm is not None, ar.comm_error
        assert ar.mode == 'split' and not ar.overlap and len(ar.buckets) == 1
        step = TrainStep(m, opt, use_graph=True, allreduce=ar)
        losses = [float(step(x.cuda())['loss']) for x in batches(steps, 1)]
        torch.cuda.synchronize()
        assert step.graph_a is not None and step.graph_b is not None and 'split' in step.exchange_description()
        dump(out, m, {'losses': losses, 'buckets': ar.buckets, 'launched': ar.launched})
        ar.close()
        torch.distributed.destroy_process_group()
    elif mode == 'rccl1_auto':
        # what a multi-rank bench / training run does (round 5): both exchange forms are built on one communicator, a few trial steps of
        # each are timed, the faster one is kept; every trial step is a real training step
        os.environ['LVAE_FORCE_DIST'] = '1'
        os.environ.pop('LVAE_DDP_MODE', None)
        from lvae_amd.engine import AutoExchangeStep
        rank, world, _ = ldist.init_from_env('nccl')
        m, opt = build(0)
        arena = m.pack()
        ldist.broadcast_flat(arena.params)
        step = AutoExchangeStep(m, opt, arena.grads, arena.segments, trial_steps=2, use_graph=True)
        assert sorted(step.steps) == ['overlap', 'split'] and step.ars['split'].comm is step.ars['overlap'].comm is not None
        assert len(step.ars['split'].buckets) == 1 and len(step.ars['overlap'].buckets) >= 1
        losses, forms = [], []
        for x in batches(steps, 1):
            forms.append(step.chosen or step.selector.current())
            losses.append(float(step(x.cuda())['loss']))
        torch.cuda.synchronize()
        assert step.ready and step.chosen in ('split', 'overlap') and set(step.timings_ms) == {'split', 'overlap'}
        assert all(t > 0 for t in step.timings_ms.values()) and 'chosen at run time' in step.exchange_description()
        loser = 'overlap' if step.chosen == 'split' else 'split'
        assert step.steps[loser].graph_a is None and step.steps[step.chosen].graph_a is not None
        assert (m.grad_tracker is step.ars['overlap']) == (step.chosen == 'overlap')
        dump(out, m, {'losses': losses, 'forms': forms, 'chosen': step.chosen, 'timings_ms': step.timings_ms})
        step.close()
        torch.distributed.destroy_process_group()
    elif mode == 'rccl1':
        os.environ['LVAE_FORCE_DIST'] = '1'
        os.environ['LVAE_DDP_MODE'] = 'overlap'
        rank, world, _ = ldist.init_from_env('nccl')
        m, opt = build(0)
        arena = m.pack()
        ldist.broadcast_flat(arena.params)
        ar = ldist.GradAllReduce(arena.grads, segments=arena.segments, bucket_mb=0.25)
        assert ar.comm is not None, ar.comm_error          # the captured exchange never goes through ProcessGroupNCCL
        step = TrainStep(m, opt, use_graph=True, allreduce=ar)
        assert step.overlap and m.grad_tracker is ar and len(ar.buckets) >= 3
        # the capture probe's graph really held work (with one rank an in-place all-reduce enqueues nothing and torch
        # warned "The CUDA Graph is empty"; the one-rank rehearsal now uses the out-of-place form: RCCL's kernel + a device copy)
        assert ar.scratch is not None and ar.probe_nodes_ran
        losses = [float(step(x.cuda())['loss']) for x in batches(steps, 1)]
        torch.cuda.synchronize()
        assert step.graph_a is not None and step.graph_b is None      # ONE graph: backward, exchange and Adamax together
        dump(out, m, {'losses': losses, 'buckets': ar.buckets, 'launched': ar.launched})
        ar.close()
        torch.distributed.destroy_process_group()
    elif mode == 'rccl1_fallback':
        # a collective that refuses to be captured: TrainStep's capture probe finds out before the step is captured and the same process
        # continues with the exchange outside the step graph (fwd+bwd graph | eager all-reduce | Adamax graph)
        os.environ['LVAE_FORCE_DIST'] = '1'
        os.environ['LVAE_DDP_MODE'] = 'overlap'
        rank, world, _ = ldist.init_from_env('nccl')
        m, opt = build(0)
        arena = m.pack()
        ldist.broadcast_flat(arena.params)
        ar = ldist.GradAllReduce(arena.grads, segments=arena.segments, bucket_mb=0.25)
        assert ar.comm is not None, ar.comm_error          # buckets go through the private RCCL communicator (rccl.py)
        real = ar.comm.enqueue

        def refusing(*a, **kw):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('operation not permitted when stream is capturing (simulated by the test)')
            return real(*a, **kw)

        ar.comm.enqueue = refusing
        step = TrainStep(m, opt, use_graph=True, allreduce=ar)     # its capture probe meets the refusal: split mode from the start
        assert not step.overlap and step.fallback_reason and m.grad_tracker is None
        losses = [float(step(x.cuda())['loss']) for x in batches(steps, 1)]
        torch.cuda.synchronize()
        ar.comm.enqueue = real
        assert step.graph_a is not None and step.graph_b is not None
        assert 'split' in step.exchange_description()
        dump(out, m, {'losses': losses, 'buckets': ar.buckets, 'reason': step.fallback_reason})
        torch.distributed.destroy_process_group()
    elif mode in ('gloo2', 'gloo2a'):   # gloo2a: weight-gradient kernels on side streams (async_wgrad) under the overlapped exchange
        rank, world, _ = ldist.init_from_env('gloo')
        torch.cuda.set_device(0)
        m, opt = build(rank)
        arena = m.pack()
        ldist.broadcast_flat(arena.params)
        ar = ldist.GradAllReduce(arena.grads, segments=arena.segments, bucket_mb=0.25)
        step = TrainStep(m, opt, use_graph=True, allreduce=ar, async_wgrad=(mode == 'gloo2a'), wgrad_streams=2,
                         wgrad_group_rows=(0 if mode == 'gloo2a' else 16384) or None)   # TrainStep itself must refuse to capture a gloo exchange
        assert step.overlap and not step.use_graph
        lo, hi = ldist.shard_batch(PER_RANK * world, rank, world)
        orders = []
        for x in batches(steps, world):
            step(x[lo:hi].cuda())
            orders.append(list(ar.launched))
        torch.cuda.synchronize()
        dump(out + '.%d' % rank, m, {'orders': orders, 'n_buckets': len(ar.buckets)})
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    elif mode == 'emul2':
        from lvae_amd import kernels as K
        (m0, opt0), (m1, _) = build(0), build(1)
        half = torch.full((1,), 0.5, device='cuda')
        opt0._state()
        opt0.gscale = half
        for x in batches(steps, 2):
            grads = []
            for m, sl in ((m0, slice(0, PER_RANK)), (m1, slice(PER_RANK, 2 * PER_RANK))):
                m.pack().zero_grad()
                forward_pass(m, x[sl].cuda())['loss'].backward()
                grads.append(m.arena.grads.clone())
            m0.arena.grads.copy_(grads[0] + grads[1])
            opt0.step()
            m1.arena.params.copy_(m0.arena.params)            # replicas stay identical; BN running statistics stay per rank
            K.prepared.weights_written()
        torch.cuda.synchronize()
        dump(out, m0)
        dump(out + '.rank1', m1)
    else:
        raise SystemExit('unknown mode ' + mode)


if __name__ == '__main__':
    main()
