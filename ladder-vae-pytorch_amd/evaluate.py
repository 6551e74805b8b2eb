"""Offline evaluation on the HIP engine — the numerical part of the reference's evaluate.py:20-114 (whose loop lives in
the absent boilr.eval.BaseOfflineEvaluator / VAEExperimentManager.test_procedure; restated):

  * `iw_log_likelihood(model, x, S)`  — importance-weighted bound log (1/S) sum_s p(x, z_s)/q(z_s|x), per image.
    In eval mode the bottom-up pass has no noise, so it is run ONCE per batch and only top-down + likelihood are
    replayed S times (saves the 26.5 % bottom-up share of forward FLOPs per extra sample);
  * `evaluate(model, batches, S, world)` — mean ELBO / IW bound over a data set, sharded over ranks, one all-reduce;
  * `prior_samples(model, n)` and `inspect_layer_repr(model, n)` — evaluate.py:34-45, 95-114 (arrays instead of PNG grids).

CLI: python -m lvae_amd.evaluate --synthetic --ll --ll-samples 100 --ps  <model flags of main.py>
"""

import numpy as np
import torch

from . import kernels as K
from . import ops


@torch.no_grad()
def iw_log_likelihood(model, x, n_samples, use_graph=None):
    """Returns (iw_bound (N,), elbo_mean (N,)): the S-sample importance-weighted bound and the mean single-sample ELBO.

    The bottom-up pass runs once; one sample = top-down pass + likelihood + KL bookkeeping + an ONLINE update of the per-image
    (max, sum exp, sum) state. With on-device Philox noise that sample is captured once as a hipGraph and replayed (the RNG
    step counter is advanced inside the graph): ~1.5 k launches x S without host work, which is what makes S = 1000 practical.
    use_graph=None: graph when the noise source is PhiloxNoise and S >= 8; a replayed noise tape (parity tests) runs eagerly."""
    from .noise import PhiloxNoise
    was_training = model.training
    model.eval()
    try:
        if not x.is_cuda:
            raise K._C.LvaeHipError("iw_log_likelihood needs a GPU tensor")
        model._begin(x)
        img_size = tuple(int(s) for s in x.shape[2:])
        x = x.contiguous().float()
        x_pad = K.pad_crop(x, True, model.get_padded_size(x.size()), False)
        x_nhwc = x_pad if img_size == tuple(x_pad.shape[1:3]) else K.pad_crop(x, True, img_size, False)
        bu_values = model._bottomup(x_pad)           # once: sample independent in eval mode
        model.noise.end()
        N, dev = x.shape[0], x.device
        state = torch.empty((3, N), dtype=torch.float32, device=dev)
        zero = torch.zeros(1, device=dev)
        K.iw_online(state, 0)

        def one_sample():
            model.noise.begin(dev)                   # every sample: same call sites, next RNG step
            out, td = model._topdown(bu_values)
            if tuple(out.shape[1:3]) != img_size:
                out = ops.CropFn.apply(out, img_size)
            ll, _ = model.likelihood(out, x_nhwc, model.noise)
            kl_ln = ops.StackFn.apply(*td['kl'])
            kl_sep, _, _ = K.kl_bookkeeping_fwd(kl_ln, float(model.free_bits))
            elbo_sep, _ = K.elbo_loss_fwd(ll, kl_sep, zero, 1.0)
            K.iw_online(state, 1, elbo=elbo_sep)
            model.noise.end()

        if use_graph is None:
            use_graph = isinstance(model.noise, PhiloxNoise) and n_samples >= 8
        done = 0
        if use_graph:
            for _ in range(2):                        # eager: allocator warm-up (these count as samples)
                one_sample()
            done = 2
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                one_sample()
            for _ in range(done, n_samples):
                graph.replay()
        else:
            for _ in range(n_samples):
                one_sample()
        iw = torch.empty((N,), dtype=torch.float32, device=dev)
        mean = torch.empty((N,), dtype=torch.float32, device=dev)
        K.iw_online(state, 2, S=n_samples, iw=iw, mean=mean)
        return iw, mean
    finally:
        model.train(was_training)


def reduce_eval_sums(tot, process_group=None):
    """[sum of IW bounds, sum of ELBOs, image count] of this rank's shard -> totals over all ranks (one all-reduce)."""
    if torch.distributed.is_initialized() and torch.distributed.get_world_size(process_group) > 1:
        torch.distributed.all_reduce(tot, group=process_group)
    return tot


@torch.no_grad()
def evaluate(model, batches, n_samples, process_group=None):
    """Mean ELBO and IW bound over an iterable of image batches (each rank passes ITS shard of the test set)."""
    tot = torch.zeros(3, dtype=torch.float64, device=next(model.parameters()).device)
    for x in batches:
        iw, elbo = iw_log_likelihood(model, x.to(tot.device), n_samples)
        tot[0] += iw.double().sum()
        tot[1] += elbo.double().sum()
        tot[2] += x.shape[0]
    tot = reduce_eval_sums(tot, process_group)
    n = float(tot[2])
    return {'elbo/elbo': float(tot[1]) / n, 'elbo/elbo_IW_%d' % n_samples: float(tot[0]) / n, 'n_images': int(n)}


@torch.no_grad()
def prior_samples(model, n_imgs):
    """evaluate.py:34-36: unconditional samples, (n, C, H, W) in [0, 1]."""
    was_training = model.training
    model.eval()
    try:
        return model.sample_prior(n_imgs)
    finally:
        model.train(was_training)


@torch.no_grad()
def inspect_layer_repr(model, n=8):
    """evaluate.py:95-114: for every layer i, n calls of `sample_prior(n, mode_layers=range(i), constant_layers=range(i+1, L))`
    concatenated — each call (one row of the reference's image grid) draws the layers above i once for its whole batch, samples
    layer i per image and takes the mode below, so a row shows what layer i encodes. Returns a list of L tensors (n*n, C, H, W);
    the reference writes each as a PNG grid with nrow = n (image files are out of scope)."""
    was_training = model.training
    model.eval()
    try:
        out = []
        for i in range(model.n_layers):
            mode_layers = range(i)
            constant_layers = range(i + 1, model.n_layers)
            rows = [model.sample_prior(n, mode_layers=mode_layers, constant_layers=constant_layers) for _ in range(n)]
            out.append(torch.cat(rows))
        return out
    finally:
        model.train(was_training)


def main(argv=None):
    from .experiment.experiment_manager import LVAEExperiment, build_parser
    from .main import synthetic_batch
    p = build_parser()
    p.add_argument('--ll', action='store_true', help='importance-weighted log-likelihood')
    p.add_argument('--ps', action='store_true', help='prior samples -> prior_samples.npy')
    p.add_argument('--layer-repr', action='store_true', dest='layer_repr', help='layer inspection -> layer_repr_<i>.npy')
    p.add_argument('--checkpoint', type=str, default='', help='state_dict file (reference key scheme)')
    p.add_argument('--n-test', type=int, default=1000)
    args = p.parse_args(argv)
    exp = LVAEExperiment(args=args)
    model = exp.model
    if args.checkpoint:
        from .checkpoint import load_checkpoint
        load_checkpoint(args.checkpoint, model)
    if args.ll:
        gen = torch.Generator().manual_seed(args.seed)
        if args.data_npz:
            data = torch.from_numpy(np.load(args.data_npz)['data']).float()
        else:
            data = synthetic_batch(exp, args.n_test, gen)
        bs = args.test_batch_size
        res = evaluate(model, (data[i:i + bs] for i in range(0, data.shape[0], bs)), args.loglikelihood_samples)
        res['elbo/recons'], res['elbo/kl'] = float('nan'), float('nan')
        print(exp.test_log_str(res, model.global_step))
    if args.ps:
        np.save('prior_samples.npy', prior_samples(model, 64).cpu().numpy())
    if args.layer_repr:
        for i, s in enumerate(inspect_layer_repr(model, 8)):
            np.save('layer_repr_%d.npy' % i, s.cpu().numpy())


if __name__ == '__main__':
    main()
