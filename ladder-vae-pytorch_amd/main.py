"""`python -m lvae_amd.main <flags>` — training entry point with the reference's flag surface (main.py:1-13 there is
`Trainer(LVAEExperiment()).run()` on boilr; here a minimal loop on the HIP engine, one process per GPU).

Launch N ranks with:  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 -m lvae_amd.main ...
"""
import time

import numpy as np
import torch

from . import dist as ldist
from .engine import TrainStep
from .experiment.experiment_manager import LVAEExperiment


def synthetic_batch(exp, batch, gen):
    shape = (batch, exp.color_ch) + tuple(exp.img_size)
    u = torch.rand(shape, generator=gen)
    return (u > 0.5).float() if exp.args.likelihood == 'bernoulli' else torch.floor(256 * u) / 255


def main(argv=None):
    rank, world, local = ldist.init_from_env()
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    exp = LVAEExperiment(argv=argv)
    args = exp.args
    loader = None
    if not (args.synthetic or args.data_npz):
        from .data import DatasetLoader  # the reference's on-disk formats (experiment/data.py); nothing is downloaded here
        try:
            loader = DatasetLoader(args)
        except RuntimeError as e:
            raise SystemExit("%s\n(or pass --synthetic / --data-npz FILE)" % e)
    model, opt = exp.model, exp.optimizer
    model.noise.seed ^= rank * 0x9E3779B9
    if args.resume:
        from .checkpoint import load_checkpoint
        load_checkpoint(args.resume, model, opt)
    model.train()
    arena = model.pack()
    per_rank = args.batch_size // max(1, world)
    data = None
    if args.data_npz:
        data = torch.from_numpy(np.load(args.data_npz)['data']).float()
    if args.simple_data_dependent_init and not args.resume:
        # experiment_manager.py:61-72: the first batch_size training images (parity unpinned, see init.py)
        from .init import data_dependent_init
        x0 = loader.train.dataset.tensors[0][:args.batch_size] if loader is not None else data[:args.batch_size] if data is not None else synthetic_batch(exp, args.batch_size, torch.Generator().manual_seed(args.seed))
        n = data_dependent_init(model, x0.to(exp.device))
        if rank == 0:
            print('data-dependent init: %d convolutions rescaled' % n)
    ldist.broadcast_flat(arena.params)
    allreduce = ldist.GradAllReduce(arena.grads, segments=arena.segments) if world > 1 else None
    step_fn = TrainStep(model, opt, beta=1.0, use_graph=not args.no_graph and args.beta_anneal == 0, allreduce=allreduce)
    if rank == 0:
        print(exp.run_description)
        print('parameters: %d   world size: %d   per-rank batch: %d' % (sum(p.numel() for p in model.parameters()), world,
                                                                       args.batch_size // world))
    if args.batch_size % world:
        raise SystemExit('--batch-size must be divisible by the world size')
    per_rank = args.batch_size // world
    gen = torch.Generator().manual_seed(args.seed + 1000 * rank)
    steps = args.steps or args.max_steps
    batches = None
    t0, seen = time.time(), 0
    for step in range(1, steps + 1):
        if loader is not None:
            if step == 1 or batches is None:
                batches = iter(loader.train)
            try:
                xb = next(batches)[0]
            except StopIteration:                      # next epoch: reshuffled by the DataLoader
                batches = iter(loader.train)
                xb = next(batches)[0]
            lo, hi = ldist.shard_batch(args.batch_size, rank, world)
            x = xb[lo:hi]
        elif data is not None:
            idx = torch.randint(0, data.shape[0], (args.batch_size,), generator=torch.Generator().manual_seed(args.seed + step))
            lo, hi = ldist.shard_batch(args.batch_size, rank, world)
            x = data[idx[lo:hi]]
        else:
            x = synthetic_batch(exp, per_rank, gen)
        if args.beta_anneal != 0:
            step_fn.beta = exp.beta()
        out = step_fn(x.to(exp.device, non_blocking=True))
        seen += args.batch_size
        if rank == 0 and (step % args.log_every == 0 or step == steps):
            m = exp.get_metrics_dict(out)
            dt = time.time() - t0
            print(exp.train_log_str(m, step) + '   [{:.0f} img/s]'.format(seen / dt))
            t0, seen = time.time(), 0
    if args.save_checkpoint and rank == 0:
        from .checkpoint import save_checkpoint
        save_checkpoint(args.save_checkpoint, model, opt)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
