"""Data-parallel glue: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm).

The reference has no distributed code; the defined semantics are standard DDP: every rank holds a full
replica, the batch is split contiguously, BatchNorm statistics stay per rank, and the gradient is averaged once per
step. Because gradients already live in ONE flat arena, the exchange is a handful of large all-reduces over
contiguous slices (bucketed so that each message is big enough to run at link rate and small enough to pipeline),
issued on a side HIP stream; the optimiser kernel applies the 1/world scale.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun). Returns (rank, world, local)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    force = os.environ.get('LVAE_FORCE_DIST') == '1'  # rehearse the collective path with a single rank
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(global_batch, rank, world):
    """Contiguous shard [lo, hi) of a global batch for `rank`; requires an even split (drop_last semantics)."""
    if global_batch % world != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


def bucket_slices(n, bucket_elems):
    """[(lo, hi)] covering [0, n) in chunks of at most bucket_elems."""
    out, lo = [], 0
    while lo < n:
        hi = min(n, lo + bucket_elems)
        out.append((lo, hi))
        lo = hi
    return out


class GradAllReduce:
    """SUM all-reduce of a flat gradient buffer in buckets on a side stream; `scale` = 1/world for the optimiser."""

    def __init__(self, flat_grads, group=None, bucket_mb=16.0):
        self.flat = flat_grads
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = os.environ.get('LVAE_FORCE_DIST') == '1' and dist.is_initialized()
        self.buckets = bucket_slices(flat_grads.numel(), max(1, int(bucket_mb * (1 << 20) / 4)))
        self.on_gpu = flat_grads.is_cuda
        self.stream = torch.cuda.Stream(device=flat_grads.device) if self.on_gpu else None
        self.scale = torch.full((1,), 1.0 / self.world, dtype=torch.float32, device=flat_grads.device)

    def run(self):
        """Reduce all buckets; the caller's current stream waits for completion (no host sync)."""
        if (self.world == 1 and not self.force) or os.environ.get('LVAE_SKIP_ALLREDUCE') == '1':  # second: profiling only
            return
        if not self.on_gpu:
            for lo, hi in self.buckets:
                dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
            return
        cur = torch.cuda.current_stream(self.flat.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for lo, hi in self.buckets:
                dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
        cur.wait_stream(self.stream)


def broadcast_flat(flat, src=0, group=None):
    """Identical initial replicas: broadcast rank `src`'s flat parameter arena once."""
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get('LVAE_FORCE_DIST') == '1'):
        dist.broadcast(flat, src=src, group=group)
        from . import kernels
        kernels.prepared.weights_written()  # parameter views do not share the flat buffer's version counter
