"""N > 1 data-parallel path on CPU: world_size 2 over gloo (the same code runs over RCCL with backend 'nccl')."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import lvae_amd  # noqa: F401
    from lvae_amd import dist as ldist
    r, w, _ = ldist.init_from_env('gloo')
    assert (r, w) == (rank, world)
    # identical replicas after the broadcast
    flat = torch.full((1000,), float(rank + 1))
    ldist.broadcast_flat(flat)
    ok_bcast = bool((flat == 1.0).all())
    # bucketed SUM all-reduce + 1/world scale == mean of the per-rank gradients
    g = torch.arange(10007, dtype=torch.float32) * (rank + 1)
    ar = ldist.GradAllReduce(g, bucket_mb=0.01)  # 2621-element buckets -> 4 messages
    nb = len(ar.buckets)
    ar.run()
    mean = g * float(ar.scale)
    want = torch.arange(10007, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    lo, hi = ldist.shard_batch(64, rank, world)
    q.put((rank, ok_bcast, nb, bool(torch.allclose(mean, want)), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo_allreduce_and_sharding():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] for r in res), 'broadcast did not produce identical replicas'
    assert all(r[2] == 4 for r in res)
    assert all(r[3] for r in res), 'bucketed all-reduce * 1/world != mean gradient'
    assert res[0][4] == (0, 32) and res[1][4] == (32, 64)


def test_bucket_slices_cover_exactly():
    import lvae_amd  # noqa: F401
    from lvae_amd.dist import bucket_slices, shard_batch
    for n, b in ((10, 3), (9, 3), (1, 5), (14467684, 4 << 20)):
        sl = bucket_slices(n, b)
        assert sl[0][0] == 0 and sl[-1][1] == n
        assert all(a[1] == c[0] for a, c in zip(sl, sl[1:])) and all(hi - lo <= b for lo, hi in sl)
    import pytest
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)
