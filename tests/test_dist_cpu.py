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
    # completion-ordered buckets: segments reported during 'backward' trigger their bucket's exchange in order; a segment that is
    # not the end of a bucket triggers nothing; finish() sends the rest. Every rank must issue the same sequence.
    segs = [(0, 1000), (1000, 1200), (1200, 5000), (5000, 5100), (5100, 10007)]
    g2 = torch.arange(10007, dtype=torch.float32) * (rank + 1)
    ar2 = ldist.GradAllReduce(g2, bucket_mb=4 * 1100 / (1 << 20), segments=segs)   # >= 1100 elements per bucket
    trace = []
    ar2.begin_step()
    for sidx in range(len(segs) - 1):           # the last segment (stem) has no marker: finish() covers it
        ar2.segment_done(sidx)
        trace.append(list(ar2.launched))
    ar2.finish()
    trace.append(list(ar2.launched))
    ok_seg = bool(torch.allclose(g2 * float(ar2.scale), want))
    # evaluation reduction over ranks (evaluate.py:86-87 restated): per-rank sums of the per-image bounds, one all-reduce
    from lvae_amd.evaluate import reduce_eval_sums
    tot = reduce_eval_sums(torch.tensor([10.0 * (rank + 1), -3.0 * (rank + 1), 4.0 + rank], dtype=torch.float64))
    q.put((rank, ok_bcast, nb, bool(torch.allclose(mean, want)), (lo, hi), ar2.buckets, trace, ok_seg, tot.tolist()))
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
    for r in res:
        assert r[5] == [(0, 1200, 1), (1200, 5000, 2), (5000, 10007, 4)], r[5]
        assert r[6] == [[], [0], [0, 1], [0, 1], [0, 1, 2]], r[6]
        assert r[7], 'segment-ordered exchange != mean gradient'
        assert r[8] == [30.0, -9.0, 9.0]


def _selector_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import lvae_amd  # noqa: F401
    from lvae_amd import dist as ldist
    ldist.init_from_env('gloo')
    # rank 0 sees 'overlap' faster, rank 1 sees it MUCH slower (e.g. its side branch is serialised): the step ends when the slowest rank
    # ends, so the cost of a form is the max over ranks, and both ranks must take the same branch
    times = {0: {'split': [0.030, 0.031, 0.029], 'overlap': [0.028, 0.028, 0.027]},
             1: {'split': [0.031, 0.031, 0.030], 'overlap': [0.034, 0.035, 0.034]}}[rank]
    sel = ldist.FormSelector(['split', 'overlap'], trial_steps=3)
    order = []
    while not sel.complete():
        f = sel.current()
        order.append(f)
        sel.record(f, times[f][len(sel.samples[f])])
    chosen = sel.decide()
    # a tie (identical numbers on both forms) goes to the first form listed, on every rank
    tie = ldist.FormSelector(['split', 'overlap'], trial_steps=1)
    tie.record('split', 0.01)
    tie.record('overlap', 0.01)
    q.put((rank, order, chosen, sel.timings_ms, tie.decide(), ldist.FormSelector(['overlap']).chosen))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_form_selection_takes_the_max_over_ranks_and_one_branch_everywhere():
    """dist.FormSelector (engine.AutoExchangeStep's decision) over gloo with two ranks whose timings disagree."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_selector_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in res:
        assert r[1] == ['split'] * 3 + ['overlap'] * 3            # trials run form by form, in the listed order, on every rank
        assert r[2] == 'split'                                    # max over ranks: split 30.7 ms, overlap 34.3 ms
        assert abs(r[3]['split'] - 30.667) < 0.01 and abs(r[3]['overlap'] - 34.333) < 0.01, r[3]
        assert r[4] == 'split' and r[5] == 'overlap'              # tie -> first form; a single form needs no trial
    assert res[0][3] == res[1][3]


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


def test_arena_is_laid_out_in_gradient_completion_order():
    """models/lvae.py:74,87-88,156,159-167 registers bottom-up / top-down layers interleaved; the gradient arena must instead follow
    reverse execution order so that data-parallel buckets are contiguous and complete early (dist.GradAllReduce)."""
    import lvae_amd  # noqa: F401
    from lvae_amd.arena import ParamArena
    from lvae_amd.dist import make_buckets
    from lvae_amd.models.lvae import LadderVAE
    from conftest import load_golden
    cfg = dict(load_golden('tiny_cifar').cfg)
    cfg['learn_top_prior'] = True
    m = LadderVAE(**cfg)
    reg_order = [k for k, _ in m.named_parameters()]
    arena = ParamArena(m, torch.device('cpu'), segment_of=m.grad_segment_of)
    segs = m.grad_segments()
    assert segs[0] == 'likelihood.' and segs[1] == 'final_top_down.' and segs[2] == 'top_down_layers.0.' and segs[-1] == 'first_bottom_up.'
    ids = [m.grad_segment_of(k) for k in arena.names[:len([p for p in m.parameters() if p.requires_grad])]]
    assert ids == sorted(ids) and set(ids) == set(range(len(segs)))
    assert m.grad_segment_of('top_down_layers.2.top_prior_params') == len(segs) - 1      # arrives via autograd accumulation: last
    assert len(arena.segments) == len(segs) and arena.segments[0][1] == 0 and arena.segments[-1][2] == arena.n_train
    assert [sg[0] for sg in arena.segments] == list(range(len(segs)))       # explicit ids: what the model's segment markers report
    assert all(a[2] == b[1] for a, b in zip(arena.segments, arena.segments[1:]))
    # slots still address every parameter; values survived the permutation; state_dict order is the registration order
    assert list(k for k, _ in m.named_parameters()) == reg_order
    for k, p in m.named_parameters():
        off, n = arena.slots[k]
        assert p.numel() == n and p.data_ptr() == arena.params.data_ptr() + 4 * off
    bk = make_buckets(arena.segments, 1)
    assert len(bk) == len(segs) and [b[2] for b in bk] == list(range(len(segs)))
    bk = make_buckets(arena.segments, arena.n_train)
    assert bk == [(0, arena.n_train, len(segs) - 1)]


def test_segment_ids_with_gaps_keep_buckets_aligned():
    """A segment without trainable parameters has no arena range; the markers still report model-level ids (positional
    indices would shift every later bucket by one and launch it before its gradients exist)."""
    import lvae_amd  # noqa: F401
    from lvae_amd.dist import GradAllReduce, make_buckets
    segs = [(0, 0, 100), (2, 100, 300), (3, 300, 350), (5, 350, 1000)]      # ids 1 and 4 own nothing
    assert make_buckets(segs, 1) == [(0, 100, 0), (100, 300, 2), (300, 350, 3), (350, 1000, 5)]
    assert make_buckets(segs, 250) == [(0, 300, 2), (300, 1000, 5)]
    g = torch.zeros(1000)
    ar = GradAllReduce(g, bucket_mb=4 * 250 / (1 << 20), segments=segs)
    assert ar.buckets == [(0, 300, 2), (300, 1000, 5)] and ar.by_segment == {2: 0, 5: 1}
    # not active (world 1): segment_done is a no-op, but the id -> bucket search must still be exact
    import bisect
    ends = ar._ends
    assert [bisect.bisect_right(ends, s) for s in range(6)] == [0, 0, 1, 1, 1, 2]


def test_bench_self_launches_its_ranks_as_child_processes():
    """`python bench.py --gpus N` without WORLD_SIZE starts the N ranks itself (torch.distributed.run as a CHILD
    process, never an exec), relays rank 0's JSON line, and exits non-zero when a child does. Rehearsed over gloo without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'LVAE_FORCE_DIST')}
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--launch-check']
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['value'] == 3.0 and rec['self_launched'] is True     # 1 + 2: both ranks took part
    env['LVAE_LAUNCH_CHECK_FAIL_RANK'] = '1'
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0
