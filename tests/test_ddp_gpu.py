"""Data-parallel semantics on the GPU box (one MI355X): SURVEY.md §8e — every rank ≡ the reference run on its shard (local
BatchNorm statistics), gradients averaged once per step.

 1. two B/2 shards run one after the other through the HIP engine, gradients averaged, against the CPU ORACLE run per shard:
    "N-rank averaged gradient == mean of per-shard single-rank gradients", pinned to the oracle without a second GPU;
 2. the N > 1 code path itself (completion-ordered buckets exchanged on a side stream during backward, captured in the step's
    hipGraph) as ONE forced rank over RCCL, against the plain single-rank step;
 3. two real ranks over gloo sharing the GPU (eager), against a single process playing both ranks.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, 'tests', 'ddp_worker.py')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(mode, out, steps, rank=0, world=1, port=None, wait=True):
    env = dict(os.environ)
    env.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port or _free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('LVAE_FORCE_DIST', None)
    p = subprocess.Popen([sys.executable, WORKER, mode, out, str(steps)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if not wait:
        return p
    log, _ = p.communicate(timeout=600)
    assert p.returncode == 0, log.decode()[-3000:]
    return p


def _max_rel(a, b):
    worst = 0.0
    for k in a:
        if a[k].dtype.is_floating_point:
            d = float((a[k].double() - b[k].double()).norm())
            worst = max(worst, d / (float(b[k].double().norm()) + 1e-12))
    return worst


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_two_shard_mean_gradient_matches_oracle_per_shard(dtype):
    """bf16: BASELINE configs[3]'s arithmetic (bf16 matrix-core operands, fp32 accumulate / statistics / KL / likelihood) under the same
    data-parallel semantics, against the fp32 oracle at north_star's bf16 tolerance (ELBO 1e-2 relative; gradients looser still)."""
    import lvae_amd  # noqa: F401
    from lvae_amd.engine import forward_pass
    from lvae_amd.models.lvae import LadderVAE
    from lvae_amd.noise import TapeNoise
    from oracle import lvae_ref as R
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from ddp_worker import CFG
    torch.manual_seed(42)
    model = LadderVAE(**CFG)
    init = {k: v.clone() for k, v in model.state_dict().items()}
    B = 48
    x = torch.floor(256 * torch.rand(B, 3, 32, 32, generator=torch.Generator().manual_seed(3))) / 255
    shards = [x[:B // 2], x[B // 2:]]
    # oracle: each shard is an independent reference run (its own BatchNorm batch statistics and noise)
    ref_grads, tapes, ref_loss = None, [], []
    for r, xs in enumerate(shards):
        sd = {k: v.clone() for k, v in init.items()}
        pkeys = [k for k in sd if R.is_parameter_key(k)]
        for k in pkeys:
            sd[k].requires_grad_(True)
        tape = R.Tape(gen=torch.Generator().manual_seed(100 + r))
        fp, _ = R.forward_pass(sd, CFG, xs, tape, param_keys=pkeys)
        fp['loss'].backward()
        tapes.append(tape.entries)
        ref_loss.append(float(fp['loss']))
        g = {k: sd[k].grad for k in pkeys}
        ref_grads = g if ref_grads is None else {k: ref_grads[k] + g[k] for k in g}
    ref_grads = {k: v / 2 for k, v in ref_grads.items()}
    # HIP engine: the same model object sees the two shards one after the other; flat gradient arenas are summed and halved,
    # exactly what the all-reduce + the optimiser's 1/world scale do
    model.cuda().train()
    model.compute_dtype = dtype
    loss_tol, grad_tol, norm_tol = (1e-5, 1e-4, 1e-6) if dtype == 'f32' else (1e-2, 5e-2, 1e-2)
    arena = model.pack()
    flat = torch.zeros_like(arena.grads)
    for r, xs in enumerate(shards):
        model.noise = TapeNoise(tapes[r])
        arena.zero_grad()
        out = forward_pass(model, xs.cuda())
        assert abs(float(out['loss']) - ref_loss[r]) <= loss_tol * abs(ref_loss[r])
        out['loss'].backward()
        flat += arena.grads
    arena.grads.copy_(flat * 0.5)
    worst, gsq, rsq = 0.0, 0.0, 0.0
    for k, p in model.named_parameters():
        ref = ref_grads[k].double()
        g = p.grad.detach().cpu().double()
        gsq += float(g.pow(2).sum())
        rsq += float(ref.pow(2).sum())
        if float(ref.norm()) < 1e-5 * max(1.0, ref.numel() ** 0.5):
            continue
        worst = max(worst, float((g - ref).norm() / ref.norm()))
    assert worst < grad_tol, worst
    assert abs(gsq ** 0.5 - rsq ** 0.5) <= norm_tol * rsq ** 0.5


def test_forced_rccl_rank_overlapped_graph_step_matches_single_rank(tmp_path):
    a, b = str(tmp_path / 'single.pt'), str(tmp_path / 'rccl1.pt')
    _run('single', a, 5)
    _run('rccl1', b, 5)
    ra, rb = torch.load(a), torch.load(b)
    la, lb = ra['extra']['losses'], rb['extra']['losses']
    assert max(abs(u - v) / abs(u) for u, v in zip(la, lb)) < 1e-6, (la, lb)
    assert _max_rel(rb['sd'], ra['sd']) < 1e-6
    nb = len(rb['extra']['buckets'])
    assert nb >= 3 and rb['extra']['launched'] == list(range(nb))     # every bucket exchanged once, in completion order


def test_forced_rccl_rank_default_split_exchange_matches_single_rank(tmp_path):
    """The default exchange of a multi-rank GPU run (one message after backward, between the two graphs) on one forced rank: parameters,
    BatchNorm statistics and losses of 5 steps equal the plain single-rank run."""
    a, b = str(tmp_path / 'single.pt'), str(tmp_path / 'split.pt')
    _run('single', a, 5)
    _run('rccl1_split', b, 5)
    ra, rb = torch.load(a), torch.load(b)
    la, lb = ra['extra']['losses'], rb['extra']['losses']
    assert max(abs(u - v) / abs(u) for u, v in zip(la, lb)) < 1e-6, (la, lb)
    assert _max_rel(rb['sd'], ra['sd']) < 1e-6
    assert rb['extra']['launched'] == [0]


def test_exchange_form_chosen_at_run_time_on_a_forced_rank_matches_single_rank(tmp_path):
    """engine.AutoExchangeStep on one forced RCCL rank: 2 x (2 eager + capture + 2 timed) trial steps, then the faster
    form; since every trial step is a real training step, parameters / BatchNorm statistics / losses after 14 steps equal the plain
    single-rank run's, whichever form was running at which step."""
    a, b = str(tmp_path / 'single.pt'), str(tmp_path / 'auto.pt')
    _run('single', a, 14)
    _run('rccl1_auto', b, 14)
    ra, rb = torch.load(a), torch.load(b)
    la, lb = ra['extra']['losses'], rb['extra']['losses']
    assert max(abs(u - v) / abs(u) for u, v in zip(la, lb)) < 1e-6, (la, lb)
    assert _max_rel(rb['sd'], ra['sd']) < 1e-6
    forms = rb['extra']['forms']
    assert forms[:5] == ['split'] * 5 and forms[5:10] == ['overlap'] * 5 and set(forms[10:]) == {rb['extra']['chosen']}, forms
    for k in ra['sd']:
        if k.endswith('num_batches_tracked'):
            assert int(ra['sd'][k]) == int(rb['sd'][k]) == 14, k


def test_capture_refusing_collective_falls_back_to_split_in_process(tmp_path):
    """if the gradient exchange cannot be captured (GradAllReduce.capture_probe meets a refusal), the SAME process
    continues with the exchange outside the step graph; parameters, BatchNorm statistics (incl. num_batches_tracked) and losses equal
    the plain single-rank run."""
    a, b = str(tmp_path / 'single.pt'), str(tmp_path / 'fb.pt')
    _run('single', a, 5)
    _run('rccl1_fallback', b, 5)
    ra, rb = torch.load(a), torch.load(b)
    la, lb = ra['extra']['losses'], rb['extra']['losses']
    assert max(abs(u - v) / abs(u) for u, v in zip(la, lb)) < 1e-6, (la, lb)
    assert _max_rel(rb['sd'], ra['sd']) < 1e-6
    for k in ra['sd']:
        if k.endswith('num_batches_tracked'):
            assert int(ra['sd'][k]) == int(rb['sd'][k]) == 5, k
    assert 'capturing' in rb['extra']['reason']


@pytest.mark.parametrize('mode', ['gloo2', 'gloo2a'])
def test_two_gloo_ranks_on_one_gpu_match_two_rank_emulation(tmp_path, mode):
    """gloo2a: weight-gradient kernels on side streams while buckets are exchanged during backward — a bucket's all-reduce
    must be ordered behind the side-stream kernels that accumulate into it, or the ranks sum stale gradients."""
    e, g = str(tmp_path / 'emul.pt'), str(tmp_path / 'gloo.pt')
    _run('emul2', e, 2)
    port = _free_port()
    procs = [_run(mode, g, 2, rank=r, world=2, port=port, wait=False) for r in range(2)]
    logs = [p.communicate(timeout=900)[0].decode() for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    r0, r1, em0, em1 = torch.load(g + '.0'), torch.load(g + '.1'), torch.load(e), torch.load(e + '.rank1')
    params = [k for k in r0['sd'] if 'running_' not in k and 'num_batches' not in k]
    # replicas stay identical, and equal to the emulation
    for k in params:
        assert torch.equal(r0['sd'][k], r1['sd'][k]), k
    assert _max_rel({k: r0['sd'][k] for k in params}, {k: em0['sd'][k] for k in params}) < 1e-6
    # BatchNorm running statistics are per rank (no SyncBN in the reference): each rank matches ITS shard's emulation
    stats = [k for k in r0['sd'] if 'running_' in k]
    assert _max_rel({k: r0['sd'][k] for k in stats}, {k: em0['sd'][k] for k in stats}) < 1e-5
    assert _max_rel({k: r1['sd'][k] for k in stats}, {k: em1['sd'][k] for k in stats}) < 1e-5
    assert any(not torch.equal(r0['sd'][k], r1['sd'][k]) for k in stats)
    nb = r0['extra']['n_buckets']
    assert nb >= 3 and all(o == list(range(nb)) for o in r0['extra']['orders'] + r1['extra']['orders'])
