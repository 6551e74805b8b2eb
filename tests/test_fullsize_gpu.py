"""Full-size parity: one whole training step of the BASELINE architectures — exactly as bench.py builds them
(lvae_amd.configs) — through engine.TrainStep, eager AND as a replayed hipGraph (prepared Winograd weights, grouped weight
gradients, statistics epilogues: everything the timed path uses), against the CPU oracle on the same weights, input and
noise tape. Reference semantics: models/lvae.py:172-214, experiment/experiment_manager.py:322-350.

  cfg3  CIFAR10 15-layer fp32, batch 256  — the configuration bench.py times
  cfg2  static-MNIST 12-layer (architecture of BASELINE configs[1]) in fp32, batch 256
  cfg5  64x64 20-layer (architecture of BASELINE configs[4]) in fp32, batch 16

The oracle pass costs 10-40 s of CPU per case and runs once per case (module cache). Measured errors are written to
gpurun_out/fullsize_parity.json; the tolerances asserted here are the ones DESIGN.md §2 states and justifies.
"""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {'cfg3_cifar15_b256': ('cifar15', 256), 'cfg2_mnist12_b256': ('mnist12', 256), 'cfg5_celeba20_b16': ('celeba20', 16)}
_oracle = {}
_report = {}


def oracle_step(case):
    """(cfg, state_dict with .grad, x, tape entries, forward_pass scalars) of one CPU oracle step, cached per case."""
    if case in _oracle:
        return _oracle[case]
    import lvae_amd  # noqa: F401
    from lvae_amd import configs
    from lvae_amd.models.lvae import LadderVAE
    from oracle import lvae_ref as R
    name, batch = CASES[case]
    cfg = configs.BY_NAME[name]
    torch.manual_seed(42)                      # bench.py's init seed
    sd = {k: v.clone() for k, v in LadderVAE(**cfg).state_dict().items()}
    init = {k: v.clone() for k, v in sd.items()}
    x = configs.synthetic_images(cfg, batch, torch.Generator().manual_seed(1234))
    pkeys = [k for k in sd if R.is_parameter_key(k)]
    for k in pkeys:
        sd[k].requires_grad_(True)
    tape = R.Tape(gen=torch.Generator().manual_seed(6))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    fp, mo = R.forward_pass(sd, cfg, x, tape, param_keys=pkeys)
    fp['loss'].backward()
    scal = {k: float(fp[k]) for k in ('loss', 'elbo', 'recons', 'kl', 'l2')}
    scal['kl_avg_layerwise'] = fp['kl_avg_layerwise'].detach().clone()
    scal['elbo_sep'] = fp['elbo_sep'].detach().clone()
    grads = {k: sd[k].grad for k in pkeys if sd[k].grad is not None}
    del fp, mo
    _oracle[case] = (cfg, init, x, tape.entries, scal, grads)
    return _oracle[case]


def run_engine(case, use_graph, dtype='f32'):
    import lvae_amd  # noqa: F401
    from lvae_amd import kernels as K
    from lvae_amd.models.lvae import LadderVAE
    from lvae_amd.noise import TapeNoise
    from lvae_amd.optim import Adamax
    from lvae_amd.engine import TrainStep
    cfg, init, x, entries, scal, grads = oracle_step(case)
    model = LadderVAE(**cfg)
    model.load_state_dict(init)
    model.cuda().train()
    model.compute_dtype = dtype
    model.noise = TapeNoise(entries, loop=True)
    opt = Adamax(model, lr=0.0)                # lr 0: every step sees the same weights, so every step must match the oracle
    K.prepared.entries.clear()
    K.prepared.table = None
    step = TrainStep(model, opt, use_graph=use_graph, eager_warmup=2)
    xg = x.cuda()
    n = 4 if use_graph else 2                  # graph: 2 eager + capture/replay + replay; eager: per-launch, then prepared weights
    for _ in range(n):
        out = step(xg)
    torch.cuda.synchronize()
    assert model.noise.exhausted()
    if use_graph:
        assert step.graph_a is not None
    res = {k: float(out[k]) for k in ('loss', 'elbo', 'recons', 'kl', 'l2')}
    kl_layers = out['kl_avg_layerwise'].cpu()
    worst, worst_key, gsq, ref_sq = 0.0, None, 0.0, 0.0
    for k, p in model.named_parameters():
        ref = grads.get(k)
        if ref is None:
            continue
        g = p.grad.detach().cpu().double()
        gsq += float(g.pow(2).sum())
        ref_sq += float(ref.double().pow(2).sum())
        rn = float(ref.double().norm())
        if rn < 1e-5 * max(1.0, ref.numel() ** 0.5):
            # biases in front of a BatchNorm have a mathematically zero gradient: rounding noise on both sides
            assert float(g.norm()) < 1e-2 * max(1.0, ref.numel() ** 0.5), k
            continue
        e = float((g - ref.double()).norm()) / rn
        if e > worst:
            worst, worst_key = e, k
    K.prepared.entries.clear()
    K.prepared.table = None
    K.set_precision('f32')
    rec = {'mode': ('graph' if use_graph else 'eager') + ('' if dtype == 'f32' else '-' + dtype)}
    for k in ('loss', 'elbo', 'recons', 'kl', 'l2'):
        rec[k] = {'hip': res[k], 'oracle': scal[k], 'rel': abs(res[k] - scal[k]) / max(abs(scal[k]), 1e-30)}
    rec['kl_layer_max_abs'] = float((kl_layers - scal['kl_avg_layerwise']).abs().max())
    rec['kl_layer_max_rel'] = float(((kl_layers - scal['kl_avg_layerwise']).abs() / scal['kl_avg_layerwise'].abs().clamp(min=1e-3)).max())
    rec['grad_worst_rel_l2'] = worst
    rec['grad_worst_key'] = worst_key
    rec['gradnorm'] = {'hip': gsq ** 0.5, 'oracle': ref_sq ** 0.5, 'rel': abs(gsq ** 0.5 - ref_sq ** 0.5) / ref_sq ** 0.5}
    _report['%s/%s' % (case, rec['mode'])] = rec
    _dump_report()
    return rec


def _dump_report():
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'fullsize_parity.json'), 'w') as f:
            json.dump(_report, f, indent=1)
    except OSError:
        pass


@pytest.mark.parametrize('use_graph', [False, True], ids=['eager', 'graph'])
@pytest.mark.parametrize('case', list(CASES))
def test_full_training_step_matches_oracle(case, use_graph):
    rec = run_engine(case, use_graph)
    # SURVEY.md §8(c) fp32 tolerances, unloosened: loss / elbo relative 1e-5 (BASELINE asks 1e-3), per-layer KL 1e-5 relative
    # (+1e-4 absolute), per-tensor gradient relative L2 1e-4; the global gradient norm to 1e-6.
    # Measured on MI355X (profiles/r02_fullsize_parity.json): scalars <= 1.7e-7, per-layer KL <= 6e-7, worst tensor 9.1e-6, norm 3e-8.
    for k in ('loss', 'elbo', 'recons', 'kl', 'l2'):
        assert rec[k]['rel'] <= 1e-5, (k, rec[k])
    assert rec['kl_layer_max_rel'] <= 1e-5 or rec['kl_layer_max_abs'] <= 1e-4, rec
    assert rec['grad_worst_rel_l2'] <= 1e-4, (rec['grad_worst_key'], rec['grad_worst_rel_l2'])
    assert rec['gradnorm']['rel'] <= 1e-6, rec['gradnorm']


@pytest.mark.parametrize('case', list(CASES))
def test_bf16_training_step_is_within_the_stated_tolerance_of_the_fp32_oracle(case):
    """The three BASELINE architectures with compute_dtype = 'bf16' (bf16 matrix-core operands in the 3x3 convolutions of the 8x8 and
    larger levels — forward, dgrad and weight gradient — everything else fp32): SURVEY.md §8(c) asks elbo relative <= 1e-2 against the
    fp32 oracle on the same weights, input and noise tape (cfg3 = BASELINE configs[3] per-GPU shard; cfg2 / cfg5 = the architectures of
    configs[1] / configs[4])."""
    rec = run_engine(case, True, dtype='bf16')
    for k in ('loss', 'elbo', 'recons'):
        assert rec[k]['rel'] <= 1e-2, (k, rec[k])
    assert rec['kl']['rel'] <= 5e-2, rec['kl']                # KL is the small difference of large terms
    assert rec['gradnorm']['rel'] <= 5e-2, rec['gradnorm']
    assert rec['grad_worst_rel_l2'] > 1e-4                      # ... and it really ran in reduced precision
    # no single gradient tensor drifts further than bf16 storage + bf16 operands explain (measured: 1.6e-2 cfg3, 1.8e-2 cfg5, round 3):
    # a storage or fusion change that moves this silently is a regression
    assert rec['grad_worst_rel_l2'] <= 3e-2, (rec['grad_worst_key'], rec['grad_worst_rel_l2'])


def test_celeba20_shard_step_at_its_real_per_gpu_batch():
    """BASELINE configs[4] at its real per-GPU batch (128 images of 64x64, 20 layers): a different N*H*W per level than the batch-16 oracle
    case above, hence other kernel variants. No CPU oracle at this size (minutes per step); the properties the step must have on one noise
    tape and one set of weights: every scalar finite, the replayed hipGraph equals the eager launches, and the bf16 step's ELBO is
    within SURVEY.md §8(c)'s 1e-2 of the fp32 HIP step (which the smaller cases pin to the oracle)."""
    import lvae_amd  # noqa: F401
    from lvae_amd import configs
    from lvae_amd import kernels as K
    from lvae_amd.engine import TrainStep
    from lvae_amd.models.lvae import LadderVAE
    from lvae_amd.noise import FrozenNoise
    from lvae_amd.optim import Adamax
    cfg = configs.CELEBA20
    noise = FrozenNoise(seed=9)                        # the first forward's draws, replayed by every later forward of all three models
    x = configs.synthetic_images(cfg, 128, torch.Generator().manual_seed(77)).cuda()
    torch.manual_seed(42)
    init = {k: v.clone() for k, v in LadderVAE(**cfg).state_dict().items()}
    res = {}
    for mode, dtype, use_graph in (('f32-eager', 'f32', False), ('f32-graph', 'f32', True), ('bf16-graph', 'bf16', True)):
        model = LadderVAE(**cfg)
        model.load_state_dict(init)
        model.cuda().train()
        model.compute_dtype = dtype
        model.noise = noise
        K.prepared.entries.clear()
        K.prepared.table = None
        step = TrainStep(model, Adamax(model, lr=0.0), use_graph=use_graph, eager_warmup=2)
        outs = []
        for i in range(4 if use_graph else 2):         # lr 0 and one tape: every step must give the same numbers
            outs.append({k: float(v) for k, v in step(x).items() if k in ('loss', 'elbo', 'recons', 'kl')})
        torch.cuda.synchronize()
        res[mode] = outs[-1]
        gn = sum(float(p.grad.double().pow(2).sum()) for p in model.parameters() if p.grad is not None) ** 0.5
        res[mode]['gradnorm'] = gn
        for k, v in res[mode].items():
            assert v == v and abs(v) < 1e30, (mode, k, v)
        del step, model
    K.prepared.entries.clear()
    K.prepared.table = None
    K.set_precision('f32')
    for k in ('loss', 'elbo', 'recons', 'kl'):
        a, b = res['f32-eager'][k], res['f32-graph'][k]
        assert abs(a - b) <= 1e-6 * abs(a), (k, a, b)              # same kernels, same order: the graph replays the eager step
    assert abs(res['f32-eager']['gradnorm'] - res['f32-graph']['gradnorm']) <= 1e-6 * res['f32-eager']['gradnorm']
    for k in ('loss', 'elbo', 'recons'):
        a, b = res['f32-graph'][k], res['bf16-graph'][k]
        assert abs(a - b) <= 1e-2 * abs(a), (k, a, b)
    assert abs(res['f32-graph']['gradnorm'] - res['bf16-graph']['gradnorm']) <= 5e-2 * res['f32-graph']['gradnorm']
    _report['cfg5_celeba20_b128/properties'] = res
    _dump_report()


def test_iw_1000_sample_evaluation_on_the_64x64_20_layer_model():
    """BASELINE configs[4]'s evaluation (1000-sample importance-weighted bound, evaluate.py:30,56-66) on its architecture: bottom-up once,
    one captured sample graph replayed 1000 times, online log-sum-exp. No oracle at this size (1000 CPU forwards); the properties the
    estimator must have: finite, IW bound >= mean ELBO (Jensen), and tighter with more samples (S = 1000 vs the first 10 of a second run
    differ in the right direction on average)."""
    import lvae_amd  # noqa: F401
    from lvae_amd import configs
    from lvae_amd.evaluate import iw_log_likelihood
    from lvae_amd.models.lvae import LadderVAE
    from lvae_amd.noise import PhiloxNoise
    cfg = configs.CELEBA20
    torch.manual_seed(42)
    model = LadderVAE(**cfg).cuda()
    x = configs.synthetic_images(cfg, 8, torch.Generator().manual_seed(5)).cuda()
    model.noise = PhiloxNoise(seed=2)
    iw1000, mean1000 = iw_log_likelihood(model, x, 1000)
    model.noise = PhiloxNoise(seed=2)
    iw10, _ = iw_log_likelihood(model, x, 10)
    assert torch.isfinite(iw1000).all() and torch.isfinite(mean1000).all()
    assert float((iw1000 - mean1000).min()) >= -1e-2
    assert float((iw1000 - iw10).mean()) > 0.0
