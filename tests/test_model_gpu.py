"""Whole-model parity on the GPU: the HIP engine against vectors captured from the real reference (tests/golden)
and against the CPU oracle, replaying the reference's noise tape."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

HIP_CASES = ['tiny_mnist', 'tiny_cifar', 'tiny_eval', 'tiny_cabdcabd', 'tiny_bacdbac', 'tiny_nobn_selu', 'tiny_gauss',
             'tiny_discrlog']


def build(g, training=True):
    import lvae_amd  # noqa: F401
    from lvae_amd.models.lvae import LadderVAE
    from lvae_amd.noise import TapeNoise
    torch.manual_seed(0)
    m = LadderVAE(**g.cfg)
    sd = g.state_dict()
    missing = m.load_state_dict(sd, strict=True)
    m.cuda()
    m.train(training)
    return m, TapeNoise


def relerr(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-20))


@pytest.mark.parametrize('name', HIP_CASES)
def test_forward_backward_matches_reference(name):
    g = load_golden(name)
    training = name != 'tiny_eval'
    m, TapeNoise = build(g, training)
    m.noise = TapeNoise(g.seq('tape'))
    x = g.t('x').cuda()
    out = m(x)
    assert m.noise.exhausted()
    ref = g.group('out')
    tol = dict(rtol=1e-5, atol=1e-4)       # per-sample ll, kl_sep, per-layer KL: 1e-5 * |ref| + 1e-4
    torch.testing.assert_close(out['ll'].cpu(), ref['ll'], **tol)
    torch.testing.assert_close(out['kl_sep'].cpu(), ref['kl_sep'], **tol)
    torch.testing.assert_close(out['kl'].cpu(), ref['kl'], **tol)
    torch.testing.assert_close(out['kl_loss'].cpu(), ref['kl_loss'], **tol)
    torch.testing.assert_close(out['kl_avg_layerwise'].cpu(), ref['kl_avg_layerwise'], **tol)
    torch.testing.assert_close(out['logp'].cpu(), ref['logp'], **tol)
    for i, z in enumerate(out['z']):
        assert tuple(z.shape) == tuple(ref['z.%d' % i].shape)
        torch.testing.assert_close(z.cpu(), ref['z.%d' % i], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(out['kl_spatial'][i].cpu(), ref['kl_spatial.%d' % i], rtol=1e-4, atol=1e-3)
    if g.cfg['likelihood_form'] == 'bernoulli':
        torch.testing.assert_close(out['out_mean'].cpu(), ref['out_mean'], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(out['likelihood_params'].cpu(), ref['likelihood_params'], rtol=1e-4, atol=1e-5)
        # mode / sample are thresholded: allow the few pixels whose probability sits within rounding of the threshold
        assert (out['out_mode'].cpu() != ref['out_mode']).float().mean() < 1e-3
        assert (out['out_sample'].cpu() != ref['out_sample']).float().mean() < 1e-3
    elif g.cfg['likelihood_form'] in ('gaussian', 'discr_log'):
        second = 'logvar' if g.cfg['likelihood_form'] == 'gaussian' else 'logscale'
        torch.testing.assert_close(out['out_mean'].cpu(), ref['out_mean'], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(out['out_mode'].cpu(), ref['out_mode'], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(out['out_sample'].cpu(), ref['out_sample'], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(out['likelihood_params']['mean'].cpu(), ref['likelihood_params.mean'], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(out['likelihood_params'][second].cpu(), ref['likelihood_params.' + second], rtol=1e-4, atol=1e-4)
    else:
        assert out['out_mean'] is None and out['out_mode'] is None
        torch.testing.assert_close(out['likelihood_params']['all_params'].cpu(), ref['likelihood_params.all_params'],
                                   rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(out['out_sample'].cpu(), ref['out_sample'], rtol=1e-4, atol=1e-4)
    if not training:
        return
    # loss and gradients (experiment_manager.py:329-344)
    loss = (-out['ll']).mean() + out['kl_loss']
    fp = g.group('fp')
    torch.testing.assert_close(loss.detach().cpu(), fp['loss'], rtol=1e-5, atol=0)   # loss relative 1e-5
    m.zero_grad()
    loss.backward()
    grads = g.group('grad')
    gsq = 0.0
    worst = (0.0, None)
    for k, p in m.named_parameters():
        if k not in grads:
            assert p.grad is None or not p.requires_grad or float(p.grad.abs().max()) == 0.0, k
            continue
        gsq += float(p.grad.double().pow(2).sum())
        ref_g = grads[k]
        # biases in front of a BatchNorm have a mathematically zero gradient: compare those absolutely
        if float(ref_g.norm()) < 1e-5:
            assert float(p.grad.norm()) < 1e-4, k
            continue
        e = relerr(p.grad.cpu(), ref_g)
        if e > worst[0]:
            worst = (e, k)
    assert worst[0] < 1e-4, worst          # per-tensor gradient relative L2 1e-4
    gn = float(g.raw['gradnorm'])
    assert abs(gsq ** 0.5 - gn) <= 1e-5 * gn
    for k, v in g.group('bnpost').items():
        torch.testing.assert_close(m.state_dict()[k].cpu(), v, rtol=1e-4, atol=1e-5)


def test_state_dict_keys_and_seeded_init_match_reference():
    """cfg1 was generated from torch.manual_seed(42) + the reference constructor; ours must reproduce it."""
    import lvae_amd  # noqa: F401
    from lvae_amd.models.lvae import LadderVAE
    g = load_golden('tiny_mnist')
    m = LadderVAE(**g.cfg)
    assert list(m.state_dict().keys()) == list(g.state_dict().keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(g.state_dict()[k].shape), k


def test_trainstep_graph_and_side_stream_match_eager():
    """hipGraph replay + wgrad on a side stream must give the same parameters as plain eager steps (same Philox noise)."""
    import lvae_amd  # noqa: F401
    from lvae_amd.models.lvae import LadderVAE
    from lvae_amd.noise import PhiloxNoise
    from lvae_amd.optim import Adamax
    from lvae_amd.engine import TrainStep
    g = load_golden('tiny_cifar')
    xs = [torch.rand(4, 3, 32, 32, generator=torch.Generator().manual_seed(i)).cuda() for i in range(5)]
    results = []
    for mode in ('eager', 'graph'):
        m = LadderVAE(**g.cfg)
        m.load_state_dict(g.state_dict())
        m.cuda().train()
        m.noise = PhiloxNoise(seed=7)
        opt = Adamax(m, lr=1e-3)
        step = TrainStep(m, opt, use_graph=(mode == 'graph'), async_wgrad=(mode == 'graph'), eager_warmup=2)
        losses = [float(step(x)['loss']) for x in xs]
        torch.cuda.synchronize()
        results.append((losses, m.arena.params.clone(), {k: v.clone() for k, v in m.state_dict().items() if 'running' in k or 'tracked' in k}))
    (l0, p0, b0), (l1, p1, b1) = results
    assert max(abs(a - b) / abs(a) for a, b in zip(l0, l1)) < 1e-5, (l0, l1)
    assert float((p0 - p1).abs().max()) < 1e-5
    for k in b0:
        torch.testing.assert_close(b0[k].float(), b1[k].float(), rtol=1e-5, atol=1e-6)


def test_exception_inside_a_captured_backward_leaves_a_clean_state(monkeypatch):
    """an exception in the middle of the CAPTURED backward (here: a kernel wrapper that refuses while the stream is capturing)
    must leave nothing behind that a later step trips over — no queued weight gradients launched on behalf of the abandoned pass, no
    transformed-weight buffer marked current for a launch that never ran, no half-built graph. The same TrainStep then captures again
    and ends up with exactly the parameters of a run that never failed."""
    import lvae_amd  # noqa: F401
    from lvae_amd import kernels as K
    from lvae_amd import ops
    from lvae_amd.models.lvae import LadderVAE
    from lvae_amd.noise import PhiloxNoise
    from lvae_amd.optim import Adamax
    from lvae_amd.engine import TrainStep
    g = load_golden('tiny_cifar')
    xs = [torch.rand(4, 3, 32, 32, generator=torch.Generator().manual_seed(i)).cuda() for i in range(5)]

    def fresh():
        m = LadderVAE(**g.cfg)
        m.load_state_dict(g.state_dict())
        m.cuda().train()
        m.noise = PhiloxNoise(seed=7)
        return m, TrainStep(m, Adamax(m, lr=1e-3), use_graph=True, eager_warmup=2)

    m0, s0 = fresh()
    for x in xs:
        s0(x)
    torch.cuda.synchronize()
    ref = m0.arena.params.clone()

    m1, s1 = fresh()
    s1(xs[0])
    s1(xs[1])
    real = K.affine_act_bwd_parts
    real_plain = K.affine_act_bwd

    def refusing(*a, **kw):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('simulated failure in the middle of the captured backward')
        return real(*a, **kw)

    def refusing_plain(*a, **kw):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('simulated failure in the middle of the captured backward')
        return real_plain(*a, **kw)

    monkeypatch.setattr(K, 'affine_act_bwd_parts', refusing)
    monkeypatch.setattr(K, 'affine_act_bwd', refusing_plain)
    with pytest.raises(RuntimeError, match='simulated failure'):
        s1(xs[2])
    monkeypatch.setattr(K, 'affine_act_bwd_parts', real)
    monkeypatch.setattr(K, 'affine_act_bwd', real_plain)
    torch.cuda.synchronize()
    assert s1.graph_a is None and not ops._side.get('group_q')
    assert all(e['stamp'] is None for e in K.prepared.entries.values())
    assert m1.global_step == 2     # the failed call's step count and BatchNorm forward counts were taken back
    for x in xs[2:]:
        s1(x)                      # captures again, then replays
    torch.cuda.synchronize()
    assert s1.graph_a is not None
    assert float((m1.arena.params - ref).abs().max()) < 1e-6
    assert m1.global_step == m0.global_step == 5
    sd0, sd1 = m0.state_dict(), m1.state_dict()
    nbt = [k for k in sd0 if k.endswith('num_batches_tracked')]
    assert nbt and all(int(sd0[k]) == int(sd1[k]) == 5 for k in nbt)


def test_cfg1_mnist3_batch64_matches_reference():
    """BASELINE configs[0]: static-MNIST-shaped 3-layer LVAE, batch 64; weights rebuilt from the reference's seed."""
    import lvae_amd  # noqa: F401
    from lvae_amd.models.lvae import LadderVAE
    from lvae_amd.noise import TapeNoise
    from lvae_amd.engine import forward_pass
    g = load_golden('cfg1_mnist3')
    torch.manual_seed(int(g.raw['init_seed']))
    m = LadderVAE(**g.cfg).cuda().train()
    m.noise = TapeNoise(g.seq('tape'))
    out = forward_pass(m, g.t('x').cuda())
    assert m.noise.exhausted()
    fp = g.group('fp')
    for k in ('loss', 'elbo', 'recons', 'l2'):
        a, b = float(out[k]), float(fp[k])
        assert abs(a - b) <= 1e-5 * abs(b), (k, a, b)   # relative 1e-5 (BASELINE: ELBO within 1e-3)
    torch.testing.assert_close(out['elbo_sep'].cpu(), fp['elbo_sep'], rtol=1e-5, atol=1e-4)
    m.zero_grad()
    out['loss'].backward()
    named = dict(m.named_parameters())
    worst = 0.0
    for k, ref in g.group('grad').items():
        if float(ref.norm()) < 1e-5:
            continue
        worst = max(worst, relerr(named[k].grad.cpu(), ref))
    assert worst < 1e-4, worst
    gsq = sum(float(p.grad.double().pow(2).sum()) for p in m.parameters() if p.grad is not None)
    assert abs(gsq ** 0.5 - float(g.raw['gradnorm'])) <= 1e-5 * float(g.raw['gradnorm'])


def test_sample_prior_matches_reference():
    g = load_golden('tiny_prior')
    m, TapeNoise = build(g, training=False)
    for tag, ml, cl in (('a', None, None), ('b', [0, 1], [2]), ('c', [0], [1, 2])):
        m.noise = TapeNoise(g.seq('prior_%s.tape' % tag))
        with torch.no_grad():
            s = m.sample_prior(3, ml, cl)
        assert m.noise.exhausted()
        assert tuple(s.shape) == (3, 3, 16, 16)
        torch.testing.assert_close(s.cpu(), g.t('prior_%s.sample' % tag), rtol=1e-4, atol=2e-4)


def test_iw_log_likelihood_with_bottom_up_reuse_matches_oracle():
    """IW bound: the engine runs bottom-up once and replays top-down S times; the oracle does S full forwards."""
    from oracle import lvae_ref as R
    from lvae_amd.evaluate import iw_log_likelihood, inspect_layer_repr
    g = load_golden('tiny_cifar')
    S = 6
    x = g.t('x')
    tape = R.Tape(gen=torch.Generator().manual_seed(3))
    iw_ref, elbo_ref = R.iw_log_likelihood(g.state_dict(), g.cfg, x, tape, S)
    m, TapeNoise = build(g, training=True)   # iw_log_likelihood switches to eval itself and restores the mode
    m.noise = TapeNoise(tape.entries)
    iw, elbo = iw_log_likelihood(m, x.cuda(), S)
    assert m.noise.exhausted() and m.training
    torch.testing.assert_close(iw.cpu(), iw_ref, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(elbo.cpu(), elbo_ref, rtol=1e-5, atol=1e-4)
    assert float((iw.cpu() - elbo.cpu()).min()) >= -1e-3   # Jensen: the IW bound is at least the mean ELBO


def test_inspect_layer_repr_matches_oracle():
    """evaluate.py:95-114: per layer, n rows of n images (n sample_prior calls with the layers below at their mode and the layers above
    constant within a call), against the oracle's restatement on the same noise tape."""
    from oracle import lvae_ref as R
    from lvae_amd.evaluate import inspect_layer_repr
    g = load_golden('tiny_cifar')
    n = 3
    tape = R.Tape(gen=torch.Generator().manual_seed(8))
    ref = R.inspect_layer_repr(g.state_dict(), g.cfg, tape, n)
    m, TapeNoise = build(g, training=True)   # switches to eval itself and restores the mode
    m.noise = TapeNoise(tape.entries)
    reps = inspect_layer_repr(m, n)
    assert m.noise.exhausted() and m.training
    assert len(reps) == m.n_layers == len(ref)
    for a, b in zip(reps, ref):
        assert tuple(a.shape) == (n * n, 3, 32, 32) == tuple(b.shape)
        torch.testing.assert_close(a.cpu(), b, rtol=1e-4, atol=2e-4)
    # within a row the constant layers were drawn once: with every layer but the top one at its mode (i = L - 1 has no constant
    # layer) rows differ; for i = 0 all layers above are constant per row, so images of a row share everything but layer 0
    top = reps[0].view(n, n, *reps[0].shape[1:])
    assert float((top[0] - top[1]).abs().max()) > 1e-3


def test_checkpoint_roundtrip_and_cli(tmp_path):
    """Reference-layout checkpoints: contiguous tensors with the reference's keys; weights + Adamax state survive a
    save/load round trip bit for bit; the training and evaluation CLIs run end to end on synthetic data."""
    import lvae_amd  # noqa: F401
    from lvae_amd import main as lmain, evaluate as leval
    from lvae_amd.checkpoint import save_checkpoint, load_checkpoint
    from lvae_amd.models.lvae import LadderVAE
    from lvae_amd.optim import Adamax
    from lvae_amd.engine import TrainStep
    g = load_golden('tiny_cifar')
    m = LadderVAE(**g.cfg)
    m.load_state_dict(g.state_dict())
    m.cuda().train()
    opt = Adamax(m, lr=1e-3)
    step = TrainStep(m, opt, use_graph=False)
    x = torch.rand(4, 3, 32, 32).cuda()
    for _ in range(3):
        step(x)
    path = str(tmp_path / 'ck.pt')
    save_checkpoint(path, m, opt)
    ck = torch.load(path)
    assert list(ck['model'].keys()) == list(g.state_dict().keys())
    assert all(v.is_contiguous() and v.device.type == 'cpu' for v in ck['model'].values())
    assert ck['optimizer']['step'] == 3 and ck['global_step'] == 3
    m2 = LadderVAE(**g.cfg).cuda().train()
    opt2 = Adamax(m2, lr=1e-3)
    load_checkpoint(path, m2, opt2)
    assert torch.equal(m2.arena.params, m.arena.params)
    from lvae_amd.checkpoint import optimizer_state_by_name
    s1, s2 = optimizer_state_by_name(m, opt)['state'], optimizer_state_by_name(m2, opt2)['state']
    assert s1.keys() == s2.keys() and len(s1) > 100
    for k in s1:  # per-parameter state (alignment padding between slots is not part of the state)
        assert torch.equal(s1[k]['exp_avg'], s2[k]['exp_avg']) and torch.equal(s1[k]['exp_inf'], s2[k]['exp_inf']), k
    assert int(opt2.step_count.item()) == 3 and m2.global_step == 3
    # CLIs (reference flag spellings)
    argv = ['-d', 'cifar10', '--zdims', '8', '8', '--downsample', '1', '1', '--nfilters', '16', '--skip', '--gated',
            '--freebits', '1.0', '--batch-size', '8', '--synthetic', '--seed', '3']
    ck2 = str(tmp_path / 'ck2.pt')
    lmain.main(argv + ['--steps', '5', '--log-every', '5', '--save-checkpoint', ck2])
    leval.main(argv + ['--ll', '--ll-samples', '4', '--n-test', '16', '--test-batch-size', '8', '--checkpoint', ck2])


def test_large_batch_step_through_winograd_paths_matches_oracle():
    """The golden vectors are tiny models whose layers stay below the Winograd / grouping thresholds. This step is big enough
    (64 filters, batch 192, 16x16 and 8x8 levels) for the Winograd forward/dgrad/wgrad kernels, the prepared-weight cache, the
    direct 1x1 weight gradient and the grouped weight gradients to run inside a real training step; loss, ELBO terms and every
    parameter gradient are compared with the CPU oracle on the same weights, input and noise tape."""
    import lvae_amd  # noqa: F401
    from lvae_amd import kernels as K
    from lvae_amd.models.lvae import LadderVAE
    from lvae_amd.noise import TapeNoise
    from lvae_amd.optim import Adamax
    from lvae_amd.engine import TrainStep
    from oracle import lvae_ref as R
    cfg = dict(color_ch=3, z_dims=[8, 8], blocks_per_layer=1, downsample=[0, 1], nonlin='elu', merge_type='residual',
               batchnorm=True, stochastic_skip=True, n_filters=64, dropout=0.2, free_bits=1.0, learn_top_prior=True,
               img_shape=(32, 32), likelihood_form='discr_log_mix', res_block_type='bacdbacd', gated=True,
               no_initial_downscaling=False, analytical_kl=False)
    torch.manual_seed(11)
    model = LadderVAE(**cfg)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = torch.floor(256 * torch.rand(192, 3, 32, 32)) / 255
    tape = R.Tape(gen=torch.Generator().manual_seed(6))
    pkeys = [k for k in sd if R.is_parameter_key(k)]
    for k in pkeys:
        sd[k].requires_grad_(True)
    fp, _ = R.forward_pass(sd, cfg, x, tape, param_keys=pkeys)
    fp['loss'].backward()
    model.cuda().train()
    model.noise = TapeNoise(tape.entries)
    opt = Adamax(model, lr=0.0)            # lr 0: the step leaves the weights (and the comparison) untouched
    K.prepared.entries.clear()
    K.prepared.table = None
    step = TrainStep(model, opt, use_graph=False)
    step(x.cuda())                          # registers the Winograd call sites (transforms their weights per launch)
    model.noise = TapeNoise(tape.entries)
    out = step(x.cuda())                    # second step: prepared (batched) weight transforms, same tape
    torch.cuda.synchronize()
    for k in ('loss', 'elbo', 'recons', 'kl'):
        a, b = float(out[k]), float(fp[k])
        assert abs(a - b) <= 1e-5 * abs(b), (k, a, b)
    worst = 0.0
    for k, p in model.named_parameters():
        ref = sd[k].grad
        if ref is None or float(ref.norm()) < 1e-5:
            continue
        worst = max(worst, float((p.grad.cpu() - ref).norm() / ref.norm()))
    assert worst < 1e-4, worst
    K.prepared.entries.clear()
    K.prepared.table = None


def test_data_dependent_init_is_a_fixed_point_after_one_pass():
    """--data-dep-init (parity unpinned: boilr is absent): after one pass every convolution's output on the init batch has zero
    mean / unit std per channel, so a second pass over the same batch with the same noise must leave the parameters where they
    are, while the first pass must have moved them."""
    import lvae_amd  # noqa: F401
    from lvae_amd.init import data_dependent_init
    from lvae_amd.models.lvae import LadderVAE
    from lvae_amd.noise import PhiloxNoise
    cfg = dict(color_ch=3, z_dims=[8, 8], blocks_per_layer=1, downsample=[1, 1], nonlin='elu', merge_type='residual',
               batchnorm=True, stochastic_skip=True, n_filters=16, dropout=0.0, free_bits=0.5, learn_top_prior=True,
               img_shape=(32, 32), likelihood_form='discr_log_mix', res_block_type='bacdbacd', gated=True,
               no_initial_downscaling=False, analytical_kl=False)
    torch.manual_seed(5)
    model = LadderVAE(**cfg).cuda()
    model.pack()
    x = (torch.floor(256 * torch.rand(32, 3, 32, 32)) / 255).cuda()
    convs = {k: v for k, v in model.named_parameters() if v.dim() == 4 and k.endswith('.weight')}
    before = {k: v.detach().clone() for k, v in convs.items()}
    model.noise = PhiloxNoise(seed=9)
    n = data_dependent_init(model, x)
    assert n >= len(convs) - 1 and n > 10
    after1 = {k: v.detach().clone() for k, v in convs.items()}
    moved = sum(float((after1[k] - before[k]).norm() / before[k].norm()) > 1e-3 for k in convs)
    assert moved >= len(convs) // 2
    model.noise = PhiloxNoise(seed=9)
    data_dependent_init(model, x)
    for k, v in convs.items():
        assert float((v.detach() - after1[k]).norm() / after1[k].norm()) < 2e-3, k
    out = model(x)                           # the product path still runs after the in-place updates
    assert torch.isfinite(out['ll']).all()



def test_data_dependent_init_normalises_every_convolution_output():
    """--data-dep-init (parity unpinned: boilr absent, no fixture can exist offline). Property of the algorithm as the reference's call
    site describes it (experiment/experiment_manager.py:61-72 -> boilr.nn.init.data_dependent_init): after the pass, on the SAME batch
    and noise, the output of every convolution — visited in execution order, each seeing the corrected outputs of its predecessors —
    has per-channel mean 0 and standard deviation 1. Checked in float64 on the tensors the HIP convolutions really produce."""
    import lvae_amd  # noqa: F401
    from lvae_amd import kernels as K
    from lvae_amd.init import data_dependent_init
    from lvae_amd.models.lvae import LadderVAE
    from lvae_amd.noise import PhiloxNoise
    cfg = dict(color_ch=3, z_dims=[8, 8, 8], blocks_per_layer=2, downsample=[0, 1, 1], nonlin='elu', merge_type='residual',
               batchnorm=True, stochastic_skip=True, n_filters=32, dropout=0.0, free_bits=0.5, learn_top_prior=True,
               img_shape=(32, 32), likelihood_form='discr_log_mix', res_block_type='bacdbacd', gated=True,
               no_initial_downscaling=False, analytical_kl=False)
    torch.manual_seed(6)
    model = LadderVAE(**cfg).cuda().train()
    model.pack()
    x = (torch.floor(256 * torch.rand(64, 3, 32, 32)) / 255).cuda()
    model.noise = PhiloxNoise(seed=4)
    n = data_dependent_init(model, x)
    seen = []
    orig_conv2d, orig_gate = K.conv2d, K.conv1x1_gate

    def note(y):
        y64 = y.detach().double().reshape(-1, y.shape[-1])
        seen.append((float(y64.mean(0).abs().max()), float((y64.std(0, unbiased=False) - 1).abs().max())))

    def conv2d(x_, weight, g, **kw):
        if kw.get('out_scale') is None and kw.get('out_act') is None:
            r = orig_conv2d(x_, weight, g, **kw)
            note(r[0] if isinstance(r, tuple) else r)
            return r
        plain = {k: v for k, v in kw.items() if k in ('bias', 'x2', 'in_scale', 'in_shift', 'in_act')}
        if 'in_bn' not in kw:   # (a folded BatchNorm finalize has no plain twin; its output is checked through the epilogue-free calls)
            note(orig_conv2d(x_, weight, g, **plain))
        return orig_conv2d(x_, weight, g, **kw)

    def gate(x_, weight, g, bias, res, act, **kw):
        r = orig_gate(x_, weight, g, bias, res, act, **kw)
        if r[0] is not None:
            note(r[0])            # ab = the gate convolution's output
        return r

    K.conv2d, K.conv1x1_gate = conv2d, gate
    try:
        model.noise = PhiloxNoise(seed=4)
        with torch.no_grad():
            model(x)
    finally:
        K.conv2d, K.conv1x1_gate = orig_conv2d, orig_gate
    assert len(seen) >= n - 2 and n > 20, (len(seen), n)
    worst_mean = max(m for m, _ in seen)
    worst_std = max(sd for _, sd in seen)
    assert worst_mean < 1e-4, worst_mean
    assert worst_std < 1e-3, worst_std


def test_iw_log_likelihood_graph_replay_matches_eager_loop():
    """evaluate.py:30,56-66 (S-sample IW bound): the captured top-down + likelihood sample replayed S times must give the bound the
    eager loop gives with the same Philox stream; S = 200 exercises the online log-sum-exp state."""
    from lvae_amd.evaluate import iw_log_likelihood
    from lvae_amd.noise import PhiloxNoise
    g = load_golden('tiny_cifar')
    m, _ = build(g, training=False)
    x = g.t('x').cuda()
    res = []
    for use_graph in (False, True):
        m.noise = PhiloxNoise(seed=11)
        res.append(iw_log_likelihood(m, x, 200, use_graph=use_graph))
    (iw0, e0), (iw1, e1) = res
    torch.testing.assert_close(iw1, iw0, rtol=1e-6, atol=1e-3)
    torch.testing.assert_close(e1, e0, rtol=1e-6, atol=1e-3)
    assert float((iw0 - e0).min()) >= -1e-3            # Jensen
    assert float((iw0 - e0).max()) > 0.1               # ... and the samples really differ from replay to replay
