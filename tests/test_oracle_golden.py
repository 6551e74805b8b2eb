"""Pins the CPU oracle (oracle/lvae_ref.py) to vectors captured from the real reference (tests/golden)."""
import pytest
import torch

from conftest import load_golden
from oracle import lvae_ref as R

TINY = ['tiny_mnist', 'tiny_cifar', 'tiny_eval', 'tiny_cabdcabd', 'tiny_bacdbac', 'tiny_nobn_selu', 'tiny_gauss',
        'tiny_discrlog', 'tiny_prior']


def close(a, b, rtol=1e-5, atol=1e-5):
    torch.testing.assert_close(a.float(), b.float(), rtol=rtol, atol=atol)


@pytest.mark.parametrize('name', TINY)
def test_oracle_forward_backward_matches_reference(name):
    torch.set_num_threads(1)
    g = load_golden(name)
    cfg = g.cfg
    sd = g.state_dict()
    pkeys = [k for k in sd if R.is_parameter_key(k)]
    for k in pkeys:
        sd[k].requires_grad_(k.endswith('top_prior_params') is False or cfg['learn_top_prior'])
    training = name != 'tiny_eval'
    tape = R.Tape(g.seq('tape'))
    fp, mo = R.forward_pass(sd, cfg, g.t('x'), tape, training=training, param_keys=pkeys)
    assert tape.exhausted()
    out = g.group('out')
    close(mo['ll'], out['ll'], rtol=2e-6, atol=1e-4)
    close(mo['kl_sep'], out['kl_sep'], atol=1e-4)
    close(mo['kl'], out['kl'])
    close(mo['kl_loss'], out['kl_loss'])
    close(mo['kl_avg_layerwise'], out['kl_avg_layerwise'])
    close(mo['logp'], out['logp'], atol=1e-4)
    close(mo['out_sample'], out['out_sample'])
    for i, z in enumerate(mo['z']):
        close(z, out['z.%d' % i])
        close(mo['kl_spatial'][i], out['kl_spatial.%d' % i], atol=1e-4)
    if cfg['likelihood_form'] == 'bernoulli':
        close(mo['out_mean'], out['out_mean'])
        close(mo['out_mode'], out['out_mode'])
        close(mo['likelihood_params'], out['likelihood_params'])
    elif cfg['likelihood_form'] == 'discr_log_mix':
        assert mo['out_mean'] is None and mo['out_mode'] is None
        close(mo['likelihood_params']['all_params'], out['likelihood_params.all_params'], atol=1e-4)
    f = g.group('fp')
    for k in ('loss', 'elbo', 'recons', 'l2'):
        close(fp[k], f[k], rtol=2e-6, atol=1e-4)
    close(fp['elbo_sep'], f['elbo_sep'], rtol=2e-6, atol=1e-4)
    if not training:
        return
    fp['loss'].backward()
    grads = g.group('grad')
    gsq = 0.0
    for k in pkeys:
        if sd[k].grad is None:
            assert k not in grads
            continue
        gsq += float(sd[k].grad.double().pow(2).sum())
        ref = grads[k]
        err = (sd[k].grad - ref).norm() / (ref.norm() + 1e-12)
        assert err < 1e-4, (k, float(err))
    assert abs(gsq ** 0.5 - float(g.raw['gradnorm'])) <= 1e-4 * float(g.raw['gradnorm'])
    # running statistics after the training forward
    for k, v in g.group('bnpost').items():
        close(sd[k].detach(), v)
    # one Adamax step (experiment/experiment_manager.py:76-81)
    ks = [k for k in pkeys if sd[k].grad is not None]
    ps = [sd[k].detach().clone() for k in ks]
    # fed with the reference's own gradients: biases in front of a BatchNorm have |grad| ~ 1e-9 < eps, so a
    # step computed from re-derived gradients is dominated by rounding noise there
    R.adamax_step(ps, [grads[k] for k in ks], [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps], 1)
    post = g.group('post')
    for k, p in zip(ks, ps):
        close(p, post[k], rtol=1e-6, atol=1e-7)


def test_oracle_sample_prior_matches_reference():
    g = load_golden('tiny_prior')
    sd = g.state_dict()
    for tag, ml, cl in (('a', None, None), ('b', [0, 1], [2]), ('c', [0], [1, 2])):
        tape = R.Tape(g.seq('prior_%s.tape' % tag))
        with torch.no_grad():
            s = R.sample_prior(sd, g.cfg, 3, tape, ml, cl)
        assert tape.exhausted()
        close(s, g.t('prior_%s.sample' % tag))


def test_oracle_likelihood_edge_vectors():
    g = load_golden('ops')
    close(R.log_bernoulli(g.t('bern.x'), g.t('bern.mean')), g.t('bern.ll'))
    assert float(g.t('bern.ll').min()) < -190  # the saturated pixels really hit the -100 clamps
    l = g.t('dmol.l').clone().requires_grad_(True)
    ll = R.discretized_mix_logistic_ll(g.t('dmol.x') * 2 - 1, l)
    close(ll, g.t('dmol.ll'), rtol=1e-5, atol=1e-4)
    ll.sum().backward()
    close(l.grad, g.t('dmol.dl'), rtol=5e-4, atol=1e-4)
    s = R.sample_discretized_mix_logistic(g.t('dmol.l'), R.Tape(g.seq('dmol.tape')))
    close(s, g.t('dmol.sample'))
    close(R.log_discretized_logistic(g.t('dlog.x') * (255 / 256) + 1 / 512, g.t('dlog.mean'), g.t('dlog.ls')),
          g.t('dlog.ll'))


def test_padded_size_and_prior_shape_tables():
    cfg = dict(load_golden('tiny_mnist').cfg)
    for ds, noinit, img, want_pad, want_prior in (
            ([1, 1, 1], False, (28, 28), [32, 32], (1, 64, 2, 2)),
            ([0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0], False, (32, 32), [32, 32], (1, 64, 2, 2)),
            ([1, 0], True, (16, 16), [16, 16], (1, 64, 8, 8)),
            ([1, 1], False, (30, 45), [32, 48], (1, 64, 4, 6))):
        cfg.update(downsample=ds, z_dims=[32] * len(ds), no_initial_downscaling=noinit, img_shape=img)
        assert R.get_padded_size(cfg, img) == want_pad
        assert R.get_padded_size(cfg, (5, 1) + img) == want_pad
        assert R.get_top_prior_param_shape(cfg) == want_prior
    with pytest.raises(RuntimeError):
        R.get_padded_size(cfg, (1, 2, 3))


def test_oracle_forced_latent_topdown_matches_reference():
    """topdown_pass(bu_values, forced_latent=[z0, None, z2]) — models/lvae.py:229-315, lib/stochastic.py:66-67."""
    g = load_golden('tiny_forced')
    sd = g.state_dict()
    forced = [g.raw.get('forced.%d' % i) for i in range(3)]
    forced = [None if f is None else torch.from_numpy(f) for f in forced]
    tape = R.Tape(g.seq('tape'))
    with torch.no_grad():
        out, data = R.topdown_pass(sd, g.cfg, tape, False, bu_values=g.seq('bu'), forced_latent=forced)
    assert tape.exhausted()
    close(out, g.t('out'), atol=1e-4)
    close(data['logprob_p'], g.t('data.logprob_p'), atol=1e-3)
    for i in range(3):
        close(data['z'][i], g.t('data.z.%d' % i))
        close(data['kl'][i], g.t('data.kl.%d' % i), atol=1e-3)
        close(data['kl_spatial'][i], g.t('data.kl_spatial.%d' % i), atol=1e-4)
    assert torch.equal(data['z'][0], forced[0]) and torch.equal(data['z'][2], forced[2])


@pytest.mark.parametrize('tag', ['mc', 'an', 'forced', 'mode'])
def test_oracle_stochastic_block_matches_reference(tag):
    """Every key of NormalStochasticBlock2d's data dict (lib/stochastic.py:102-112) incl. kl_elementwise."""
    g = load_golden('stoch')
    sd = {'s.' + k: v for k, v in g.state_dict().items()}
    cfg = {'analytical_kl': tag == 'an'}
    tape = R.Tape(g.seq(tag + '.tape'))
    out, data = R.stochastic_block(sd, 's', g.t('p_in'), g.t('q_in'), cfg, True, tape,
                                   forced_latent=g.t('forced') if tag == 'forced' else None, use_mode=tag == 'mode')
    assert tape.exhausted()
    close(out, g.t(tag + '.out'))
    for k in ('z', 'p_params', 'q_params', 'logprob_p', 'logprob_q', 'kl_elementwise', 'kl_samplewise', 'kl_spatial'):
        close(data[k], g.t('%s.data.%s' % (tag, k)), atol=1e-4)


def test_oracle_kl_normal_mc_broadcast():
    g = load_golden('stoch')
    z, p, q = g.t('klmc.z'), g.t('klmc.p'), g.t('klmc.q')
    p_mu, p_lv = p.chunk(2, 1)
    q_mu, q_lv = q.chunk(2, 1)
    close(R.normal_log_prob(z, q_mu, q_lv) - R.normal_log_prob(z, p_mu, p_lv), g.t('klmc.out'))
