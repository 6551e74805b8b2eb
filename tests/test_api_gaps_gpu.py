"""Reference API surface that round 1 left untested or unbuilt: `kl_elementwise` / `kl_normal_mc` (lib/stochastic.py:88-112,
209-226), the `forced_latent` path (lib/stochastic.py:66-67, models/lvae.py:234,296), fresh noise on repeated
`topdown_pass` calls, Adamax with weight decay (experiment/experiment_manager.py:78-80), the prepared-weight cache's
life time. Vectors: tests/golden/stoch.npz and tiny_forced.npz, captured from the reference by oracle/gen_golden.py."""
import gc

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize('tag', ['mc', 'an', 'forced', 'mode'])
def test_stochastic_block_all_keys_and_kl_elementwise_gradients(tag):
    import lvae_amd  # noqa: F401
    from lvae_amd.lib.stochastic import NormalStochasticBlock2d
    from lvae_amd.noise import TapeNoise
    g = load_golden('stoch')
    blk = NormalStochasticBlock2d(c_in=8, c_vars=4, c_out=8)
    blk.load_state_dict(g.state_dict())
    blk.cuda()
    p_in = nhwc(g.t('p_in')).requires_grad_(True)
    q_in = nhwc(g.t('q_in')).requires_grad_(True)
    kw = {}
    if tag == 'an':
        kw['analytical_kl'] = True
    elif tag == 'forced':
        kw['forced_latent'] = nhwc(g.t('forced'))
    elif tag == 'mode':
        kw['use_mode'] = True
    noise = TapeNoise(g.seq(tag + '.tape'))
    out, data = blk(p_in, q_in, noise=noise, **kw)
    assert noise.exhausted()
    tol = dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(nchw(out.detach()), g.t(tag + '.out'), **tol)
    for k in ('z', 'p_params', 'q_params', 'kl_elementwise'):
        torch.testing.assert_close(nchw(data[k].detach()), g.t('%s.data.%s' % (tag, k)), **tol)
    for k in ('logprob_p', 'logprob_q', 'kl_samplewise', 'kl_spatial'):
        torch.testing.assert_close(data[k].detach().cpu(), g.t('%s.data.%s' % (tag, k)), rtol=1e-4, atol=1e-3)
    # gradients through kl_elementwise (weighted) and the block output, as the golden script formed them
    blk.zero_grad()
    ((data['kl_elementwise'] * nhwc(g.t('w_el'))).sum() + 0.1 * out.sum()).backward()
    torch.testing.assert_close(nchw(p_in.grad), g.t(tag + '.dp_in'), rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(nchw(q_in.grad), g.t(tag + '.dq_in'), rtol=2e-4, atol=2e-4)
    for k, p in blk.named_parameters():
        ref = g.raw.get('%s.grad.%s' % (tag, k))
        if ref is not None:
            torch.testing.assert_close(p.grad.cpu(), torch.from_numpy(ref), rtol=2e-4, atol=2e-4)


def test_kl_normal_mc_elementwise_with_broadcast_prior():
    import lvae_amd  # noqa: F401
    from lvae_amd.lib.stochastic import kl_normal_mc
    g = load_golden('stoch')
    out = kl_normal_mc(nhwc(g.t('klmc.z')), nhwc(g.t('klmc.p')), nhwc(g.t('klmc.q')))
    assert tuple(out.shape) == (3, 4, 4, 4)
    torch.testing.assert_close(nchw(out), g.t('klmc.out'), rtol=1e-5, atol=1e-5)


def test_topdown_pass_with_forced_latents_matches_reference():
    import lvae_amd  # noqa: F401
    from lvae_amd.models.lvae import LadderVAE
    from lvae_amd.noise import TapeNoise
    g = load_golden('tiny_forced')
    m = LadderVAE(**g.cfg)
    m.load_state_dict(g.state_dict(), strict=True)
    m.cuda().eval()
    m.noise = TapeNoise(g.seq('tape'))
    forced = [g.raw.get('forced.%d' % i) for i in range(3)]
    forced = [None if f is None else torch.from_numpy(f).cuda() for f in forced]
    with torch.no_grad():
        out, data = m.topdown_pass([b.cuda() for b in g.seq('bu')], forced_latent=forced)
    assert m.noise.exhausted()
    torch.testing.assert_close(out.cpu(), g.t('out'), rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(data['logprob_p'].cpu(), g.t('data.logprob_p'), rtol=2e-5, atol=2e-3)
    for i in range(3):
        torch.testing.assert_close(data['z'][i].cpu(), g.t('data.z.%d' % i), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(data['kl'][i].cpu(), g.t('data.kl.%d' % i), rtol=2e-5, atol=2e-3)
        torch.testing.assert_close(data['kl_spatial'][i].cpu(), g.t('data.kl_spatial.%d' % i), rtol=1e-4, atol=1e-3)
    assert torch.equal(data['z'][0].cpu(), forced[0].cpu())       # a forced latent is passed through untouched
    # and the reference's own bottom-up values are reproduced by bottomup_pass
    with torch.no_grad():
        bu = m.bottomup_pass(m.pad_input(g.t('x').cuda()))
    for a, b in zip(bu, g.seq('bu')):
        torch.testing.assert_close(a.cpu(), b, rtol=1e-4, atol=1e-4)


def test_repeated_passes_draw_fresh_noise():
    """topdown_pass / bottomup_pass must advance the Philox step — two calls return different samples."""
    import lvae_amd  # noqa: F401
    from lvae_amd.models.lvae import LadderVAE
    from lvae_amd.noise import PhiloxNoise
    g = load_golden('tiny_cifar')
    m = LadderVAE(**g.cfg)
    m.load_state_dict(g.state_dict())
    m.cuda().train()
    m.noise = PhiloxNoise(seed=3)
    x = m.pad_input(g.t('x').cuda())
    with torch.no_grad():
        bu1 = m.bottomup_pass(x)
        bu2 = m.bottomup_pass(x)
        assert not torch.equal(bu1[0], bu2[0])                  # Dropout2d masks differ between calls
        _, d1 = m.topdown_pass(bu1)
        _, d2 = m.topdown_pass(bu1)
    for z1, z2 in zip(d1['z'], d2['z']):
        assert float((z1 - z2).abs().max()) > 1e-3
    m.eval()
    with torch.no_grad():
        s1, s2 = m.sample_prior(2), m.sample_prior(2)
    assert not torch.equal(s1, s2)


@pytest.mark.parametrize('wd', [0.0, 1e-2])
def test_adamax_weight_decay_matches_torch(wd):
    import lvae_amd  # noqa: F401
    from lvae_amd import kernels as K
    g = torch.Generator().manual_seed(10)
    n = 4100
    p0, grads = torch.randn(n, generator=g), [torch.randn(n, generator=g) for _ in range(4)]
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adamax([pr], lr=2e-3, weight_decay=wd)
    pd, m, u = p0.cuda(), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    step = torch.zeros(1, dtype=torch.int64, device='cuda')
    half = torch.full((1,), 0.5, device='cuda')
    for gr in grads:
        pr.grad = gr.clone()
        opt.step()
        # the data-parallel path hands over SUMMED gradients and a 1/world scale: 2 ranks with the same gradient
        K.adamax_step(pd, (2 * gr).cuda(), m, u, None, 2e-3, 0.9, 0.999, 1e-8, wd, half, step)
        K.counter_advance(step)
    torch.testing.assert_close(pd.cpu(), pr.detach(), rtol=2e-6, atol=2e-7)


def test_prepared_weight_entries_die_with_their_model():
    """the transformed-weight cache must not pin a dropped model's arena nor keep transforming it."""
    import lvae_amd  # noqa: F401
    from lvae_amd import kernels as K
    K.prepared.entries.clear()
    K.prepared.table = None
    w = (torch.randn(64, 64, 3, 3) / 24).permute(2, 3, 1, 0).contiguous().cuda().permute(3, 2, 0, 1)
    geom = K.ConvGeom(w, 1, 1)
    x = torch.randn(200, 16, 16, 64, device='cuda')
    K.conv2d(x, w, geom)
    assert len(K.prepared.entries) == 1 and K.prepared.prepare_all() == 1
    del w, geom
    gc.collect()
    assert K.prepared.prepare_all() == 0 and len(K.prepared.entries) == 0
