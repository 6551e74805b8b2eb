"""Host-side logic of the engine that needs no GPU: module tree / state_dict scheme, seeded init, argument checks."""
import pytest
import torch

from conftest import load_golden


def make(**over):
    import lvae_amd  # noqa: F401
    from lvae_amd.models.lvae import LadderVAE
    g = load_golden('tiny_mnist')
    cfg = dict(g.cfg)
    cfg.update(over)
    return LadderVAE(**cfg)


@pytest.mark.parametrize('name', ['tiny_mnist', 'tiny_cifar', 'tiny_cabdcabd', 'tiny_bacdbac', 'tiny_nobn_selu', 'tiny_prior'])
def test_state_dict_scheme_matches_reference(name):
    import lvae_amd  # noqa: F401
    from lvae_amd.models.lvae import LadderVAE
    g = load_golden(name)
    m = LadderVAE(**g.cfg)
    ref = g.state_dict()
    assert list(m.state_dict().keys()) == list(ref.keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(ref[k].shape), k
    m.load_state_dict(ref)


def test_seeded_default_init_reproduces_reference():
    """cfg1 golden: the reference built under torch.manual_seed(42); fp.l2 is the L2 norm of ALL its initial parameters."""
    import lvae_amd  # noqa: F401
    from lvae_amd.models.lvae import LadderVAE
    g = load_golden('cfg1_mnist3')
    torch.manual_seed(int(g.raw['init_seed']))
    m = LadderVAE(**g.cfg)
    assert sum(p.numel() for p in m.parameters()) == 2055105
    l2 = torch.sqrt(sum((p.double() ** 2).sum() for p in m.parameters()))
    assert abs(float(l2) - float(g.raw['fp.l2'])) < 1e-4 * float(l2)
    # spot-check tensors through the Adamax post-step values: post = init - lr*sign-ish step, |post - init| <= lr
    named = dict(m.named_parameters())
    for k, post in g.group('post').items():
        assert float((named[k].detach() - post).abs().max()) <= 3.001e-4, k  # one Adamax step moves every weight by at most lr (fp32 rounding at |w| ~ 1)


def test_geometry_helpers_and_argument_checks():
    m = make()
    assert m.n_layers == 2 and m.overall_downscale_factor == 8
    assert m.get_padded_size((28, 28)) == [32, 32] and m.get_padded_size((3, 1, 28, 28)) == [32, 32]
    assert m.get_top_prior_param_shape() == (1, 16, 4, 4)
    with pytest.raises(RuntimeError):
        m.get_padded_size((1, 2, 3))
    with pytest.raises(RuntimeError):
        make(likelihood_form='nope')
    with pytest.raises(KeyError):
        make(nonlin='gelu')
    with pytest.raises(AssertionError):
        make(downsample=[3, 0])  # more downsampling steps than blocks per layer
    with pytest.raises(TypeError):
        make(dropout=None)  # reference: nn.Dropout2d(None) in 'bacdbacd'
    with pytest.raises(ValueError):
        make(res_block_type='abcd')


def test_noise_tape_layout_conversion():
    import lvae_amd  # noqa: F401
    from lvae_amd.noise import TapeNoise
    keep = (torch.rand(3, 5, 1, 1) < 0.8).float()
    eps = torch.randn(3, 4, 2, 2)
    t = TapeNoise([keep, eps])
    m = t.dropout_mask(3, 5, 0.2, torch.device('cpu'))
    assert torch.allclose(m, keep.view(3, 5) / 0.8)
    e = t.normal((3, 2, 2, 4), torch.device('cpu'))
    assert torch.equal(e, eps.permute(0, 2, 3, 1)) and t.exhausted()
    with pytest.raises(RuntimeError):
        t.normal((1, 1, 1, 1), torch.device('cpu'))
