"""The five BASELINE.json configurations as LadderVAE keyword dicts (flag sets fixed in SURVEY.md §8d, built the way
experiment/experiment_manager.py:40-59 of the reference maps flags to constructor arguments). bench.py, the full-size
parity tests and the CLI tests all build their models from here, so what is timed is what is checked."""


def _cfg(color_ch, img, likelihood, downsample, bpl, free_bits, learn_top_prior=True):
    return dict(color_ch=color_ch, z_dims=[32] * len(downsample), blocks_per_layer=bpl, downsample=list(downsample),
                nonlin='elu', merge_type='residual', batchnorm=True, stochastic_skip=True, n_filters=64, dropout=0.2,
                free_bits=free_bits, learn_top_prior=learn_top_prior, img_shape=(img, img), likelihood_form=likelihood,
                res_block_type='bacdbacd', gated=True, no_initial_downscaling=False, analytical_kl=False)


# configs[0]: static_mnist 3-layer, --zdims 32 32 32 --downsample 1 1 1 --skip --gated --freebits 0.5 (batch 64)
MNIST3 = _cfg(1, 28, 'bernoulli', [1, 1, 1], 2, 0.5, learn_top_prior=False)
# configs[1]: static_mnist 12-layer (batch 256)
MNIST12 = _cfg(1, 28, 'bernoulli', [0, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0], 4, 1.0)
# configs[2] / configs[3]: CIFAR10 15-layer (batch 256 per GPU)
CIFAR15 = _cfg(3, 32, 'discr_log_mix', [0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0], 4, 1.0)
# configs[4]: CelebA 64x64 20-layer (batch 128 per GPU)
CELEBA20 = _cfg(3, 64, 'discr_log_mix', [0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], 4, 1.0)

BY_NAME = {'mnist3': MNIST3, 'mnist12': MNIST12, 'cifar15': CIFAR15, 'celeba20': CELEBA20}


def synthetic_images(cfg, batch, gen):
    """Synthetic inputs of SURVEY.md §8d: binary pixels for the Bernoulli head, 8-bit levels in [0,1] otherwise."""
    import torch
    u = torch.rand((batch, cfg['color_ch']) + tuple(cfg['img_shape']), generator=gen)
    return (u > 0.5).float() if cfg['likelihood_form'] == 'bernoulli' else torch.floor(256 * u) / 255
