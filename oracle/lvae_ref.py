"""CPU oracle: a functional, plain-PyTorch fp32 restatement of the reference Ladder-VAE hot path.

TEST INFRASTRUCTURE ONLY. Nothing under `ladder-vae-pytorch_amd/` may import this file; it is used by
`tests/`, by `__graft_entry__.smoke()` as the checker and by `bench.py`'s `cpu_baseline` leg.

It is a pure function of (state_dict, cfg, x, noise tape): no nn.Module graph, no torch.distributions.
Every function cites the reference lines it restates (paths relative to /root/reference). It is pinned
by golden vectors that `oracle/gen_golden.py` captured from the real reference modules in the build
container (tests/golden/*.npz). The five `boilr` helpers the reference imports (pad/crop/Interpolate/
free_bits_kl/BaseGenerativeModel) are absent from the container: their semantics here are our restatement
from the call sites, so AT THAT BOUNDARY parity is unpinned.
"""
import math

import torch
import torch.nn.functional as F

LOG_SQRT_2PI = math.log(math.sqrt(2 * math.pi))


# --------------------------------------------------------------------------------------------------
# noise tape
# --------------------------------------------------------------------------------------------------
class Tape:
    """Ordered record of every RNG draw of one forward (SURVEY.md §8c "noise tape").

    replay mode (entries given): `draw` pops the next recorded tensor and checks its shape.
    record mode (entries None): `draw` samples from `gen` and appends, so the tape can be replayed
    by another implementation (the HIP path) afterwards.
    """

    def __init__(self, entries=None, gen=None):
        self.replay = entries is not None
        self.entries = list(entries) if entries is not None else []
        self.pos = 0
        self.gen = gen

    def draw(self, kind, shape, **kw):
        shape = tuple(int(s) for s in shape)
        if self.replay:
            if self.pos >= len(self.entries):
                raise RuntimeError("noise tape exhausted at draw #%d (%s %s)" % (self.pos, kind, shape))
            t = torch.as_tensor(self.entries[self.pos])
            self.pos += 1
            if tuple(t.shape) != shape:
                raise RuntimeError("noise tape entry %d has shape %s, expected %s (%s)" %
                                   (self.pos - 1, tuple(t.shape), shape, kind))
            return t.float()
        if kind == 'bernoulli':
            t = torch.empty(shape).bernoulli_(kw['p'], generator=self.gen)
        elif kind == 'normal':
            t = torch.empty(shape).normal_(generator=self.gen)
        elif kind == 'uniform':
            t = torch.empty(shape).uniform_(kw.get('lo', 0.0), kw.get('hi', 1.0), generator=self.gen)
        else:
            raise ValueError(kind)
        self.entries.append(t)
        return t

    def exhausted(self):
        return (not self.replay) or self.pos == len(self.entries)


# --------------------------------------------------------------------------------------------------
# lib/nn.py
# --------------------------------------------------------------------------------------------------
def _act(name, x):
    """Nonlinearity table of models/lvae.py:64-69 (module defaults: LeakyReLU slope 0.01, ELU alpha 1)."""
    if name == 'elu':
        return F.elu(x)
    if name == 'relu':
        return F.relu(x)
    if name == 'leakyrelu':
        return F.leaky_relu(x, 0.01)
    if name == 'selu':
        return F.selu(x)
    raise KeyError(name)


def block_layout(block_type, batchnorm, dropout, gated):
    """Order of sub-modules inside ResidualBlock.block (lib/nn.py:50-96) -> list of op kinds.

    The position in the list is the nn.Sequential index, i.e. the state_dict key `block.<idx>`.
    """
    ops = []
    if block_type == 'cabdcabd':
        for _ in range(2):
            ops += ['conv', 'act']
            if batchnorm:
                ops.append('bn')
            if dropout is not None:
                ops.append('drop')
    elif block_type == 'bacdbac':
        for i in range(2):
            if batchnorm:
                ops.append('bn')
            ops += ['act', 'conv']
            if dropout is not None and i == 0:
                ops.append('drop')
    elif block_type == 'bacdbacd':
        if dropout is None:
            # lib/nn.py:89 builds nn.Dropout2d(None) unconditionally, which raises in torch
            raise TypeError("residual block 'bacdbacd' requires a dropout probability")
        for _ in range(2):
            if batchnorm:
                ops.append('bn')
            ops += ['act', 'conv', 'drop']
    else:
        raise ValueError("unrecognized block type '{}'".format(block_type))
    if gated:
        ops.append('gate')
    return ops


def _dropout2d(x, p, tape, training):
    """nn.Dropout2d: one Bernoulli(1-p) draw per (sample, channel), result divided by (1-p)."""
    if not training or p == 0.0:
        return x
    keep = tape.draw('bernoulli', (x.shape[0], x.shape[1], 1, 1), p=1.0 - p)
    return x * (keep / (1.0 - p))


def _batchnorm(sd, key, x, training):
    """nn.BatchNorm2d defaults (momentum 0.1, eps 1e-5); running stats updated in place in `sd`."""
    out = F.batch_norm(x, sd[key + '.running_mean'], sd[key + '.running_var'], sd[key + '.weight'],
                       sd[key + '.bias'], training, 0.1, 1e-5)
    if training:
        sd[key + '.num_batches_tracked'] += 1
    return out


def gate_layer(sd, key, x, nonlin):
    """GateLayer2d.forward, lib/nn.py:121-126: 1x1 conv C->2C, act(first half) * sigmoid(second half)."""
    y = F.conv2d(x, sd[key + '.conv.weight'], sd[key + '.conv.bias'])
    a, b = y.chunk(2, dim=1)
    return _act(nonlin, a) * torch.sigmoid(b)


def residual_block(sd, prefix, x, cfg, gated, tape, training):
    """ResidualBlock.forward, lib/nn.py:98-99: block(x) + x with the recipes of lib/nn.py:50-96."""
    h = x
    for idx, op in enumerate(block_layout(cfg['res_block_type'], cfg['batchnorm'], cfg['dropout'], gated)):
        key = '%s.block.%d' % (prefix, idx)
        if op == 'conv':
            h = F.conv2d(h, sd[key + '.weight'], sd[key + '.bias'], padding=1)
        elif op == 'act':
            h = _act(cfg['nonlin'], h)
        elif op == 'bn':
            h = _batchnorm(sd, key, h, training)
        elif op == 'drop':
            h = _dropout2d(h, cfg['dropout'], tape, training)
        elif op == 'gate':
            h = gate_layer(sd, key, h, cfg['nonlin'])
    return h + x


# --------------------------------------------------------------------------------------------------
# models/lvae_layers.py
# --------------------------------------------------------------------------------------------------
def resampling_block(sd, prefix, x, cfg, mode, resample, gated, tape, training):
    """ResBlockWithResampling.forward, models/lvae_layers.py:300-306 (c_in == c_out everywhere)."""
    if resample:
        w, b = sd[prefix + '.pre_conv.weight'], sd[prefix + '.pre_conv.bias']
        if mode == 'bottom-up':  # models/lvae_layers.py:263-268
            x = F.conv2d(x, w, b, stride=2, padding=1)
        else:  # models/lvae_layers.py:270-276
            x = F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)
    return residual_block(sd, prefix + '.res', x, cfg, gated, tape, training)


def merge_layer(sd, prefix, x, y, cfg, merge_type, tape, training):
    """MergeLayer.forward, models/lvae_layers.py:358-360."""
    h = torch.cat((x, y), dim=1)
    if merge_type == 'linear':
        return F.conv2d(h, sd[prefix + '.layer.weight'], sd[prefix + '.layer.bias'])
    h = F.conv2d(h, sd[prefix + '.layer.0.weight'], sd[prefix + '.layer.0.bias'])
    return residual_block(sd, prefix + '.layer.1', h, cfg, True, tape, training)


def normal_log_prob(z, mu, lv):
    """torch.distributions.Normal(mu, exp(lv/2)).log_prob(z), as used at lib/stochastic.py:79,84,226."""
    std = (lv / 2).exp()
    return -((z - mu) ** 2) / (2 * std ** 2) - std.log() - LOG_SQRT_2PI


def normal_kl(q_mu, q_lv, p_mu, p_lv):
    """kl_divergence(Normal q, Normal p), lib/stochastic.py:87."""
    q_std, p_std = (q_lv / 2).exp(), (p_lv / 2).exp()
    var_ratio = (q_std / p_std) ** 2
    t1 = ((q_mu - p_mu) / p_std) ** 2
    return 0.5 * (var_ratio + t1 - 1 - var_ratio.log())


def stochastic_block(sd, prefix, p_params, q_params, cfg, transform_p, tape, forced_latent=None,
                     use_mode=False, force_constant_output=False):
    """NormalStochasticBlock2d.forward, lib/stochastic.py:29-112."""
    assert forced_latent is None or not use_mode
    if transform_p:
        p_params = F.conv2d(p_params, sd[prefix + '.conv_in_p.weight'], sd[prefix + '.conv_in_p.bias'],
                            padding=1)
    p_mu, p_lv = p_params.chunk(2, dim=1)
    if q_params is not None:
        q_params = F.conv2d(q_params, sd[prefix + '.conv_in_q.weight'], sd[prefix + '.conv_in_q.bias'],
                            padding=1)
        q_mu, q_lv = q_params.chunk(2, dim=1)
        s_mu, s_lv = q_mu, q_lv
    else:
        s_mu, s_lv = p_mu, p_lv
    if forced_latent is not None:
        z = forced_latent
    elif use_mode:
        z = s_mu
    else:
        eps = tape.draw('normal', s_mu.shape)
        z = s_mu + (s_lv / 2).exp() * eps
    if force_constant_output:  # lib/stochastic.py:71-73
        z = z[0:1].expand_as(z).clone()
        p_params = p_params[0:1].expand_as(p_params).clone()
    out = F.conv2d(z, sd[prefix + '.conv_out.weight'], sd[prefix + '.conv_out.bias'], padding=1)
    logprob_p = normal_log_prob(z, p_mu, p_lv).sum((1, 2, 3))
    data = {'z': z, 'p_params': p_params, 'q_params': q_params, 'logprob_p': logprob_p, 'logprob_q': None,
            'kl_elementwise': None, 'kl_samplewise': None, 'kl_spatial': None}
    if q_params is not None:
        data['logprob_q'] = normal_log_prob(z, q_mu, q_lv).sum((1, 2, 3))
        kl_analytical = normal_kl(q_mu, q_lv, p_mu, p_lv)
        if cfg['analytical_kl']:
            kl_elem = kl_analytical
        else:  # kl_normal_mc, lib/stochastic.py:209-226
            kl_elem = normal_log_prob(z, q_mu, q_lv) - normal_log_prob(z, p_mu, p_lv)
        data['kl_elementwise'] = kl_elem
        data['kl_samplewise'] = kl_elem.sum((1, 2, 3))
        data['kl_spatial'] = kl_analytical.sum(1)
    return out, data


def top_down_layer(sd, i, cfg, tape, training, input_=None, skip_input=None, inference_mode=False,
                   bu_value=None, n_img_prior=None, forced_latent=None, use_mode=False,
                   force_constant_output=False):
    """TopDownLayer.forward, models/lvae_layers.py:115-178."""
    prefix = 'top_down_layers.%d' % i
    is_top = i == len(cfg['z_dims']) - 1
    if is_top and not (input_ is None and skip_input is None):
        raise ValueError("In top layer, inputs should be None")
    if is_top:
        p_params = sd[prefix + '.top_prior_params']
        if n_img_prior is not None:
            p_params = p_params.expand(n_img_prior, -1, -1, -1)
    else:
        p_params = input_
    if inference_mode:
        q_params = bu_value if is_top else merge_layer(sd, prefix + '.merge', bu_value, p_params, cfg,
                                                       cfg['merge_type'], tape, training)
    else:
        q_params = None
    x, data = stochastic_block(sd, prefix + '.stochastic', p_params, q_params, cfg, not is_top, tape,
                               forced_latent, use_mode, force_constant_output)
    if cfg['stochastic_skip'] and not is_top:
        x = merge_layer(sd, prefix + '.skip_connection_merger', x, skip_input, cfg, 'residual', tape, training)
    x_pre_residual = x
    dws_left = cfg['downsample'][i]
    for j in range(cfg['blocks_per_layer']):
        up = dws_left > 0
        dws_left -= int(up)
        x = resampling_block(sd, '%s.deterministic_block.%d' % (prefix, j), x, cfg, 'top-down', up,
                             cfg['gated'], tape, training)
    return x, x_pre_residual, data


# --------------------------------------------------------------------------------------------------
# boilr helpers (restated; parity unpinned, see module docstring)
# --------------------------------------------------------------------------------------------------
def _centre(cur, tgt):
    d = tgt - cur
    return d // 2, d - d // 2


def pad_img_tensor(x, size):
    h0, h1 = _centre(x.shape[2], int(size[0]))
    w0, w1 = _centre(x.shape[3], int(size[1]))
    return F.pad(x, (w0, w1, h0, h1))


def crop_img_tensor(x, size):
    h0, _ = _centre(int(size[0]), x.shape[2])
    w0, _ = _centre(int(size[1]), x.shape[3])
    return x[:, :, h0:h0 + int(size[0]), w0:w0 + int(size[1])]


def free_bits_kl(kl, free_bits, eps=1e-6):
    if free_bits < eps:
        return kl.mean(0)
    return kl.clamp(min=free_bits).mean(0)


# --------------------------------------------------------------------------------------------------
# models/lvae.py
# --------------------------------------------------------------------------------------------------
def overall_downscale_factor(cfg):
    """models/lvae.py:56-58."""
    f = 2 ** sum(cfg['downsample'])
    return f if cfg['no_initial_downscaling'] else 2 * f


def get_padded_size(cfg, size):
    """models/lvae.py:327-349."""
    d = overall_downscale_factor(cfg)
    if len(size) == 4:
        size = size[2:]
    if len(size) != 2:
        raise RuntimeError("input size must be either (N, C, H, W) or (H, W), but it has length {} "
                           "(size={})".format(len(size), size))
    return [((s - 1) // d + 1) * d for s in size]


def get_top_prior_param_shape(cfg, n_imgs=1):
    """models/lvae.py:364-372."""
    d = overall_downscale_factor(cfg)
    sz = get_padded_size(cfg, cfg['img_shape'])
    return (n_imgs, cfg['z_dims'][-1] * 2, sz[0] // d, sz[1] // d)


def bottomup_pass(sd, cfg, x, tape, training):
    """models/lvae.py:216-227 with the stem of models/lvae.py:73-84."""
    stride = 1 if cfg['no_initial_downscaling'] else 2
    x = F.conv2d(x, sd['first_bottom_up.0.weight'], sd['first_bottom_up.0.bias'], stride=stride, padding=2)
    x = _act(cfg['nonlin'], x)
    x = resampling_block(sd, 'first_bottom_up.2', x, cfg, 'bottom-up', False, False, tape, training)
    bu_values = []
    for i in range(len(cfg['z_dims'])):
        dws_left = cfg['downsample'][i]
        for j in range(cfg['blocks_per_layer']):
            down = dws_left > 0
            dws_left -= int(down)
            x = resampling_block(sd, 'bottom_up_layers.%d.net.%d' % (i, j), x, cfg, 'bottom-up', down,
                                 cfg['gated'], tape, training)
        bu_values.append(x)
    return bu_values


def topdown_pass(sd, cfg, tape, training, bu_values=None, n_img_prior=None, mode_layers=None,
                 constant_layers=None, forced_latent=None):
    """models/lvae.py:229-315."""
    L = len(cfg['z_dims'])
    mode_layers = [] if mode_layers is None else mode_layers
    constant_layers = [] if constant_layers is None else constant_layers
    prior_experiment = len(mode_layers) > 0 or len(constant_layers) > 0
    inference_mode = bu_values is not None
    if inference_mode != (n_img_prior is None):
        raise RuntimeError("Number of images for top-down generation has to be given if and only if "
                           "we're not doing inference")
    if inference_mode and prior_experiment:
        raise RuntimeError("Prior experiments (e.g. sampling from mode) are not compatible with "
                           "inference mode")
    z, kl, kl_spatial = [None] * L, [None] * L, [None] * L
    if forced_latent is None:
        forced_latent = [None] * L
    logprob_p = 0.
    out = None
    for i in reversed(range(L)):
        bu = bu_values[i] if bu_values is not None else None
        out, _, aux = top_down_layer(sd, i, cfg, tape, training, out, out, inference_mode, bu, n_img_prior,
                                     forced_latent[i], i in mode_layers, i in constant_layers)
        z[i], kl[i], kl_spatial[i] = aux['z'], aux['kl_samplewise'], aux['kl_spatial']
        logprob_p = logprob_p + aux['logprob_p'].mean()
    # final_top_down, models/lvae.py:141-156
    k = 0
    if not cfg['no_initial_downscaling']:
        out = F.interpolate(out, scale_factor=2, mode='bilinear', align_corners=False)
        k = 1
    for j in range(cfg['blocks_per_layer']):
        out = resampling_block(sd, 'final_top_down.%d' % (k + j), out, cfg, 'top-down', False, cfg['gated'],
                               tape, training)
    return out, {'z': z, 'kl': kl, 'kl_spatial': kl_spatial, 'logprob_p': logprob_p}


# --------------------------------------------------------------------------------------------------
# lib/likelihoods.py
# --------------------------------------------------------------------------------------------------
def log_bernoulli(x, mean):
    """lib/likelihoods.py:385-388; torch BCE clamps each log term at -100."""
    return -F.binary_cross_entropy(mean, x, reduction='none').sum((1, 2, 3))


def discretized_mix_logistic_ll(x, l):
    """-discretized_mix_logistic_loss, lib/likelihoods.py:291-382. x in [-1,1] (B,3,H,W); l (B,10*nmix,H,W)."""
    B, _, H, W = x.shape
    nmix = l.shape[1] // 10
    xt = x.permute(0, 2, 3, 1)  # B H W 3
    lt = l.permute(0, 2, 3, 1)
    logits = lt[..., :nmix]
    rest = lt[..., nmix:].reshape(B, H, W, 3, 3 * nmix)
    means = rest[..., :nmix]
    log_scales = rest[..., nmix:2 * nmix].clamp(min=-7.)
    coeffs = torch.tanh(rest[..., 2 * nmix:])
    xe = xt.unsqueeze(-1).expand(B, H, W, 3, nmix)
    m1 = means[..., 0, :]
    m2 = means[..., 1, :] + coeffs[..., 0, :] * xe[..., 0, :]
    m3 = means[..., 2, :] + coeffs[..., 1, :] * xe[..., 0, :] + coeffs[..., 2, :] * xe[..., 1, :]
    means = torch.stack((m1, m2, m3), dim=3)
    centered = xe - means
    inv_s = torch.exp(-log_scales)
    plus_in = inv_s * (centered + 1. / 255.)
    min_in = inv_s * (centered - 1. / 255.)
    cdf_delta = torch.sigmoid(plus_in) - torch.sigmoid(min_in)
    log_cdf_plus = plus_in - F.softplus(plus_in)
    log_one_minus_cdf_min = -F.softplus(min_in)
    mid_in = inv_s * centered
    log_pdf_mid = mid_in - log_scales - 2. * F.softplus(mid_in)
    c_in = (cdf_delta > 1e-5).float()
    inner_inner = c_in * torch.log(cdf_delta.clamp(min=1e-12)) + (1. - c_in) * (log_pdf_mid - math.log(127.5))
    c_hi = (xe > 0.999).float()
    inner = c_hi * log_one_minus_cdf_min + (1. - c_hi) * inner_inner
    c_lo = (xe < -0.999).float()
    lp = c_lo * log_cdf_plus + (1. - c_lo) * inner
    lp = lp.sum(3) + torch.log_softmax(logits, dim=-1)
    return torch.logsumexp(lp, dim=-1).sum((1, 2))


def sample_discretized_mix_logistic(l, tape):
    """sample_from_discretized_mix_logistic, lib/stochastic.py:141-206. Returns (B,3,H,W) in [-1,1]."""
    B, _, H, W = l.shape
    nmix = l.shape[1] // 10
    lt = l.permute(0, 2, 3, 1)
    logits = lt[..., :nmix]
    rest = lt[..., nmix:].reshape(B, H, W, 3, 3 * nmix)
    u = tape.draw('uniform', (B, H, W, nmix), lo=1e-5, hi=1. - 1e-5)
    sel_idx = (logits.detach() - torch.log(-torch.log(u))).argmax(dim=3)
    sel = F.one_hot(sel_idx, nmix).float().view(B, H, W, 1, nmix)
    means = (rest[..., :nmix] * sel).sum(4)
    log_scales = (rest[..., nmix:2 * nmix] * sel).sum(4).clamp(min=-7.)
    coeffs = (torch.tanh(rest[..., 2 * nmix:]) * sel).sum(4)
    u2 = tape.draw('uniform', (B, H, W, 3), lo=1e-5, hi=1. - 1e-5)
    xs = means + torch.exp(log_scales) * (torch.log(u2) - torch.log(1. - u2))
    x0 = xs[..., 0].clamp(-1., 1.)
    x1 = (xs[..., 1] + coeffs[..., 0] * x0).clamp(-1., 1.)
    x2 = (xs[..., 2] + coeffs[..., 1] * x0 + coeffs[..., 2] * x1).clamp(-1., 1.)
    return torch.stack((x0, x1, x2), dim=3).permute(0, 3, 1, 2)


def log_discretized_logistic(x, mean, log_scale, n_bins=256, eps=1e-7):
    """lib/likelihoods.py:233-288 (single precision branch, reduce='none')."""
    scale = log_scale.exp()
    x = torch.floor(x * n_bins) / n_bins
    cdf_plus = torch.where(x < (n_bins - 1) / n_bins, torch.sigmoid((x + 1 / n_bins - mean) / scale),
                           torch.ones_like(x))
    cdf_minus = torch.where(x >= 1 / n_bins, torch.sigmoid((x - mean) / scale), torch.zeros_like(x))
    return torch.log(cdf_plus - cdf_minus + eps).sum((1, 2, 3))


def likelihood(sd, cfg, h, x, tape):
    """LikelihoodModule.forward, lib/likelihoods.py:33-48, for the four heads of models/lvae.py:158-170."""
    form = cfg['likelihood_form']
    p = F.conv2d(h, sd['likelihood.parameter_net.weight'], sd['likelihood.parameter_net.bias'], padding=1)
    if form == 'bernoulli':  # lib/likelihoods.py:51-78
        mean = torch.sigmoid(p)
        u = tape.draw('uniform', mean.shape)
        info = {'mean': mean, 'mode': torch.round(mean), 'sample': (u < mean).float(), 'params': mean}
        ll = None if x is None else log_bernoulli(x, mean)
    elif form == 'gaussian':  # lib/likelihoods.py:81-114, 391-411
        mean, lv = p.chunk(2, dim=1)
        eps = tape.draw('normal', mean.shape)
        info = {'mean': mean, 'mode': mean, 'sample': mean + (lv / 2).exp() * eps,
                'params': {'mean': mean, 'logvar': lv}}
        ll = None if x is None else (-0.5 * ((x - mean) ** 2 / lv.exp() + lv + math.log(2 * math.pi))
                                     ).sum((1, 2, 3))
    elif form == 'discr_log':  # lib/likelihoods.py:117-180
        mean, ls = p.chunk(2, dim=1)
        ls = (ls - 1.).clamp(min=-7.)
        mean = mean + 0.5
        u = tape.draw('uniform', mean.shape, lo=1e-7, hi=1 - 1e-7)  # logistic_rsample, lib/stochastic.py:115-138
        sample = (mean + ls.exp() * (torch.log(u) - torch.log(1 - u))).clamp(0., 1.)
        info = {'mean': mean, 'mode': mean, 'sample': sample, 'params': {'mean': mean, 'logscale': ls}}
        ll = None if x is None else log_discretized_logistic(x * (255 / 256) + 1 / 512, mean, ls)
    elif form == 'discr_log_mix':  # lib/likelihoods.py:183-230
        sample = ((sample_discretized_mix_logistic(p, tape) + 1) / 2).clamp(0., 1.)
        info = {'mean': None, 'mode': None, 'sample': sample, 'params': {'mean': None, 'all_params': p}}
        ll = None if x is None else discretized_mix_logistic_ll(x * 2 - 1, p)
    else:
        raise RuntimeError("Unrecognized likelihood '{}'".format(form))
    return ll, info


def lvae_forward(sd, cfg, x, tape, training=True):
    """LadderVAE.forward, models/lvae.py:172-214 -> the 12-key dict. `sd` BN buffers are updated in place."""
    img_size = x.shape[2:]
    x_pad = pad_img_tensor(x, get_padded_size(cfg, x.shape))
    bu_values = bottomup_pass(sd, cfg, x_pad, tape, training)
    out, td = topdown_pass(sd, cfg, tape, training, bu_values=bu_values)
    out = crop_img_tensor(out, img_size)
    ll, info = likelihood(sd, cfg, out, x, tape)
    kl = torch.stack(td['kl'], dim=1)  # (B, L)
    kl_sep = kl.sum(1)
    return {
        'll': ll, 'z': td['z'], 'kl': kl_sep.mean(), 'kl_sep': kl_sep, 'kl_avg_layerwise': kl.mean(0),
        'kl_spatial': td['kl_spatial'], 'kl_loss': free_bits_kl(kl, cfg['free_bits']).sum(),
        'logp': td['logprob_p'], 'out_mean': info['mean'], 'out_mode': info['mode'],
        'out_sample': info['sample'], 'likelihood_params': info['params'],
    }


def sample_prior(sd, cfg, n_imgs, tape, mode_layers=None, constant_layers=None, training=False):
    """LadderVAE.sample_prior, models/lvae.py:351-362."""
    out, _ = topdown_pass(sd, cfg, tape, training, n_img_prior=n_imgs, mode_layers=mode_layers,
                          constant_layers=constant_layers)
    out = crop_img_tensor(out, cfg['img_shape'])
    _, info = likelihood(sd, cfg, out, None, tape)
    return info['sample']


def inspect_layer_repr(sd, cfg, tape, n=8):
    """evaluate.py:95-114 without the image files: per layer i, n calls of sample_prior(n, mode_layers=range(i),
    constant_layers=range(i + 1, n_layers)) concatenated (the rows of the reference's nrow = n grid). Returns a list of n_layers tensors."""
    L = len(cfg['z_dims'])
    out = []
    for i in range(L):
        rows = [sample_prior(sd, cfg, n, tape, mode_layers=range(i), constant_layers=range(i + 1, L)) for _ in range(n)]
        out.append(torch.cat(rows))
    return out


# --------------------------------------------------------------------------------------------------
# experiment/experiment_manager.py
# --------------------------------------------------------------------------------------------------
def forward_pass(sd, cfg, x, tape, beta=1.0, training=True, param_keys=None):
    """LVAEExperiment.forward_pass, experiment/experiment_manager.py:322-367."""
    mo = lvae_forward(sd, cfg, x, tape, training)
    recons_sep = -mo['ll']
    elbo_sep = -(recons_sep + mo['kl_sep'])
    recons = recons_sep.mean()
    loss = recons + mo['kl_loss'] * beta
    keys = param_keys if param_keys is not None else [k for k in sd if is_parameter_key(k)]
    l2 = sum(torch.sum(sd[k] ** 2) for k in keys).sqrt()
    out = {'loss': loss, 'elbo': elbo_sep.mean(), 'elbo_sep': elbo_sep, 'kl': mo['kl'], 'l2': l2,
           'recons': recons, 'out_mean': mo['out_mean'], 'out_mode': mo['out_mode'],
           'out_sample': mo['out_sample'], 'likelihood_params': mo['likelihood_params'],
           'kl_avg_layerwise': mo['kl_avg_layerwise']}
    return out, mo


def is_parameter_key(k):
    return not (k.endswith('running_mean') or k.endswith('running_var') or k.endswith('num_batches_tracked'))


def adamax_step(params, grads, exp_avg, exp_inf, step, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """torch.optim.Adamax single-tensor update (experiment/experiment_manager.py:76-81 defaults). In place."""
    b1, b2 = betas
    for p, g, m, u in zip(params, grads, exp_avg, exp_inf):
        if weight_decay != 0:
            g = g + weight_decay * p
        m.mul_(b1).add_(g, alpha=1 - b1)
        torch.maximum(u * b2, g.abs() + eps, out=u)
        p.addcdiv_(m, u, value=-lr / (1 - b1 ** step))


def iw_log_likelihood(sd, cfg, x, tape, n_samples):
    """Importance-weighted bound per image: S eval-mode forward passes, logsumexp(elbo_sep) - log S
    (evaluate.py:30,86-87; the loop is boilr's test_procedure, restated)."""
    elbos = []
    with torch.no_grad():
        for _ in range(n_samples):
            mo = lvae_forward(sd, cfg, x, tape, training=False)
            elbos.append(mo['ll'] - mo['kl_sep'])
    e = torch.stack(elbos, 0)
    return torch.logsumexp(e, 0) - math.log(n_samples), e.mean(0)
