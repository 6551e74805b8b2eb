"""Likelihood heads on HIP kernels — mirrors lib/likelihoods.py:13-78 (LikelihoodModule, BernoulliLikelihood) and
:183-230 (DiscretizedLogisticMixLikelihood): 3x3 parameter conv -> (mean, mode, sample, log-likelihood), returned as
`(ll, {'mean','mode','sample','params'})`. Input feature map NHWC; image `x` NHWC; outputs NHWC.
"""
import torch
from torch import nn

from .. import kernels as K
from .. import ops
from .nn import Conv2dParams


class LikelihoodModule(nn.Module):
    def forward(self, input_, x, noise):
        raise NotImplementedError


class BernoulliLikelihood(LikelihoodModule):
    def __init__(self, ch_in, color_channels):
        super().__init__()
        self.parameter_net = Conv2dParams(ch_in, color_channels, 3, padding=1)

    def forward(self, input_, x, noise):
        logits = self.parameter_net(input_)
        # the reference draws rand_like(params) in NCHW order (lib/likelihoods.py:75); a tape entry is permuted to NHWC
        u = noise.uniform(tuple(logits.shape), 0.0, 1.0, logits.device, channel_last=False)
        ll, mean, mode, sample = ops.BernoulliFn.apply(logits, x, u)
        if x is None:
            ll = None
        return ll, {'mean': mean, 'mode': mode, 'sample': sample, 'params': mean}


class DiscretizedLogisticMixLikelihood(LikelihoodModule):
    """Mixture of n_components discretized logistics (PixelCNN++ form; the reference's only use is 10). Mean and mode are None, as in
    the reference (lib/likelihoods.py:207-218)."""

    SUPPORTED = (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20)   # counts the kernels are instantiated for (csrc/likelihood.hip)

    def __init__(self, ch_in, n_components=10):
        super().__init__()
        if n_components not in self.SUPPORTED:
            raise NotImplementedError("the DMoL kernels are instantiated for %s mixture components, not %r" % (self.SUPPORTED, n_components))
        self.n_components = n_components
        self.parameter_net = Conv2dParams(ch_in, 10 * n_components, 3, padding=1)

    def forward(self, input_, x, noise):
        l = self.parameter_net(input_)
        N, H, W, _ = l.shape
        u_mix = noise.uniform((N, H, W, self.n_components), 1e-5, 1. - 1e-5, l.device)
        u_log = noise.uniform((N, H, W, 3), 1e-5, 1. - 1e-5, l.device)
        with torch.no_grad():
            sample = K.dmol_sample(l.detach(), u_mix, u_log)
        ll = ops.DmolFn.apply(l, x) if x is not None else None
        return ll, {'mean': None, 'mode': None, 'sample': sample, 'params': {'mean': None, 'all_params': l}}


class GaussianLikelihood(LikelihoodModule):
    """lib/likelihoods.py:81-114: conv -> (mean, logvar); sample = mean + exp(logvar/2) * eps."""

    def __init__(self, ch_in, color_channels):
        super().__init__()
        self.color_channels = color_channels
        self.parameter_net = Conv2dParams(ch_in, 2 * color_channels, 3, padding=1)

    def forward(self, input_, x, noise):
        p = self.parameter_net(input_)
        N, H, W, _ = p.shape
        eps = noise.normal((N, H, W, self.color_channels), p.device)
        ll, sample = ops.GaussianFn.apply(p, x, eps)
        if x is None:
            ll = None
        mean, lv = p[..., :self.color_channels], p[..., self.color_channels:]
        return ll, {'mean': mean, 'mode': mean, 'sample': sample, 'params': {'mean': mean, 'logvar': lv}}


class DiscretizedLogisticLikelihood(LikelihoodModule):
    """lib/likelihoods.py:117-180 (256 bins, log_scale_bias -1, clamp -7, mean + 0.5)."""

    log_scale_bias = -1.

    def __init__(self, ch_in, color_channels, n_bins, double=False):
        super().__init__()
        if n_bins != 256 or double:
            raise NotImplementedError("the discretized-logistic kernel is built for 256 bins in single precision")
        self.n_bins = n_bins
        self.color_channels = color_channels
        self.parameter_net = Conv2dParams(ch_in, 2 * color_channels, 3, padding=1)

    def forward(self, input_, x, noise):
        raw = self.parameter_net(input_)
        N, H, W, _ = raw.shape
        # logistic_rsample draws uniform_(1e-7, 1-1e-7) with the shape of the mean, NCHW (lib/stochastic.py:131-132)
        u = noise.uniform((N, H, W, self.color_channels), 1e-7, 1 - 1e-7, raw.device, channel_last=False)
        ll, mean, ls, sample = ops.DiscrLogisticFn.apply(raw, x, u)
        if x is None:
            ll = None
        return ll, {'mean': mean, 'mode': mean, 'sample': sample, 'params': {'mean': mean, 'logscale': ls}}
