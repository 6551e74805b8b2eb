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
    """10-component mixture (PixelCNN++ form). Mean and mode are None, as in the reference (lib/likelihoods.py:207-218)."""

    def __init__(self, ch_in, n_components=10):
        super().__init__()
        if n_components != 10:
            raise NotImplementedError("the DMoL kernels are built for 10 mixture components")
        self.parameter_net = Conv2dParams(ch_in, 10 * n_components, 3, padding=1)

    def forward(self, input_, x, noise):
        l = self.parameter_net(input_)
        N, H, W, _ = l.shape
        u_mix = noise.uniform((N, H, W, 10), 1e-5, 1. - 1e-5, l.device)
        u_log = noise.uniform((N, H, W, 3), 1e-5, 1. - 1e-5, l.device)
        with torch.no_grad():
            sample = K.dmol_sample(l.detach(), u_mix, u_log)
        ll = ops.DmolFn.apply(l, x) if x is not None else None
        return ll, {'mean': None, 'mode': None, 'sample': sample, 'params': {'mean': None, 'all_params': l}}


class GaussianLikelihood(LikelihoodModule):
    def __init__(self, ch_in, color_channels):
        super().__init__()
        raise NotImplementedError("likelihood 'gaussian' (lib/likelihoods.py:81-114) is not on the BASELINE hot path and "
                                  "has no HIP kernel yet (SURVEY.md §8f rank 4)")


class DiscretizedLogisticLikelihood(LikelihoodModule):
    def __init__(self, ch_in, color_channels, n_bins, double=False):
        super().__init__()
        raise NotImplementedError("likelihood 'discr_log' (lib/likelihoods.py:117-180) is not on the BASELINE hot path and "
                                  "has no HIP kernel yet (SURVEY.md §8f rank 4)")
