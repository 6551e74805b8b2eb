"""Stochastic layer on HIP kernels — mirrors the reference's lib/stochastic.py:7-112 (NormalStochasticBlock2d)
and :209-226 (kl_normal_mc): same constructor, same `forward` keyword arguments, same keys in the returned dict.
Tensors are NHWC inside the engine; `data['z']` etc. are returned NHWC and converted by the model.
"""
import torch
from torch import nn

from .. import kernels as K
from .. import ops
from .nn import Conv2dParams


class NormalStochasticBlock2d(nn.Module):
    """conv to (mu, logvar) of q (and of p unless top layer), sample z, conv_out(z); log p(z), log q(z), KL."""

    def __init__(self, c_in, c_vars, c_out, kernel=3, transform_p_params=True):
        super().__init__()
        assert kernel % 2 == 1
        pad = kernel // 2
        self.transform_p_params = transform_p_params
        self.c_in, self.c_out, self.c_vars = c_in, c_out, c_vars
        if transform_p_params:
            self.conv_in_p = Conv2dParams(c_in, 2 * c_vars, kernel, padding=pad)
        self.conv_in_q = Conv2dParams(c_in, 2 * c_vars, kernel, padding=pad)
        self.conv_out = Conv2dParams(c_vars, c_out, kernel, padding=pad)

    def forward(self, p_params, q_params=None, forced_latent=None, use_mode=False, force_constant_output=False,
                analytical_kl=False, noise=None, n_img=None, need_kl_elementwise=True, rows=None):
        """need_kl_elementwise=False (engine-only keyword, used by TopDownLayer which drops that key, models/lvae_layers.py:163-170):
        skip the pass that materialises `kl_elementwise` (lib/stochastic.py:88-91); the per-sample and per-pixel sums do not need it."""
        assert (forced_latent is None) or (not use_mode)
        if self.transform_p_params:
            p_params = self.conv_in_p(p_params)
        else:
            assert p_params.size(3) == 2 * self.c_vars
        if q_params is not None:
            q_params = self.conv_in_q(q_params)
        ref = q_params if q_params is not None else p_params
        N = ref.shape[0] if n_img is None else n_img
        H, W = ref.shape[1], ref.shape[2]
        dev = ref.device
        if forced_latent is not None:
            mode, src = 2, forced_latent.contiguous()
        elif use_mode:
            mode, src = 1, None
        else:
            mode, src = 0, noise.normal((N, H, W, self.c_vars), dev)
        # rows: engine-only keyword — (N,) views the kernel writes log p(z) and the KL into (rows of the model's [L][N] matrices)
        outs = ops.NormalStochFn.apply(p_params, q_params, src, mode, bool(analytical_kl), self.c_vars, N, rows)
        z = outs[0]
        if force_constant_output:
            # lib/stochastic.py:71-73 — prior experiments only (no gradient flows here)
            if torch.is_grad_enabled() and z.requires_grad:
                raise RuntimeError("force_constant_output is a sampling-time option; run it under torch.no_grad()")
            z = z[0:1].expand_as(z).contiguous()
            p_params = p_params[0:1].expand(N, -1, -1, -1).contiguous()
        out = self.conv_out(z)
        data = {'z': z, 'p_params': p_params, 'q_params': q_params, 'logprob_p': outs[1], 'logprob_q': None,
                'kl_elementwise': None, 'kl_samplewise': None, 'kl_spatial': None}
        if q_params is not None:
            data['logprob_q'], data['kl_samplewise'], data['kl_spatial'] = outs[2], outs[3], outs[4]
            if need_kl_elementwise:
                data['kl_elementwise'] = ops.KlElementwiseFn.apply(z, p_params, q_params, bool(analytical_kl))
        return out, data


def kl_normal_mc(z, p_mulv, q_mulv):
    """lib/stochastic.py:209-226 on the HIP kernel: elementwise log q(z) - log p(z). NHWC tensors (mu | logvar on the channel
    axis); p_mulv / q_mulv may have batch size 1 (broadcast over the batch of z), as the reference's broadcasting allows."""
    return ops.KlElementwiseFn.apply(z.contiguous(), p_mulv.contiguous(), q_mulv.contiguous(), False)
