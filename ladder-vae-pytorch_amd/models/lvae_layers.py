"""Per-level blocks of the Ladder VAE on the HIP engine — mirrors models/lvae_layers.py of the reference:
TopDownLayer (:8-178), BottomUpLayer (:181-219), ResBlockWithResampling (:222-306), TopDownDeterministicResBlock
(:309-313), BottomUpDeterministicResBlock (:316-320), MergeLayer (:323-360), SkipConnectionMerger (:363-376).
Same constructor arguments, attribute / parameter names and argument checks; tensors are NHWC and every forward
takes the model's noise source explicitly.
"""
import torch
from torch import nn

from .. import ops
from ..lib.nn import Conv2dParams, ResidualBlock, ResidualGatedBlock
from ..lib.stochastic import NormalStochasticBlock2d


class ResBlockWithResampling(nn.Module):
    """Residual block with an optional x2 resampling step in front (strided 3x3 conv going up the inference path,
    3x3 transposed conv going down the generative path) — models/lvae_layers.py:222-306."""

    def __init__(self, mode, c_in, c_out, nonlin=nn.LeakyReLU, resample=False, res_block_kernel=None, groups=1,
                 batchnorm=True, res_block_type=None, dropout=None, min_inner_channels=None, gated=None):
        super().__init__()
        assert mode in ['top-down', 'bottom-up']
        if min_inner_channels is None:
            min_inner_channels = 0
        inner_filters = max(c_out, min_inner_channels)
        if resample:
            if mode == 'bottom-up':
                self.pre_conv = Conv2dParams(c_in, inner_filters, 3, stride=2, padding=1)
            else:
                self.pre_conv = Conv2dParams(c_in, inner_filters, 3, stride=2, padding=1, transposed=True,
                                             output_padding=1)
        elif c_in != inner_filters:
            self.pre_conv = Conv2dParams(c_in, inner_filters, 1)
        else:
            self.pre_conv = None
        self.res = ResidualBlock(channels=inner_filters, nonlin=nonlin, kernel=res_block_kernel, groups=groups,
                                 batchnorm=batchnorm, dropout=dropout, gated=gated, block_type=res_block_type)
        if inner_filters != c_out:
            self.post_conv = Conv2dParams(inner_filters, c_out, 1)
        else:
            self.post_conv = None

    def forward(self, x, noise):
        if self.pre_conv is not None:
            x = self.pre_conv(x)
        x = self.res(x, noise)
        if self.post_conv is not None:
            x = self.post_conv(x)
        return x


class TopDownDeterministicResBlock(ResBlockWithResampling):
    def __init__(self, *args, upsample=False, **kwargs):
        kwargs['resample'] = upsample
        super().__init__('top-down', *args, **kwargs)


class BottomUpDeterministicResBlock(ResBlockWithResampling):
    def __init__(self, *args, downsample=False, **kwargs):
        kwargs['resample'] = downsample
        super().__init__('bottom-up', *args, **kwargs)


class BottomUpLayer(nn.Module):
    """models/lvae_layers.py:181-219: `n_res_blocks` bottom-up blocks, the first `downsampling_steps` strided."""

    def __init__(self, n_res_blocks, n_filters, downsampling_steps=0, nonlin=None, batchnorm=True, dropout=None,
                 res_block_type=None, gated=None):
        super().__init__()
        blocks = []
        for _ in range(n_res_blocks):
            do_resample = downsampling_steps > 0
            downsampling_steps -= int(do_resample)
            blocks.append(BottomUpDeterministicResBlock(c_in=n_filters, c_out=n_filters, nonlin=nonlin,
                                                        downsample=do_resample, batchnorm=batchnorm, dropout=dropout,
                                                        res_block_type=res_block_type, gated=gated))
        self.net = nn.ModuleList(blocks)

    def forward(self, x, noise):
        for blk in self.net:
            x = blk(x, noise)
        return x


class MergeLayer(nn.Module):
    """models/lvae_layers.py:323-360: channel concat -> 1x1 conv [-> gated residual block]. The concat is never
    materialised: the 1x1 conv kernel reads its K range from the two tensors."""

    def __init__(self, channels, merge_type, nonlin=nn.LeakyReLU, batchnorm=True, dropout=None, res_block_type=None):
        super().__init__()
        try:
            iter(channels)
        except TypeError:
            channels = [channels] * 3
        else:
            if len(channels) == 1:
                channels = [channels[0]] * 3
        assert len(channels) == 3
        self.merge_type = merge_type
        if merge_type == 'linear':
            self.layer = Conv2dParams(channels[0] + channels[1], channels[2], 1)
        elif merge_type == 'residual':
            self.layer = nn.ModuleList([
                Conv2dParams(channels[0] + channels[1], channels[2], 1, padding=0),
                ResidualGatedBlock(channels[2], nonlin, batchnorm=batchnorm, dropout=dropout, block_type=res_block_type),
            ])

    def forward(self, x, y, noise):
        if self.merge_type == 'linear':
            return self.layer(x, x2=y)
        h = self.layer[0](x, x2=y)
        return self.layer[1](h, noise)


class SkipConnectionMerger(MergeLayer):
    merge_type = 'residual'

    def __init__(self, channels, nonlin, batchnorm, dropout, res_block_type):
        super().__init__(channels, self.merge_type, nonlin, batchnorm, dropout=dropout, res_block_type=res_block_type)


class TopDownLayer(nn.Module):
    """One stochastic level of the generative path — models/lvae_layers.py:8-178."""

    def __init__(self, z_dim, n_res_blocks, n_filters, is_top_layer=False, downsampling_steps=None, nonlin=None,
                 merge_type=None, batchnorm=True, dropout=None, stochastic_skip=False, res_block_type=None, gated=None,
                 learn_top_prior=False, top_prior_param_shape=None, analytical_kl=False):
        super().__init__()
        self.is_top_layer = is_top_layer
        self.z_dim = z_dim
        self.stochastic_skip = stochastic_skip
        self.learn_top_prior = learn_top_prior
        self.analytical_kl = analytical_kl
        if is_top_layer:
            self.top_prior_params = nn.Parameter(torch.zeros(top_prior_param_shape), requires_grad=learn_top_prior)
        dws_left = downsampling_steps
        blocks = []
        for _ in range(n_res_blocks):
            do_resample = dws_left > 0
            dws_left -= int(do_resample)
            blocks.append(TopDownDeterministicResBlock(n_filters, n_filters, nonlin, upsample=do_resample,
                                                       batchnorm=batchnorm, dropout=dropout,
                                                       res_block_type=res_block_type, gated=gated))
        self.deterministic_block = nn.ModuleList(blocks)
        self.stochastic = NormalStochasticBlock2d(c_in=n_filters, c_vars=z_dim, c_out=n_filters,
                                                  transform_p_params=(not is_top_layer))
        if not is_top_layer:
            self.merge = MergeLayer(channels=n_filters, merge_type=merge_type, nonlin=nonlin, batchnorm=batchnorm,
                                    dropout=dropout, res_block_type=res_block_type)
            if stochastic_skip:
                self.skip_connection_merger = SkipConnectionMerger(channels=n_filters, nonlin=nonlin,
                                                                   batchnorm=batchnorm, dropout=dropout,
                                                                   res_block_type=res_block_type)

    def forward(self, input_=None, skip_connection_input=None, inference_mode=False, bu_value=None, n_img_prior=None,
                forced_latent=None, use_mode=False, force_constant_output=False, noise=None, rows=None):
        inputs_none = input_ is None and skip_connection_input is None
        if self.is_top_layer and not inputs_none:
            raise ValueError("In top layer, inputs should be None")
        n_img = None
        if self.is_top_layer:
            # logical (1, 2Z, h, w) parameter; the kernel reads it as NHWC and broadcasts it over the batch itself
            p_params = self.top_prior_params.permute(0, 2, 3, 1)
            n_img = n_img_prior
        else:
            # input_ feeds conv_in_p, the merge layer (inference) and — when it is also the skip input, as in LadderVAE — the skip
            # merger: explicit aliases, so that the three gradients are summed by one kernel (ops.FanoutFn)
            same_skip = self.stochastic_skip and skip_connection_input is input_
            al = ops.fanout(input_, 1 + int(inference_mode) + int(same_skip))
            p_params = al[0]
            p_merge = al[1] if inference_mode else None
            if same_skip:
                skip_connection_input = al[-1]
        if inference_mode:
            q_params = bu_value if self.is_top_layer else self.merge(bu_value, p_merge, noise)
        else:
            q_params = None
        x, data_stoch = self.stochastic(p_params=p_params, q_params=q_params, forced_latent=forced_latent,
                                        use_mode=use_mode, force_constant_output=force_constant_output,
                                        analytical_kl=self.analytical_kl, noise=noise, n_img=n_img,
                                        need_kl_elementwise=False, rows=rows)
        if self.stochastic_skip and not self.is_top_layer:
            x = self.skip_connection_merger(x, skip_connection_input, noise)
        x_pre_residual = x
        for blk in self.deterministic_block:
            x = blk(x, noise)
        keys = ['z', 'kl_samplewise', 'kl_spatial', 'logprob_p', 'logprob_q']
        return x, x_pre_residual, {k: data_stoch[k] for k in keys}
