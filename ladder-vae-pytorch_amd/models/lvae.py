"""LadderVAE on the MI355X HIP engine — drop-in for the reference's models/lvae.py:15-372.

Same constructor keyword arguments and defaults, the same 12-key forward() dict (shapes in the reference's NCHW
convention: tensors are returned as permuted views of the engine's NHWC buffers, no copy), the same helper methods
(`bottomup_pass`, `topdown_pass`, `sample_prior`, `pad_input`, `get_padded_size`, `get_top_prior_param_shape`),
attributes (`n_layers`, `img_shape`, `likelihood`, `global_step`), exceptions, and state_dict key scheme.

Differences by design (not by omission):
  * `logp` (data['logprob_p'] of topdown_pass) is computed by one kernel from the per-layer rows and carries no autograd graph
    (the reference never differentiates it: it is a logged metric, experiment/experiment_manager.py does not read it);
  * every arithmetic op is a hand-written gfx950 kernel from liblvae_hip.so; the model refuses to run on CPU;
  * parameters live in one flat arena (arena.py); their gradients are accumulated in place by the wgrad kernels;
  * noise comes from `self.noise` (noise.PhiloxNoise on device; noise.TapeNoise to replay the reference's draws).

The boilr helpers the reference imports (pad/crop, Interpolate, free_bits_kl) are restated — parity unpinned
(SURVEY.md §8c).
"""
import numpy as np
import torch
from torch import nn

from .. import kernels as K
from .. import ops
from ..arena import ParamArena
from ..lib.likelihoods import (BernoulliLikelihood, DiscretizedLogisticLikelihood, DiscretizedLogisticMixLikelihood,
                               GaussianLikelihood)
from ..lib.nn import BatchNorm2dParams, Conv2dParams, Placeholder, act_name
from ..noise import PhiloxNoise
from .lvae_layers import BottomUpDeterministicResBlock, BottomUpLayer, TopDownDeterministicResBlock, TopDownLayer


def _nchw(t):
    """NHWC engine tensor -> logical NCHW view (no copy)."""
    return None if t is None else t.permute(0, 3, 1, 2)


class LadderVAE(nn.Module):

    def __init__(self, color_ch, z_dims, blocks_per_layer=2, downsample=None, nonlin='elu', merge_type=None,
                 batchnorm=True, stochastic_skip=False, n_filters=32, dropout=None, free_bits=0.0,
                 learn_top_prior=False, img_shape=None, likelihood_form=None, res_block_type=None, gated=False,
                 no_initial_downscaling=False, analytical_kl=False):
        super().__init__()
        self.global_step = 0  # boilr.models.BaseGenerativeModel attribute, read by forward_pass for beta annealing
        self.color_ch = color_ch
        self.z_dims = z_dims
        self.blocks_per_layer = blocks_per_layer
        self.downsample = downsample
        self.n_layers = len(self.z_dims)
        self.stochastic_skip = stochastic_skip
        self.n_filters = n_filters
        self.dropout = dropout
        self.free_bits = free_bits
        self.learn_top_prior = learn_top_prior
        self.img_shape = tuple(img_shape)
        self.res_block_type = res_block_type
        self.gated = gated
        self.no_initial_downscaling = no_initial_downscaling
        if self.downsample is None:
            self.downsample = [0] * self.n_layers
        self.overall_downscale_factor = int(np.power(2, sum(self.downsample)))
        if not no_initial_downscaling:
            self.overall_downscale_factor *= 2
        assert max(self.downsample) <= self.blocks_per_layer
        assert len(self.downsample) == self.n_layers
        if nonlin not in ('relu', 'leakyrelu', 'elu', 'selu'):
            raise KeyError(nonlin)
        self.act = act_name(nonlin)

        stride = 1 if no_initial_downscaling else 2
        self.first_bottom_up = nn.ModuleList([
            Conv2dParams(color_ch, n_filters, 5, stride=stride, padding=2),
            Placeholder(self.act),
            BottomUpDeterministicResBlock(c_in=n_filters, c_out=n_filters, nonlin=nonlin, batchnorm=batchnorm,
                                          dropout=dropout, res_block_type=res_block_type),
        ])
        self.top_down_layers = nn.ModuleList([])
        self.bottom_up_layers = nn.ModuleList([])
        for i in range(self.n_layers):
            is_top = i == self.n_layers - 1
            self.bottom_up_layers.append(
                BottomUpLayer(n_res_blocks=self.blocks_per_layer, n_filters=n_filters,
                              downsampling_steps=self.downsample[i], nonlin=nonlin, batchnorm=batchnorm, dropout=dropout,
                              res_block_type=res_block_type, gated=gated))
            self.top_down_layers.append(
                TopDownLayer(z_dim=z_dims[i], n_res_blocks=blocks_per_layer, n_filters=n_filters, is_top_layer=is_top,
                             downsampling_steps=self.downsample[i], nonlin=nonlin, merge_type=merge_type,
                             batchnorm=batchnorm, dropout=dropout, stochastic_skip=stochastic_skip,
                             learn_top_prior=learn_top_prior, top_prior_param_shape=self.get_top_prior_param_shape(),
                             res_block_type=res_block_type, gated=gated, analytical_kl=analytical_kl))
        modules = []
        if not no_initial_downscaling:
            modules.append(Placeholder('bilinear x2'))
        for i in range(blocks_per_layer):
            modules.append(TopDownDeterministicResBlock(c_in=n_filters, c_out=n_filters, nonlin=nonlin,
                                                        batchnorm=batchnorm, dropout=dropout,
                                                        res_block_type=res_block_type, gated=gated))
        self.final_top_down = nn.ModuleList(modules)

        if likelihood_form == 'bernoulli':
            self.likelihood = BernoulliLikelihood(n_filters, color_ch)
        elif likelihood_form == 'gaussian':
            self.likelihood = GaussianLikelihood(n_filters, color_ch)
        elif likelihood_form == 'discr_log':
            self.likelihood = DiscretizedLogisticLikelihood(n_filters, color_ch, 256)
        elif likelihood_form == 'discr_log_mix':
            self.likelihood = DiscretizedLogisticMixLikelihood(n_filters)
        else:
            raise RuntimeError("Unrecognized likelihood '{}'".format(likelihood_form))

        self.noise = PhiloxNoise(seed=0)
        self.arena = None
        # engine attribute (not a constructor argument of the reference): 'f32', or 'bf16' = bf16 matrix-core operands with fp32
        # accumulation, statistics, KL and likelihood (the arithmetic of the reference under torch.autocast(bfloat16)) for the
        # convolutions that have a bf16 kernel (lvae_conv2d_bf16); activations, parameters and gradients stay fp32 in HBM
        self.compute_dtype = 'f32'
        self.grad_tracker = None  # dist.GradAllReduce when gradients are exchanged while backward runs (engine.TrainStep sets it)

    # ------------------------------------------------------------------------------------------------------------
    # engine plumbing
    # ------------------------------------------------------------------------------------------------------------
    def _apply(self, fn, *a, **kw):
        self.arena = None  # .to()/.cuda() re-create the parameter storages; re-pack lazily
        return super()._apply(fn, *a, **kw)

    def pack(self, device=None):
        """Move all parameters into the flat arena (idempotent). Called lazily by the first forward."""
        if device is None:
            device = next(self.parameters()).device
        device = torch.device(device)
        if device.type != 'cuda':
            raise K._C.LvaeHipError("LadderVAE (HIP engine) runs on an MI355X only: move the model with .cuda() first; "
                                    "there is no CPU fallback (the CPU restatement lives in oracle/ for tests)")
        first = next(self.parameters())
        if self.arena is None or not self.arena.owns(first) or first.device != device:
            if first.device != device:
                super()._apply(lambda t: t.to(device))
            self.arena = ParamArena(self, device, segment_of=self.grad_segment_of)
        return self.arena

    def grad_segments(self):
        """Parameter-name prefixes in the order their gradients complete during backward = reverse execution order of forward
        (forward: stem, bottom_up_layers[0..L-1], top_down_layers[L-1..0], final_top_down, likelihood — models/lvae.py:172-315).
        The last segment (stem + first block) also takes `top_prior_params`, whose gradient arrives through autograd's own
        accumulation node rather than from one of our backward launches."""
        L = self.n_layers
        return (['likelihood.', 'final_top_down.'] + ['top_down_layers.%d.' % i for i in range(L)] +
                ['bottom_up_layers.%d.' % i for i in reversed(range(L))] + ['first_bottom_up.'])

    def grad_segment_of(self, name):
        segs = self.grad_segments()
        if name.endswith('top_prior_params'):
            return len(segs) - 1
        for i, pre in enumerate(segs):
            if name.startswith(pre):
                return i
        raise KeyError(name)

    def _mark(self, x, prefix):
        """Backward of this identity node runs once every backward launch of segment `prefix` has been issued (x is the segment's
        input, and every node of the segment is an ancestor of x's gradient); it tells the gradient exchange so."""
        if self.grad_tracker is None or not torch.is_grad_enabled() or not x.requires_grad:
            return x
        return ops.segment_mark(x, self.grad_tracker, self.grad_segments().index(prefix))

    def zero_grad(self, set_to_none=False):
        if self.arena is not None:
            self.arena.zero_grad()
        else:
            super().zero_grad(set_to_none=set_to_none)

    def bn_modules(self):
        return [m for m in self.modules() if isinstance(m, BatchNorm2dParams)]

    def _mask_plan(self, N):
        """(number of Dropout2d draws of one training forward, N, C, p) when they all share one shape, else None."""
        if not self.training or not self.dropout:
            return None
        if getattr(self, '_n_drop', None) is None:
            from ..lib.nn import ResidualBlock
            self._n_drop = sum(sum(m._drops) for m in self.modules() if isinstance(m, ResidualBlock))
        return (self._n_drop, N, self.n_filters, float(self.dropout))

    def _begin(self, ref_tensor, batch=None):
        self.pack(ref_tensor.device if ref_tensor is not None else None)
        K.set_precision(self.compute_dtype)
        self.noise.begin(next(self.parameters()).device, self._mask_plan(batch) if batch else None)

    # ------------------------------------------------------------------------------------------------------------
    # reference API
    # ------------------------------------------------------------------------------------------------------------
    def forward(self, x):
        if not x.is_cuda:
            raise K._C.LvaeHipError("LadderVAE (HIP engine) needs a GPU tensor; got %s" % x.device)
        self._begin(x, batch=x.shape[0])
        img_size = x.size()[2:]
        x = x.contiguous().float()
        # NCHW image -> centred zero pad -> NHWC, one kernel (models/lvae.py:176, 317-325)
        x_pad = K.pad_crop(x, True, self.get_padded_size(x.size()), False)
        x_nhwc = x_pad if tuple(img_size) == tuple(x_pad.shape[1:3]) else K.pad_crop(x, True, img_size, False)

        bu_values = self._bottomup(x_pad)
        out, td_data = self._topdown(bu_values)
        out = ops.CropFn.apply(out, tuple(int(s) for s in img_size)) if tuple(out.shape[1:3]) != tuple(img_size) else out
        out = self._mark(out, 'likelihood.')
        ll, likelihood_info = self.likelihood(out, x_nhwc, self.noise)

        kl_ln = ops.StackFn.apply(*td_data['kl'])  # (L, N)
        kl_sep, kl_avg_layerwise, kl_loss, kl = ops.KLBookFn.apply(kl_ln, float(self.free_bits))
        self.noise.end()

        params = likelihood_info['params']
        if isinstance(params, dict):
            params = {k: _nchw(v) for k, v in params.items()}
        else:
            params = _nchw(params)
        return {
            'll': ll,
            'z': [_nchw(z) for z in td_data['z']],
            'kl': kl,
            'kl_sep': kl_sep,
            'kl_avg_layerwise': kl_avg_layerwise,
            'kl_spatial': td_data['kl_spatial'],
            'kl_loss': kl_loss,
            'logp': td_data['logprob_p'],
            'out_mean': _nchw(likelihood_info['mean']),
            'out_mode': _nchw(likelihood_info['mode']),
            'out_sample': _nchw(likelihood_info['sample']),
            'likelihood_params': params,
        }

    def _bottomup(self, x):
        stem, _, block = self.first_bottom_up
        x = stem(x, out_act=self.act)
        x = block(x, self.noise)
        bu_values = []
        for i in range(self.n_layers):
            x = self.bottom_up_layers[i](self._mark(x, 'bottom_up_layers.%d.' % i), self.noise)
            if i + 1 < self.n_layers:
                x, x_td = ops.fanout(x, 2)   # consumers: the next bottom-up layer and this level's top-down layer
            else:
                x_td = x
            bu_values.append(x_td)
        return bu_values

    def bottomup_pass(self, x):
        """models/lvae.py:216-227. x: padded image (N,C,Hp,Wp); returns the list of per-level NCHW feature maps."""
        self._begin(x)
        x_pad = K.pad_crop(x.contiguous().float(), True, x.shape[2:], False)
        out = [_nchw(b) for b in self._bottomup(x_pad)]
        self.noise.end()  # the next call draws fresh dropout masks
        return out

    def _topdown(self, bu_values=None, n_img_prior=None, mode_layers=None, constant_layers=None, forced_latent=None):
        if mode_layers is None:
            mode_layers = []
        if constant_layers is None:
            constant_layers = []
        prior_experiment = len(mode_layers) > 0 or len(constant_layers) > 0
        inference_mode = bu_values is not None
        if inference_mode != (n_img_prior is None):
            raise RuntimeError("Number of images for top-down generation has to be given if and only if we're "
                               "not doing inference")
        if inference_mode and prior_experiment:
            raise RuntimeError("Prior experiments (e.g. sampling from mode) are not compatible with inference mode")
        z = [None] * self.n_layers
        kl = [None] * self.n_layers
        kl_spatial = [None] * self.n_layers
        if forced_latent is None:
            forced_latent = [None] * self.n_layers
        out = None
        # per-layer, per-sample log p(z) and KL are written by the stochastic kernels straight into the rows of two [L][N] matrices
        n_rows = bu_values[0].shape[0] if inference_mode else int(n_img_prior)
        dev = next(self.parameters()).device
        lp_mat = torch.empty((self.n_layers, n_rows), dtype=torch.float32, device=dev)
        kl_mat = torch.empty((self.n_layers, n_rows), dtype=torch.float32, device=dev)
        for i in reversed(range(self.n_layers)):
            try:
                bu_value = bu_values[i]
            except TypeError:
                bu_value = None
            fl = forced_latent[i]
            if fl is not None:
                fl = fl.permute(0, 2, 3, 1).contiguous()
            if out is not None:
                out = self._mark(out, 'top_down_layers.%d.' % i)
            elif bu_value is not None:
                bu_value = self._mark(bu_value, 'top_down_layers.%d.' % i)
            out, _, aux = self.top_down_layers[i](out, skip_connection_input=out, inference_mode=inference_mode,
                                                  bu_value=bu_value, n_img_prior=n_img_prior, use_mode=i in mode_layers,
                                                  force_constant_output=i in constant_layers, forced_latent=fl,
                                                  noise=self.noise, rows=(lp_mat[i], kl_mat[i]))
            z[i] = aux['z']
            kl[i] = aux['kl_samplewise']
            kl_spatial[i] = aux['kl_spatial']
        with torch.no_grad():  # models/lvae.py:301: sum over layers of the batch-mean log p(z); a metric (returned without a graph)
            logprob_p = K.sum_of_row_means(lp_mat).view(())
        out = self._mark(out, 'final_top_down.')
        for mod in self.final_top_down:
            if isinstance(mod, Placeholder):
                out = ops.UpsampleFn.apply(out)
            else:
                out = mod(out, self.noise)
        return out, {'z': z, 'kl': kl, 'kl_spatial': kl_spatial, 'logprob_p': logprob_p}

    def topdown_pass(self, bu_values=None, n_img_prior=None, mode_layers=None, constant_layers=None,
                     forced_latent=None):
        """models/lvae.py:229-315 (NCHW in / NCHW out wrapper of the engine's NHWC pass)."""
        self._begin(bu_values[0] if bu_values is not None else None)
        if bu_values is not None:
            bu_values = [b.permute(0, 2, 3, 1).contiguous() for b in bu_values]
        out, data = self._topdown(bu_values, n_img_prior, mode_layers, constant_layers, forced_latent)
        self.noise.end()  # every call returns a fresh sample, as the reference's rsample() does
        data = dict(data)
        data['z'] = [_nchw(t) for t in data['z']]
        return _nchw(out), data

    def pad_input(self, x):
        """models/lvae.py:317-325 — centred zero pad to a multiple of the overall downscale factor (NCHW in/out)."""
        size = self.get_padded_size(x.size())
        return K.pad_crop(x.contiguous().float(), True, size, True)

    def get_padded_size(self, size):
        """models/lvae.py:327-349."""
        dwnsc = self.overall_downscale_factor
        if len(size) == 4:
            size = size[2:]
        if len(size) != 2:
            raise RuntimeError("input size must be either (N, C, H, W) or (H, W), but it has length {} (size={})".format(
                len(size), size))
        return list(((s - 1) // dwnsc + 1) * dwnsc for s in size)

    def sample_prior(self, n_imgs, mode_layers=None, constant_layers=None):
        """models/lvae.py:351-362."""
        self._begin(None)
        out, _ = self._topdown(n_img_prior=n_imgs, mode_layers=mode_layers, constant_layers=constant_layers)
        if tuple(out.shape[1:3]) != tuple(self.img_shape):
            out = ops.CropFn.apply(out, tuple(self.img_shape))
        _, likelihood_data = self.likelihood(out, None, self.noise)
        self.noise.end()
        return _nchw(likelihood_data['sample'])

    def get_top_prior_param_shape(self, n_imgs=1):
        """models/lvae.py:364-372."""
        dwnsc = self.overall_downscale_factor
        sz = self.get_padded_size(self.img_shape)
        return (n_imgs, self.z_dims[-1] * 2, sz[0] // dwnsc, sz[1] // dwnsc)
