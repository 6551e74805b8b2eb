"""Residual block family on HIP kernels — mirrors the public surface of the reference's lib/nn.py
(ResidualBlock lib/nn.py:5-99, ResidualGatedBlock :102-105, GateLayer2d :108-126): same constructor arguments,
same `block.<idx>` parameter names, same exceptions. The arithmetic is in liblvae_hip (see ops.ResBlockFn).

Modules take and return NHWC tensors (N,H,W,C); the model converts at its boundary.
"""
import math

import torch
from torch import nn
from torch.nn import init

from .. import kernels as K
from .. import ops

NONLIN_NAMES = ('relu', 'leakyrelu', 'elu', 'selu')


def act_name(nonlin):
    """Accepts the reference's nn.Module classes (nn.ELU ...) or our string ids."""
    if isinstance(nonlin, str):
        if nonlin not in NONLIN_NAMES:
            raise KeyError(nonlin)
        return nonlin
    table = {nn.ReLU: 'relu', nn.LeakyReLU: 'leakyrelu', nn.ELU: 'elu', nn.SELU: 'selu'}
    if nonlin in table:
        return table[nonlin]
    raise KeyError(nonlin)


class Placeholder(nn.Module):
    """Parameter-free slot that keeps the reference's nn.Sequential numbering (activation, dropout, interpolate)."""

    def __init__(self, what=''):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return self.what


class Conv2dParams(nn.Module):
    """Parameters of one nn.Conv2d / nn.ConvTranspose2d call site: same names, shapes and default initialisation
    (kaiming_uniform(a=sqrt(5)) + uniform bias, drawn in the same order from the global generator) as torch's."""

    def __init__(self, c_in, c_out, kernel, stride=1, padding=0, transposed=False, output_padding=0):
        super().__init__()
        self.c_in, self.c_out, self.kernel = c_in, c_out, kernel
        self.stride, self.padding, self.transposed, self.output_padding = stride, padding, transposed, output_padding
        shape = (c_in, c_out, kernel, kernel) if transposed else (c_out, c_in, kernel, kernel)
        self.weight = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(c_out))
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
        if fan_in != 0:
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)
        self._geom = None
        self._geom_key = None

    def geom(self):
        key = (self.weight.data_ptr(), tuple(self.weight.stride()))
        if self._geom_key != key:
            self._geom = K.ConvGeom(self.weight, self.stride, self.padding, self.transposed, self.output_padding)
            self._geom_key = key
        return self._geom

    def forward(self, x, x2=None, out_act=None):
        return ops.conv(x, self, x2=x2, out_act=out_act)

    def extra_repr(self):
        return '%d->%d k%d s%d p%d%s' % (self.c_in, self.c_out, self.kernel, self.stride, self.padding,
                                          ' transposed' if self.transposed else '')


class BatchNorm2dParams(nn.Module):
    """Parameters and buffers of nn.BatchNorm2d (momentum 0.1, eps 1e-5). `num_batches_tracked` is counted on the
    host and flushed into the buffer when the state dict is read (one tiny kernel per BN per step otherwise)."""

    def __init__(self, channels, eps=1e-5, momentum=0.1):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.register_buffer('running_mean', torch.zeros(channels))
        self.register_buffer('running_var', torch.ones(channels))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        self._pending = 0
        self._register_state_dict_hook(BatchNorm2dParams._flush_hook)

    @staticmethod
    def _flush_hook(module, state_dict, prefix, local_metadata):
        if module._pending:
            module._buffers['num_batches_tracked'] += module._pending
            module._pending = 0
            state_dict[prefix + 'num_batches_tracked'] = module._buffers['num_batches_tracked']

    def note_training_forward(self, n=1):
        self._pending += n


class GateLayer2d(nn.Module):
    """lib/nn.py:108-126: conv C -> 2C, then nonlin(first half) * sigmoid(second half)."""

    def __init__(self, channels, kernel_size, nonlin=nn.LeakyReLU):
        super().__init__()
        assert kernel_size % 2 == 1
        pad = kernel_size // 2
        self.conv = Conv2dParams(channels, 2 * channels, kernel_size, padding=pad)
        self.act = act_name(nonlin)
        self.nonlin = Placeholder(self.act)

    def forward(self, x):
        return ops.GateFn.apply(self.conv(x), None, self.act)



class ResidualBlock(nn.Module):
    """lib/nn.py:5-99.  out = gate(f(x)) + x with f one of the recipes 'cabdcabd', 'bacdbac', 'bacdbacd'
    (a = activation, b = batch norm, c = conv, d = dropout)."""

    default_kernel_size = (3, 3)

    def __init__(self, channels, nonlin, kernel=None, groups=1, batchnorm=True, block_type=None, dropout=None,
                 gated=None):
        super().__init__()
        if kernel is None:
            kernel = self.default_kernel_size
        elif isinstance(kernel, int):
            kernel = (kernel, kernel)
        elif len(kernel) != 2:
            raise ValueError("kernel has to be None, int, or an iterable of length 2")
        assert all([k % 2 == 1 for k in kernel]), "kernel sizes have to be odd"
        if groups != 1:
            raise NotImplementedError("grouped convolutions are not used by the LVAE hot path (groups=%r)" % (groups,))
        kernel = list(kernel)
        pad = [k // 2 for k in kernel]
        self.gated = gated
        self.block_type = block_type
        self.act = act_name(nonlin)
        self.dropout = dropout
        self.channels = channels

        modules, kinds = [], []

        def add(kind, mod):
            kinds.append(kind)
            modules.append(mod)

        if block_type == 'cabdcabd':
            for i in range(2):
                add('conv', Conv2dParams(channels, channels, kernel[i], padding=pad[i]))
                add('act', Placeholder(self.act))
                if batchnorm:
                    add('bn', BatchNorm2dParams(channels))
                if dropout is not None:
                    add('drop', Placeholder('dropout2d p=%s' % dropout))
        elif block_type == 'bacdbac':
            for i in range(2):
                if batchnorm:
                    add('bn', BatchNorm2dParams(channels))
                add('act', Placeholder(self.act))
                add('conv', Conv2dParams(channels, channels, kernel[i], padding=pad[i]))
                if dropout is not None and i == 0:
                    add('drop', Placeholder('dropout2d p=%s' % dropout))
        elif block_type == 'bacdbacd':
            for i in range(2):
                if batchnorm:
                    add('bn', BatchNorm2dParams(channels))
                add('act', Placeholder(self.act))
                add('conv', Conv2dParams(channels, channels, kernel[i], padding=pad[i]))
                if dropout is None:
                    # the reference builds nn.Dropout2d(None) here (lib/nn.py:89), which raises
                    raise TypeError("residual block 'bacdbacd' needs a dropout probability, got None")
                add('drop', Placeholder('dropout2d p=%s' % dropout))
        else:
            raise ValueError("unrecognized block type '{}'".format(block_type))

        if gated:
            add('gate', GateLayer2d(channels, 1, nonlin))
        self.block = nn.ModuleList(modules)  # same indices / parameter names as the reference's nn.Sequential
        self._kinds = kinds

        convs = [m for k, m in zip(kinds, modules) if k == 'conv']
        bns = [m for k, m in zip(kinds, modules) if k == 'bn']
        self.__dict__['conv1'], self.__dict__['conv2'] = convs
        self.__dict__['bn1'], self.__dict__['bn2'] = (bns if batchnorm else (None, None))
        self.__dict__['gate'] = modules[-1].conv if gated else None
        if block_type == 'cabdcabd':
            self._drops = (dropout is not None, dropout is not None)
        elif block_type == 'bacdbac':
            self._drops = (dropout is not None, False)
        else:
            self._drops = (True, True)

    def _masks(self, x, noise):
        N, C = x.shape[0], x.shape[3]
        out = []
        for has in self._drops:
            if has and self.training and self.dropout > 0.0:
                out.append(noise.dropout_mask(N, C, self.dropout, x.device))
            else:
                out.append(None)
        return out

    def forward(self, x, noise):
        if self.block_type == 'cabdcabd':
            return self._forward_post_activation(x, noise)
        # masks are drawn in execution order: conv1's dropout, then conv2's (SURVEY.md §8c noise tape)
        m1, m2 = self._masks(x, noise)
        params = [p for p in self.parameters()]
        # BatchNorm partials of x, if the previous block's gate kernel produced them: they travel as an attribute of exactly that
        # tensor object (a view, a copy or any other tensor does not carry them)
        ent = getattr(x, '_lvae_bn_parts', None)
        if ent is not None and self.training:
            self.__dict__['_in_parts'] = ent
        # the residual block that produced exactly this tensor object (its only consumer is this block): this block's last backward launch,
        # the BatchNorm-1 apply, can be left to that block's first backward launch (ops._DEFER_APPLY)
        src = x.__dict__.pop('_lvae_rb_src', None) if hasattr(x, '__dict__') else None
        if src is not None and self.training and torch.is_grad_enabled():
            self.__dict__['_in_src'] = src
        self.__dict__.pop('_pending_apply', None)   # (left behind by a backward pass that was abandoned)
        self.__dict__['_accepts_deferred'] = None
        out = ops.ResBlockFn.apply(x, self, m1, m2, self.training, *params)
        self.__dict__.pop('_in_src', None)
        self.__dict__.pop('_in_parts', None)
        if self.training and torch.is_grad_enabled() and self.__dict__.get('_accepts_deferred'):
            out._lvae_rb_src = self
        oparts = self.__dict__.pop('_out_parts', None)
        if oparts is not None:
            out._lvae_bn_parts = oparts
        if self.training:
            for bn in (self.bn1, self.bn2):
                if bn is not None:
                    bn.note_training_forward()
        return out

    def _forward_post_activation(self, x, noise):
        """'cabdcabd' (lib/nn.py:50-62): conv -> act -> [BN] -> [dropout], twice. Not on the default path: composed
        from single-op nodes (conv with the activation in its epilogue, then a materialised BN*mask)."""
        h = x
        for i, (cv, bn) in enumerate(((self.conv1, self.bn1), (self.conv2, self.bn2))):
            h = ops.conv(h, cv, out_act=self.act)
            mask = None
            if self._drops[i] and self.training and self.dropout > 0.0:
                mask = noise.dropout_mask(h.shape[0], h.shape[3], self.dropout, h.device)
            if bn is not None or mask is not None:
                params = [] if bn is None else [bn.weight, bn.bias]
                h = ops.BnDropFn.apply(h, bn, mask, self.training, *params)
                if bn is not None and self.training:
                    bn.note_training_forward()
        if self.gate is not None:
            return ops.GateFn.apply(ops.conv(h, self.gate), x, self.act)
        return ops.AddFn.apply(h, x)


class ResidualGatedBlock(ResidualBlock):
    """lib/nn.py:102-105."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs, gated=True)
