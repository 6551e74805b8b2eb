"""In-memory stand-ins for the five `boilr==0.7.4` symbols the reference model imports.

TEST INFRASTRUCTURE ONLY (see oracle/README.md). `boilr` (requirements.txt:6 of the reference) is not
installed in this container and cannot be fetched, so the reference's `models/lvae.py:3-4` cannot import
without these. They are OUR restatement of boilr's semantics, inferred from the reference's call sites
(models/lvae.py:15,144,176->324,185,197,357): **parity unpinned** at this boundary.

Only `oracle/gen_golden.py` (which runs in the build container where /root/reference exists) installs them.
"""
import sys
import types

import torch
from torch import nn
import torch.nn.functional as F


class BaseGenerativeModel(nn.Module):
    """boilr.models.BaseGenerativeModel: an nn.Module that carries a `global_step` counter."""

    def __init__(self):
        super().__init__()
        self.global_step = 0


def _centre_offsets(cur, tgt):
    d = tgt - cur
    lo = d // 2
    return lo, d - lo


def pad_img_tensor(x, size):
    """Zero-pad the two trailing dims of `x` to `size`, centred (floor(delta/2) before, rest after)."""
    h0, h1 = _centre_offsets(x.shape[2], int(size[0]))
    w0, w1 = _centre_offsets(x.shape[3], int(size[1]))
    return F.pad(x, (w0, w1, h0, h1))


def crop_img_tensor(x, size):
    """Inverse of pad_img_tensor: centre crop of the two trailing dims."""
    h0, _ = _centre_offsets(int(size[0]), x.shape[2])
    w0, _ = _centre_offsets(int(size[1]), x.shape[3])
    return x[:, :, h0:h0 + int(size[0]), w0:w0 + int(size[1])]


class Interpolate(nn.Module):
    """boilr.nn.Interpolate(scale=2): bilinear, align_corners=False."""

    def __init__(self, size=None, scale=None, mode='bilinear', align_corners=False):
        super().__init__()
        self.size, self.scale, self.mode, self.align_corners = size, scale, mode, align_corners

    def forward(self, x):
        return F.interpolate(x, size=self.size, scale_factor=self.scale, mode=self.mode,
                             align_corners=self.align_corners)


def free_bits_kl(kl, free_bits, batch_average=False, eps=1e-6):
    """kl (B, L) -> (L,): batch mean of the per-sample, per-layer KL clamped from below at free_bits."""
    assert kl.dim() == 2
    if free_bits < eps:
        return kl.mean(0)
    if batch_average:
        return kl.mean(0, keepdim=True).clamp(min=free_bits).squeeze(0)
    return kl.clamp(min=free_bits).mean(0)


def install():
    """Register the stand-ins as `boilr`, `boilr.models`, `boilr.nn` in sys.modules."""
    boilr = types.ModuleType('boilr')
    models = types.ModuleType('boilr.models')
    bnn = types.ModuleType('boilr.nn')
    models.BaseGenerativeModel = BaseGenerativeModel
    bnn.pad_img_tensor = pad_img_tensor
    bnn.crop_img_tensor = crop_img_tensor
    bnn.Interpolate = Interpolate
    bnn.free_bits_kl = free_bits_kl
    boilr.models, boilr.nn = models, bnn
    sys.modules['boilr'] = boilr
    sys.modules['boilr.models'] = models
    sys.modules['boilr.nn'] = bnn
