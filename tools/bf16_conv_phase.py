"""The bf16 3x3 convolution as the residual blocks of the bf16 step run it (256x16x16x64, bf16-stored x and y, BatchNorm + ELU prologue,
Dropout2d mask and BatchNorm partials in the epilogue), 20 back-to-back launches in a hipGraph. With a -DLVAE_PHASE_DEBUG library
(tools/phase_run.sh builds one) LVAE_BF16_DEBUG skips phases: 1 halo staging, 2 epilogue, 4 MFMAs, 8 weight fragments.
python tools/bf16_conv_phase.py [library]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import lvae_amd  # noqa: F401
from lvae_amd import _C
if len(sys.argv) > 1:
    _C.LIB_PATH = os.path.abspath(sys.argv[1])
from lvae_amd import kernels as K
from rb_bench import timeit, packed

K.set_precision('bf16')
B, H, C = 256, 16, 64
x = torch.randn(B, H, H, C, device='cuda')
w = packed(C, C, 3)
g = K.ConvGeom(w, 1, 1)
b = torch.randn(C, device='cuda')
sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.3
drop = (torch.rand(B, C, device='cuda') < 0.8).float() / 0.8
piv = torch.zeros(C, device='cuda')
xb = x.bfloat16()
tag = 'debug=%s' % os.environ.get('LVAE_BF16_DEBUG', '0')
K.conv2d(xb, w, g, bias=b, out_bf16=True)
K.prepared.prepare_all()
print(tag, 'plain, bf16 in/out                  %6.1f us' % timeit(lambda: K.conv2d(xb, w, g, bias=b, out_bf16=True)))
print(tag, '+ BN+ELU prologue                   %6.1f us' % timeit(lambda: K.conv2d(xb, w, g, bias=b, in_scale=sc, in_shift=sh, in_act='elu', out_bf16=True)))
print(tag, '+ prologue, mask, statistics        %6.1f us' % timeit(lambda: K.conv2d(xb, w, g, bias=b, in_scale=sc, in_shift=sh, in_act='elu', out_scale=drop, stats_pivot=piv, out_bf16=True)))
print(tag, 'fp32 in/out, prologue, mask, stats  %6.1f us' % timeit(lambda: K.conv2d(x, w, g, bias=b, in_scale=sc, in_shift=sh, in_act='elu', out_scale=drop, stats_pivot=piv)))
