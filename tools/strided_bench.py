"""Isolated timing of the strided 3x3 (stride 2) convolutions and their dgrads (conv_igemm_kernel): python tools/strided_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import lvae_amd  # noqa: F401
from lvae_amd import kernels as K
from conv_bench import packed, timeit

B, C = 256, 64
for H in (32, 16, 8, 4):
    x = torch.randn(B, H, H, C, device='cuda')
    w = packed(C, C, 3)
    g = K.ConvGeom(w, 2, 1)
    b = torch.randn(C, device='cuda')
    y = K.conv2d(x, w, g, bias=b)
    dy = torch.randn_like(y)
    t_f = timeit(lambda: K.conv2d(x, w, g, bias=b))
    t_d = timeit(lambda: K.conv2d_dgrad(dy, w, g, (H, H)))
    fl = 2.0 * B * (H // 2) ** 2 * C * C * 9
    print('3x3 s2 %2dx%-2d -> %2dx%-2d: fwd %6.1f us | dgrad %6.1f us | fp32-MFMA floor %5.1f us' % (H, H, H // 2, H // 2, t_f, t_d, fl / 157.3e12 * 1e6))
