"""What do the statistics epilogues of the 256-pixel Winograd kernel cost? forward / dgrad launches at 256x16x16 with and without them,
captured in a hipGraph (20 back-to-back launches, hot caches). Profiling helper."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import lvae_amd  # noqa: F401
from lvae_amd import kernels as K
from rb_bench import timeit, packed

B, H, C = 256, 16, 64
x = torch.randn(B, H, H, C, device='cuda')
dy = torch.randn(B, H, H, C, device='cuda')
w = packed(C, C, 3)
g = K.ConvGeom(w, 1, 1)
b = torch.randn(C, device='cuda')
sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
drop = (torch.rand(B, C, device='cuda') < 0.8).float() / 0.8
piv = torch.zeros(C, device='cuda')
coef = K.bn_stats(x, sc, sh, None, None)
K.conv2d(x, w, g, bias=b)
K.conv2d_dgrad(dy, w, g, (H, H))
K.prepared.prepare_all()
print('forward  plain                         %6.1f us' % timeit(lambda: K.conv2d(x, w, g, bias=b)))
print('forward  BN+ELU prologue, mask         %6.1f us' % timeit(lambda: K.conv2d(x, w, g, bias=b, in_scale=sc, in_shift=sh, in_act='elu', out_scale=drop)))
print('forward  ... + output statistics       %6.1f us' % timeit(lambda: K.conv2d(x, w, g, bias=b, in_scale=sc, in_shift=sh, in_act='elu', out_scale=drop, stats_pivot=piv)))
print('dgrad    plain                         %6.1f us' % timeit(lambda: K.conv2d_dgrad(dy, w, g, (H, H))))
print('dgrad    + BatchNorm-backward sums     %6.1f us' % timeit(lambda: K.conv2d_dgrad(dy, w, g, (H, H), bn_bwd=(x, coef[0], 'elu'))))
