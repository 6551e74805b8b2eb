"""SQ counters of conv_wgrad_wino_kernel<8> at 256x16x16 (see tools/wino_pmc.sh for the counter list)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import lvae_amd  # noqa: F401
from lvae_amd import kernels as K
from conv_bench import packed

B, C, H = 256, 64, 16
x = torch.randn(B, H, H, C, device='cuda')
dy = torch.randn(B, H, H, C, device='cuda')
w = packed(C, C, 3)
g = K.ConvGeom(w, 1, 1)
dw, db = torch.zeros_like(w), torch.zeros(C, device='cuda')
sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
for _ in range(12):
    K.conv2d_wgrad(x, dy, w, g, dw, db, in_scale=sc, in_shift=sh, in_act='elu')
torch.cuda.synchronize()
