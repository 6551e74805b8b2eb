"""SQ counters of the dominant kernel (conv3x3_wino_kernel<64, 2, 1> at 256x16x16): run under
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import lvae_amd  # noqa: F401
from lvae_amd import kernels as K
from conv_bench import packed

B, C, H = 256, 64, int(os.environ.get("PMC_H", "16"))
x = torch.randn(B, H, H, C, device='cuda')
w = packed(C, C, 3)
g = K.ConvGeom(w, 1, 1)
b = torch.randn(C, device='cuda')
sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
for _ in range(12):
    K.conv2d(x, w, g, bias=b, in_scale=sc, in_shift=sh, in_act='elu')
torch.cuda.synchronize()
