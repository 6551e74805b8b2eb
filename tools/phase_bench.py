"""Phase-skip timing of conv3x3_bf16 (needs a -DLVAE_PHASE_DEBUG build: make EXTRA=-DLVAE_PHASE_DEBUG); LVAE_BF16_DEBUG is read once
per process, so each configuration runs in its own process: python tools/phase_bench.py <H> <prec> [path of the debug liblvae_hip.so]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import lvae_amd  # noqa: F401
from lvae_amd import _C
if len(sys.argv) > 3:
    _C.LIB_PATH = sys.argv[3]   # the debug build made by tools/phase_run.sh in a scratch directory
from lvae_amd import kernels as K
from conv_bench import packed, timeit

H, prec = int(sys.argv[1]), sys.argv[2]
B, C = 256, 64
x = torch.randn(B, H, H, C, device='cuda')
w = packed(C, C, 3)
g = K.ConvGeom(w, 1, 1)
b = torch.randn(C, device='cuda')
K.set_precision(prec)
t = timeit(lambda: K.conv2d(x, w, g, bias=b))
print('%dx%d %s debug=%s: %.1f us' % (H, H, prec, os.environ.get('LVAE_BF16_DEBUG', '0'), t))
