"""Phase-skip timing of conv3x3_wino_kernel<64, 2, 1, true> (needs a -DLVAE_PHASE_DEBUG build; LVAE_WINO_DEBUG is read once per process):
python tools/wino_phase.py <H> <path of the debug liblvae_hip.so>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import lvae_amd  # noqa: F401
from lvae_amd import _C
if len(sys.argv) > 2:
    _C.LIB_PATH = sys.argv[2]
from lvae_amd import kernels as K
from conv_bench import packed, timeit

H = int(sys.argv[1])
B, C = int(os.environ.get('PHASE_B', '256')), 64
x = torch.randn(B, H, H, C, device='cuda')
w = packed(C, C, 3)
g = K.ConvGeom(w, 1, 1)
b = torch.randn(C, device='cuda')
sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
drop = (torch.rand(B, C, device='cuda') < 0.8).float() / 0.8
piv = torch.zeros(C, device='cuda')
K.conv2d(x, w, g, bias=b)
K.prepared.prepare_all()
t_p = timeit(lambda: K.conv2d(x, w, g, bias=b), 200)
t_f = timeit(lambda: K.conv2d(x, w, g, bias=b, in_scale=sc, in_shift=sh, in_act='elu', out_scale=drop, stats_pivot=piv), 200)
print('B=%d ' % B + '%dx%d wino debug=%s: plain %.1f us, fused prologue/epilogue/stats %.1f us' % (H, H, os.environ.get('LVAE_WINO_DEBUG', '0'), t_p, t_f))
