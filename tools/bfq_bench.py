"""bf16 weight gradient of a 3x3 64->64 layer at 256x16x16 (x, dy bf16-stored, BatchNorm + ELU prologue) in a hipGraph of 20 back-to-back
launches: python tools/bfq_bench.py [library]   (a scratch build with -DLVAE_BFH_DBG=<mask> skips phases: tools/bfq_phase.sh)."""
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
for (B, H) in ((256, 16), (64, 32)):
    C = 64
    x = torch.randn(B, H, H, C, device='cuda')
    dy = torch.randn(B, H, H, C, device='cuda')
    w = packed(C, C, 3)
    g = K.ConvGeom(w, 1, 1)
    sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
    dw, db = torch.zeros_like(w), torch.zeros(C, device='cuda')
    for name, xs, ds in (('x bf16 dy bf16', x.bfloat16(), dy.bfloat16()), ('x f32  dy bf16', x, dy.bfloat16()), ('x f32  dy f32 ', x, dy)):
        f = lambda: K.conv2d_wgrad(xs, ds, w, g, dw, db, in_scale=sc, in_shift=sh, in_act='elu')
        f()
        print('%dx%dx%d %s  kernel + reduce %6.1f us' % (B, H, H, name, timeit(f)))
