"""Single-launch timings of the 3x3 forms: Winograd fp32 (LVAE_F32_SPLIT=0), six-product bf16 split, bf16 operands."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvae_amd  # noqa: F401
from lvae_amd import kernels as K
from conv_bench import packed, timeit

B = 256
for H in (16, 32, 8):
    C = 64
    x = torch.randn(B, H, H, C, device='cuda')
    dy = torch.randn(B, H, H, C, device='cuda')
    w = packed(C, C, 3)
    g = K.ConvGeom(w, 1, 1)
    sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
    drop = (torch.rand(B, C, device='cuda') < 0.8).float() / 0.8
    b = torch.randn(C, device='cuda')
    piv = torch.zeros(C, device='cuda')
    for name, env, prec in (('winograd f32', '0', 'f32'), ('split-6 f32 ', '1', 'f32'), ('bf16        ', '1', 'bf16')):
        os.environ['LVAE_F32_SPLIT'] = env
        K.set_precision(prec)
        t_f = timeit(lambda: K.conv2d(x, w, g, bias=b, in_scale=sc, in_shift=sh, in_act='elu', out_scale=drop, stats_pivot=piv))
        t_p = timeit(lambda: K.conv2d(x, w, g, bias=b))
        t_d = timeit(lambda: K.conv2d_dgrad(dy, w, g, (H, H)))
        print('3x3 %2dx%-2d %s: fwd+bn/elu/drop/stats %7.1f us | fwd plain %7.1f | dgrad %7.1f' % (H, H, name, t_f, t_p, t_d), flush=True)
    K.set_precision('f32')
