"""Time the Winograd 3x3 kernel alone (weight transform excluded via HIP-event bracketing of repeated launches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvae_amd  # noqa: F401
from lvae_amd import kernels as K
from conv_bench import packed, timeit


def main():
    B, C = 256, 64
    for H in (16, 32):
        x = torch.randn(B, H, H, C, device='cuda')
        w = packed(C, C, 3)
        g = K.ConvGeom(w, 1, 1)
        sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
        drop = (torch.rand(B, C, device='cuda') < 0.8).float() / 0.8
        b = torch.randn(C, device='cuda')
        t_f = timeit(lambda: K.conv2d(x, w, g, bias=b, in_scale=sc, in_shift=sh, in_act='elu', out_scale=drop), 200)
        t_p = timeit(lambda: K.conv2d(x, w, g, bias=b), 200)
        print('%dx%d: fused %.1f us  plain %.1f us (each includes the ~5 us weight transform launch)' % (H, H, t_f, t_p))


if __name__ == '__main__':
    main()
