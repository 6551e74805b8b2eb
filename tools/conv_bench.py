"""Micro-benchmark of single conv launches (HIP events, 50 back-to-back launches each). Profiling helper."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvae_amd  # noqa: F401
from lvae_amd import kernels as K


def packed(co, ci, k):
    return torch.randn(k, k, ci, co, device='cuda').permute(3, 2, 0, 1) * 0.05


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B = 256
    rows = []
    for (H, C) in ((16, 64), (32, 64), (8, 64), (4, 64), (2, 64)):
        x = torch.randn(B, H, H, C, device='cuda')
        dy = torch.randn(B, H, H, C, device='cuda')
        w = packed(C, C, 3)
        g = K.ConvGeom(w, 1, 1)
        sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
        drop = (torch.rand(B, C, device='cuda') < 0.8).float() / 0.8
        b = torch.randn(C, device='cuda')
        dw, db = torch.zeros_like(w), torch.zeros(C, device='cuda')
        flops = 2.0 * B * H * H * C * C * 9
        t_f = timeit(lambda: K.conv2d(x, w, g, bias=b, in_scale=sc, in_shift=sh, in_act='elu', out_scale=drop))
        t_p = timeit(lambda: K.conv2d(x, w, g, bias=b))
        t_d = timeit(lambda: K.conv2d_dgrad(dy, w, g, (H, H)))
        t_w = timeit(lambda: K.conv2d_wgrad(x, dy, w, g, dw, db, in_scale=sc, in_shift=sh, in_act='elu'))
        ideal = flops / 157.3e12 * 1e6
        rows.append('3x3 %2dx%-2d C%d: fwd+bn/elu/drop %7.1f us | fwd plain %7.1f | dgrad %7.1f | wgrad %7.1f | mfma floor %6.1f us' %
                    (H, H, C, t_f, t_p, t_d, t_w, ideal))
    # 1x1 gate (64->128) and merge (128->64) at 16x16
    x = torch.randn(B, 16, 16, 64, device='cuda')
    wg = packed(128, 64, 1)
    gg = K.ConvGeom(wg, 1, 0)
    dab = torch.randn(B, 16, 16, 128, device='cuda')
    dwg, dbg = torch.zeros_like(wg), torch.zeros(128, device='cuda')
    rows.append('1x1 gate 16x16: fwd %7.1f us | dgrad %7.1f | wgrad %7.1f | mfma floor %5.1f, hbm floor ~%4.1f us' % (
        timeit(lambda: K.conv2d(x, wg, gg, bias=dbg)), timeit(lambda: K.conv2d_dgrad(dab, wg, gg, (16, 16))),
        timeit(lambda: K.conv2d_wgrad(x, dab, wg, gg, dwg, dbg)), 2.0 * B * 256 * 64 * 128 / 157.3e12 * 1e6,
        (16.8 + 33.5) / 5.0e3 * 1e3 / 1e3 * 1e0))
    print('\n'.join(rows))


if __name__ == '__main__':
    main()
