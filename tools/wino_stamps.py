"""In-kernel phase stamps of conv3x3_wino_kernel (needs the -DLVAE_WINO_DBG=64 library built by tools/wino_phase.sh):
python tools/wino_stamps.py <H> <path of lib_64.so>  -> median cycles per phase over the waves of one launch"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import lvae_amd  # noqa: F401
from lvae_amd import _C
_C.LIB_PATH = sys.argv[2]
from lvae_amd import kernels as K
from conv_bench import packed

H = int(sys.argv[1])
B, C = 256, 64
x = torch.randn(B, H, H, C, device='cuda')
w = packed(C, C, 3)
g = K.ConvGeom(w, 1, 1)
b = torch.randn(C, device='cuda')
sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
drop = (torch.rand(B, C, device='cuda') < 0.8).float() / 0.8
piv = torch.zeros(C, device='cuda')
K.conv2d(x, w, g, bias=b)
K.prepared.prepare_all()
lib = ctypes.CDLL(sys.argv[2])
for name, fn in (('plain', lambda: K.conv2d(x, w, g, bias=b)),
                 ('fused', lambda: K.conv2d(x, w, g, bias=b, in_scale=sc, in_shift=sh, in_act='elu', out_scale=drop, stats_pivot=piv))):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    w2 = os.environ.get('LVAE_DISABLE_WINO2') != '1'
    nw = (B * H * H // 256) * (4 if os.environ.get('LVAE_STAMPS_4WAVES') == '1' else 8) if w2 else (B * H * H // 128) * 4
    nw = min(nw, 1024 * ((4 if os.environ.get('LVAE_STAMPS_4WAVES') == '1' else 8) if w2 else 4))
    buf = np.zeros(nw * 8, dtype=np.uint64)
    assert lib.lvae_debug_wino_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
    st = buf.reshape(nw, 8)[:, :(8 if w2 else 6)].astype(np.int64)
    d = np.diff(st, axis=1)
    names = (['prologue: index math, slice-0 fetch+stage, barrier', 'GEMM loop (transform, split, MFMA, slices 1-3)', 'barrier after the loop',
              'block 0: partial sums into LDS + barrier', 'block 0: store pass', 'block 1: barrier + partial sums + barrier', 'block 1: store pass'] if w2 else
             ['prologue: index math, slice-0 fetch+stage, barrier', 'GEMM loop (transform, split, MFMA, slices 1-3)', 'barrier after the loop',
              'R = M.A into LDS + barrier', 'store pass (A^T.R, bias/drop/act, stores, stats sums)'])
    t0 = st[:, 0].min()
    print('%s %dx%d: launch spans %d cycles (first wave start -> last stamp 5); per-wave medians:' % (name, H, H, st[:, -1].max() - t0))
    for i, nm in enumerate(names):
        print('   %-58s median %6d  p10 %6d  p90 %6d cycles' % (nm, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
    print('   wave start spread (p90 - p10 of stamp 0): %d cycles; total per wave median %d' % (
        np.percentile(st[:, 0], 90) - np.percentile(st[:, 0], 10), np.median(st[:, -1] - st[:, 0])))
