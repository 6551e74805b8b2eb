"""In-kernel phase stamps of rb_conv_kernel (needs the -DLVAE_RB_DBG library built by tools/rb_stamps.sh):
python tools/rb_stamps.py <H> <path of lib_rb.so>  -> median s_memtime ticks per phase over the waves of one launch"""
import ctypes
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lvae_amd  # noqa: F401
from lvae_amd import _C
_C.LIB_PATH = sys.argv[2]
from lvae_amd import kernels as K

H = int(sys.argv[1])
B, C, dev = 256, 64, 'cuda'
packed = lambda co, ci, k: torch.randn(k, k, ci, co, device=dev).permute(3, 2, 0, 1) * 0.05
x, dout = torch.randn(B, H, H, C, device=dev), torch.randn(B, H, H, C, device=dev)
w1, w2, wg = packed(C, C, 3), packed(C, C, 3), packed(2 * C, C, 1)
g1, g2, gg = K.ConvGeom(w1, 1, 1), K.ConvGeom(w2, 1, 1), K.ConvGeom(wg, 1, 0)
b1, b2, bg = torch.randn(C, device=dev), torch.randn(C, device=dev), torch.randn(2 * C, device=dev)
m1 = (torch.rand(B, C, device=dev) < 0.8).float() / 0.8
m2 = (torch.rand(B, C, device=dev) < 0.8).float() / 0.8
mk = lambda: types.SimpleNamespace(weight=torch.ones(C, device=dev), bias=torch.zeros(C, device=dev), running_mean=torch.zeros(C, device=dev),
                                   running_var=torch.ones(C, device=dev), eps=1e-5, momentum=0.1)
bn1, bn2 = mk(), mk()
coef1 = K.bn_stats(x, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
st = {}


def conv1():
    st['y1'], st['p2'], st['c1'] = K.rb_conv(x, w1, g1, b1, 'elu', m1, coef=coef1, stats_pivot=bn2.running_mean)


def conv2_gate():
    y2, st['ab'], out, op, st['c2'] = K.rb_conv_gate(st['y1'], w2, g2, b2, 'elu', m2, wg, gg, bg, x, 'elu', in_bn=(st['p2'], bn2.running_mean, bn2),
                                                     stats_pivot=st['c1'][2])


def gate_dgrad():
    _, _, st['dh2'], st['bp2'] = K.rb_gate_dgrad(dout, st['ab'], wg, gg, 'elu', m2, w2, g2, bn_bwd=(st['y1'], st['c2'][0], 'elu'))


def apply_dgrad():
    K.rb_apply_dgrad(st['bp2'], st['dh2'], st['y1'], st['c2'][0], 'elu', dg, db, m1, w1, g1, bn_bwd=(x, st['c1'][0], 'elu'))


for fn in (conv1, conv2_gate, gate_dgrad, apply_dgrad):
    fn()
K.prepared.prepare_all()
lib = ctypes.CDLL(sys.argv[2])
nwg = min(1024, K.rb_rows(x, w1, g1))
names = ['0-1 loads + statistics fold', '1-2 staging (act, split, LDS) + ring', '2-3 barrier', '3-4 3x3 reduction loop', '4-5 barrier, partial tiles, barrier',
         '5-6 epilogue (sum, bias, mask, stores [, gate GEMM + stores])', '6-7 statistics rows']
for name, fn in (('conv1 (plain, given coefficients)', conv1), ('conv2 + gate (folded finalize)', conv2_gate), ('gate-bwd + dgrad', gate_dgrad),
                 ('bn-apply + dgrad', apply_dgrad)):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 4 * 12, dtype=np.uint64)
    assert lib.lvae_debug_rb_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
    s = buf.reshape(1024 * 4, 12)[:nwg * 4].astype(np.int64)
    t0 = s[:, 0].min()
    print('%s %dx%d (%d workgroups): launch spans %d ticks (first wave start -> last end stamp); wave start spread %d' % (
        name, H, H, nwg, s[:, 7].max() - t0, s[:, 0].max() - t0))
    if name.startswith('gate-bwd'):
        seq = [0, 8, 9, 2, 3, 4, 5, 6, 7]
        nm = ['0-8 loads, gate derivative, dab planes', '8-9 barrier + 1x1 dgrad GEMM', '9-2 dy2 staging, mask, store, patch + ring'] + names[2:]
    elif name.startswith('bn-apply'):
        seq = [0, 2, 3, 4, 5, 6, 7]
        nm = ['0-2 loads, partial-row sums, apply, store, patch + ring'] + names[2:]
    else:
        seq = [0, 1, 10, 11, 2, 3, 4, 5, 6, 7]
        nm = [names[0], '1-10 act + split + interior writes', '10-11 zero ring', '11-2 epilogue operand requests, B ring, accumulator init'] + names[2:]
    for i in range(len(seq) - 1):
        dlt = s[:, seq[i + 1]] - s[:, seq[i]]
        print('   %-66s median %6d  p10 %6d  p90 %6d' % (nm[i], np.median(dlt), np.percentile(dlt, 10), np.percentile(dlt, 90)))
    print('   total per wave: median %d' % np.median(s[:, 7] - s[:, 0]))
