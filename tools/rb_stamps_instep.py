"""Phase stamps of the LAST fused residual-block launch of a captured training step (needs the -DLVAE_RB_DBG library of tools/rb_stamps.sh):
in the step the operands and weights of a launch are cold, unlike in tools/rb_stamps.py's back-to-back launches of one layer.
python tools/rb_stamps_instep.py <path of lib_rb.so>"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lvae_amd  # noqa: F401
from lvae_amd import _C
_C.LIB_PATH = sys.argv[1]
from lvae_amd.configs import CIFAR15, synthetic_images
from lvae_amd.engine import TrainStep
from lvae_amd.models.lvae import LadderVAE
from lvae_amd.noise import PhiloxNoise
from lvae_amd.optim import Adamax

torch.manual_seed(42)
model = LadderVAE(**CIFAR15).cuda().train()
model.noise = PhiloxNoise(seed=42)
model.pack()
step = TrainStep(model, Adamax(model, lr=3e-4), use_graph=True)
x = synthetic_images(CIFAR15, 256, torch.Generator().manual_seed(1)).cuda()
for _ in range(6):
    step(x)
torch.cuda.synchronize()
lib = ctypes.CDLL(sys.argv[1])
buf = np.zeros(1024 * 4 * 12, dtype=np.uint64)
assert lib.lvae_debug_rb_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
s = buf.reshape(1024 * 4, 12).astype(np.int64)
# the last fused launch of the backward pass: BatchNorm-apply + dgrad of the first 8x8 bottom-up block (256 workgroups); its stamps
# overwrote everything earlier launches left in rows 0..1023
s = s[:256 * 4]
seq = [0, 8, 9, 1, 2, 3, 4, 5, 6, 7]
nm = ['0-8 requests; partial rows arrived and summed per thread', '8-9 LDS write + barrier', '9-1 combine in double, finalize, barrier', '1-2 apply, store, patch, operand requests', '2-3 barrier', '3-4 3x3 reduction loop', '4-5 barrier, partial tiles, barrier',
      '5-6 epilogue', '6-7 statistics rows']
print('last fused launch of the step (bn-apply + dgrad, 8x8, 256 workgroups), in-step: wave start spread %d ticks, first start -> last end %d ticks' %
      (s[:, 0].max() - s[:, 0].min(), s[:, 7].max() - s[:, 0].min()))
for i in range(len(seq) - 1):
    d = s[:, seq[i + 1]] - s[:, seq[i]]
    print('   %-50s median %6d  p10 %6d  p90 %6d' % (nm[i], np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
print('   total per wave: median %d' % np.median(s[:, 7] - s[:, 0]))
