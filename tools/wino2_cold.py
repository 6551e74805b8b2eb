"""How much of a 16x16 Winograd launch inside the step is cold operands? A chain y = conv(y, w_i) as the model runs it (every input was just
written by the launch before), with ONE weight set (U pieces hot in every L2) or L different ones (cold, as in the step), and the same for
the input. python tools/wino2_cold.py [H]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import lvae_amd  # noqa: F401
from lvae_amd import kernels as K
from conv_bench import packed

H = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B, C, L = 256, 64, 48
ws = [packed(C, C, 3) for _ in range(L)]
for w in ws:
    w.mul_(0.2)
g = K.ConvGeom(ws[0], 1, 1)
b = torch.zeros(C, device='cuda')
sc, sh = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
piv = torch.zeros(C, device='cuda')
xs = [torch.randn(B, H, H, C, device='cuda') for _ in range(L)]
for w in ws:
    K.conv2d(xs[0], w, g, bias=b)
K.prepared.prepare_all()


def run(chain, rot_w, rot_x, reps=6):
    def body():
        y = xs[0]
        for i in range(L):
            src = y if chain else (xs[i] if rot_x else xs[0])
            y = K.conv2d(src, ws[i if rot_w else 0], g, bias=b, in_scale=sc, in_shift=sh, in_act='elu', stats_pivot=piv)[0]
    body()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        body()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * L)


for name, args in (('same x, same w', (False, False, False)), ('same x, L weights', (False, True, False)), ('L inputs, same w', (False, False, True)),
                   ('L inputs, L weights', (False, True, True)), ('chain, same w', (True, False, False)), ('chain, L weights', (True, True, False))):
    print('%dx%d %-22s %.2f us per launch' % (H, H, name, run(*args)))
