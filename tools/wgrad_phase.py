"""Times lvae_conv2d_wgrad_f32 (Winograd weight gradient: slab kernel + slab reduce) at 256 x H x H x 64 through a given library:
python tools/wgrad_phase.py <H> <liblvae_hip.so>. With a -DLVAE_WGW_DBG=<mask> build (tools/wgrad_ab.sh) the difference to the
unmasked build is the cost of the skipped phase. Measurement tooling only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import lvae_amd  # noqa: F401
from lvae_amd import _C
if len(sys.argv) > 2:
    _C.LIB_PATH = os.path.abspath(sys.argv[2])
from lvae_amd import kernels as K
from conv_bench import packed

H = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B, C = 256, 64
x = torch.randn(B, H, H, C, device='cuda')
dy = torch.randn(B, H, H, C, device='cuda')
w = packed(C, C, 3)
g = K.ConvGeom(w, 1, 1)
dw, db = torch.zeros_like(w), torch.zeros(C, device='cuda')
sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')


def run():
    K.conv2d_wgrad(x, dy, w, g, dw, db, in_scale=sc, in_shift=sh, in_act='elu')


for _ in range(5):
    run()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(20):
        run()
gr.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(5):
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20 * 1000)
print('debug wgrad %dx%d: %.1f us per gradient (slab kernel + reduce)' % (H, H, best))
