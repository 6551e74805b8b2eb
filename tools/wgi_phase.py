"""Times lvae_conv2d_wgrad_f32 on the whole-image-tile kernel (conv_wgrad_img.hip) for ONE large gradient, N x H x H x 64 -> 64 (default
1024 x 8 x 8 = 1024 tiles: the pixel count of a 256 x 16 x 16 layer), through a given library: python tools/wgi_phase.py <H> <N> <lib>.
With a -DLVAE_WGI_DBG=<mask> build (tools/wgi_ab.sh) the difference to the unmasked build is the cost of the skipped phase."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import lvae_amd  # noqa: F401
from lvae_amd import _C
if len(sys.argv) > 3:
    _C.LIB_PATH = os.path.abspath(sys.argv[3])
from lvae_amd import kernels as K
from conv_bench import packed

H = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
C = 64
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
print('debug wgrad_img %dx%dx%d: %.1f us per gradient (slab kernel + reduce), tiles per workgroup %s' % (B, H, H, best, os.environ.get('LVAE_WGRAD_IMG_TPW', 'default')))
