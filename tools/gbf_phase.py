"""Times the persistent gate backward (lvae_conv1x1_gate_bwd_wgrad_f32: gate derivative + 1x1 dgrad + 1x1 weight gradient, optionally with the
deferred BatchNorm-backward apply in front) at 256 x H x H x 64 through a given library: python tools/gbf_phase.py <H> <liblvae_hip.so> [ap].
With a -DLVAE_GBF_DBG=<mask> build (tools/gbf_ab.sh) the difference to the unmasked build is the cost of the skipped phase."""
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
AP = len(sys.argv) > 3 and sys.argv[3] == 'ap'
N, C = 256, 64
rn = lambda *s: torch.randn(*s, device='cuda')
x, dh, add, ab, y2, dout = rn(N, H, H, C), rn(N, H, H, C), rn(N, H, H, C), rn(N, H, H, 2 * C), rn(N, H, H, C), rn(N, H, H, C)
wg = packed(2 * C, C, 1)
geg = K.ConvGeom(wg, 1, 0)
coef = K.bn_stats(x, None, None, None, None)
parts = torch.randn(256, 2, C, device='cuda')
dgam, dbet = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
out = torch.empty_like(x)
dw, db = torch.zeros_like(wg), torch.zeros(2 * C, device='cuda')


def run():
    if AP:
        pend = K.PendingApply(parts, dh, x, coef[0], 'elu', dgam, dbet, add, out)
        return K.conv1x1_gate_bwd_wgrad(out, ab, y2, wg, geg, 'elu', dw, db, apply=pend)
    return K.conv1x1_gate_bwd_wgrad(dout, ab, y2, wg, geg, 'elu', dw, db)


assert run() is not None
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(20):
        run()
gr.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    gr.replay()
e1.record()
torch.cuda.synchronize()
print('debug gate backward %dx%d%s: %.1f us per launch (kernel + its slab reduce)' % (H, H, ' + deferred apply' if AP else '', e0.elapsed_time(e1) * 1e3 / 100))
