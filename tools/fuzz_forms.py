"""Random-shape cross-check of kernel forms that must agree to fp32 accuracy: Winograd position GEMMs on the fp32 MFMA vs six-product
(forward + dgrad), persistent gate forward vs the single-shot kernel, fused gate backward six-product vs fp32 MFMA."""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvae_amd  # noqa: F401
from lvae_amd import kernels as K

random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
torch.manual_seed(0)


def packed(co, ci, k):
    return (torch.randn(k, k, ci, co, device='cuda') / (ci * k * k) ** 0.5).permute(3, 2, 0, 1)


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-20))


worst = 0.0
for it in range(40):
    H = random.choice([8, 16, 32]); W = random.choice([8, 16, 32])
    N = random.randint(max(1, 16384 // (H * W)), max(2, 70000 // (H * W)))
    Co = random.choice([32, 64, 100, 128]); C = 64
    x = torch.randn(N, H, W, C, device='cuda'); w = packed(Co, C, 3); g = K.ConvGeom(w, 1, 1)
    b = torch.randn(Co, device='cuda'); sc = torch.rand(C, device='cuda') + 0.5; sh = torch.randn(C, device='cuda') * 0.3
    drop = (torch.rand(N, Co, device='cuda') < 0.8).float() / 0.8
    outs = []
    for form in ('0', '1'):
        os.environ['LVAE_WINO_SPLIT'] = form
        K.prepared.entries.clear(); K.prepared.table = None
        y = K.conv2d(x, w, g, bias=b, in_scale=sc, in_shift=sh, in_act='elu', out_scale=drop, out_act='elu')
        outs.append(y.clone())
    e = rel(outs[1], outs[0]); worst = max(worst, e)
    assert e < 4e-6, ('wino fwd', N, H, W, Co, e)
    if Co in (64, 128):
        dy = torch.randn(N, H, W, Co, device='cuda'); outs = []
        for form in ('0', '1'):
            os.environ['LVAE_WINO_SPLIT'] = form
            K.prepared.entries.clear(); K.prepared.table = None
            outs.append(K.conv2d_dgrad(dy, w, g, (H, W)).clone())
        e = rel(outs[1], outs[0]); worst = max(worst, e)
        assert e < 4e-6, ('wino dgrad', N, H, W, Co, e)
os.environ.pop('LVAE_WINO_SPLIT', None)
print('winograd forms: worst relative difference %.2e over 40 shapes' % worst)

worst = 0.0
for it in range(30):
    H = random.choice([2, 3, 4, 8, 16]); N = random.randint(1, max(2, 70000 // (H * H)))
    C = 64
    x = torch.randn(N, H, H, C, device='cuda'); res = torch.randn(N, H, H, C, device='cuda')
    w = packed(2 * C, C, 1); g = K.ConvGeom(w, 1, 0); b = torch.randn(2 * C, device='cuda'); piv = torch.randn(C, device='cuda')
    ab, out, parts = K.conv1x1_gate(x, w, g, b, res, 'elu', stats_pivot=piv)
    abr = (x.reshape(-1, C).double() @ w[:, :, 0, 0].t().double() + b.double())
    outr = (torch.nn.functional.elu(abr[:, :C]) * torch.sigmoid(abr[:, C:]) + res.reshape(-1, C).double())
    e = max(rel(ab.reshape(-1, 2 * C).double(), abr), rel(out.reshape(-1, C).double(), outr)); worst = max(worst, e)
    assert e < 3e-6, ('gate fwd', N, H, e)
    d = out.reshape(-1, C).double() - piv.double()
    pr = parts.rows_view().double()
    e = max(rel(pr[:, 0].sum(0), d.sum(0)), rel(pr[:, 1].sum(0), (d * d).sum(0)))
    assert e < 1e-5, ('gate stats', N, H, e)
print('gate forward: worst relative error %.2e over 30 shapes' % worst)

worst = 0.0
for it in range(20):
    H = random.choice([8, 16, 32]); N = random.randint(max(1, 16384 // (H * H)) + 1, max(3, 70000 // (H * H)))
    C = 64
    dout = torch.randn(N, H, H, C, device='cuda'); ab = torch.randn(N, H, H, 2 * C, device='cuda'); y = torch.randn(N, H, H, C, device='cuda')
    w = packed(2 * C, C, 1); g = K.ConvGeom(w, 1, 0)
    res = []
    for form in ('1', '0'):
        os.environ['LVAE_GATE_BWD_F32_MFMA'] = form
        dw, db = torch.zeros_like(w), torch.zeros(2 * C, device='cuda')
        dx = K.conv1x1_gate_bwd_wgrad(dout, ab, y, w, g, 'elu', dw, db)
        res.append((dx.clone(), dw.clone(), db.clone()))
    e = max(rel(res[1][0], res[0][0]), rel(res[1][1], res[0][1]), rel(res[1][2], res[0][2])); worst = max(worst, e)
    assert e < 5e-6, ('gate bwd', N, H, e)
print('fused gate backward forms: worst relative difference %.2e over 20 shapes' % worst)
