"""Every consumer of per-workgroup partial statistics rows behind a producer that wrote FEWER rows than the consumer has reduction groups.

Round 4 found (as an aborted test run, profiles/r05_faults/README.md has the pointer) that five reducers clamped an out-of-range row index to
the row-GROUP index instead of row 0 — an out-of-bounds read whenever a producer wrote fewer rows than the reducer has groups. The fix is in;
this file pins the whole class: each site is driven with 1 and with 3 partial rows (and, as a control, with many), against the same operation
fed explicit coefficients computed in float64 from the full data. Sites:
  * conv3x3_pos.hip   folded BatchNorm finalize (lvae_bn_fold) of the position-major 3x3 kernel (<= 4x4 levels)
  * conv3x3_wino.hip  wino_fold_bn of the Winograd kernels (8x8 / 16x16 levels)
  * norm_act.hip      lvae_affine_act_bwd_parts_f32 (BatchNorm-backward apply from partial sums)
  * resblock_img.hip  rb_parts_issue / rb_parts_finish: forward fold (PRO_AFFINE), BatchNorm-backward apply (PRO_BN_APPLY) and the deferred
                      apply in front of the gate backward (round 5)
  * conv1x1_gate_bwd_fused.hip  the deferred apply of the persistent gate-backward kernel (round 5)
"""
import math
import types

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
C = 64


@pytest.fixture(scope='module')
def K():
    import lvae_amd  # noqa: F401
    from lvae_amd import kernels
    return kernels


def packed_weight(w):
    return w.float().permute(2, 3, 1, 0).contiguous().cuda().permute(3, 2, 0, 1)


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def fwd_parts(K, x, rows, pivot):
    """StatParts of x (N,H,W,C) as a producer with `rows` workgroups would have written them: (sum(x - pivot), sum((x - pivot)^2)) per chunk
    of pixels, then the pivot row."""
    d = (x.reshape(-1, C).double() - pivot.double())
    buf = torch.zeros(rows + 1, 2, C, device=x.device)
    for r, ch in enumerate(d.tensor_split(rows)):
        buf[r, 0], buf[r, 1] = ch.sum(0).float(), (ch * ch).sum(0).float()
    buf[rows, 0] = pivot
    return K.StatParts(buf, rows, True)


def make_bn(g):
    dev = 'cuda'
    return types.SimpleNamespace(weight=(1 + 0.1 * torch.randn(C, generator=g)).to(dev), bias=(0.1 * torch.randn(C, generator=g)).to(dev),
                                 running_mean=torch.zeros(C, device=dev), running_var=torch.ones(C, device=dev), eps=1e-5, momentum=0.1)


def ref_coef(x, bn):
    xd = x.reshape(-1, C).double()
    mean, var = xd.mean(0), xd.var(0, unbiased=False)
    rstd = 1 / torch.sqrt(var + bn.eps)
    sc = bn.weight.double() * rstd
    return sc.float(), (bn.bias.double() - mean * sc).float(), mean.float(), rstd.float()


@pytest.mark.parametrize('rows', [1, 3, 40])
@pytest.mark.parametrize('shape', [(64, 4, 4), (33, 2, 2),        # position-major kernel's fold
                                   (256, 8, 8), (64, 16, 16)])    # Winograd kernels' fold
def test_folded_finalize_of_a_convolution_input(K, shape, rows):
    N, H, W = shape
    g = torch.Generator().manual_seed(N + H + rows)
    x = (torch.randn(N, H, W, C, generator=g) * 1.5 + 0.3).cuda()
    w = packed_weight(torch.randn(C, C, 3, 3, generator=g) / 24)
    geom = K.ConvGeom(w, 1, 1)
    bn = make_bn(g)
    pivot = x[0, 0, 0].clone()
    sc, sh, mean, rstd = ref_coef(x, bn)
    want = K.conv2d(x, w, geom, in_scale=sc, in_shift=sh, in_act='elu')
    K.prepared.prepare_all()
    y, _, coef = K.conv2d(x, w, geom, in_act='elu', in_bn=(fwd_parts(K, x, rows, pivot), pivot, bn))
    torch.cuda.synchronize()
    assert rel(y, want) < 2e-5
    assert rel(coef[2], mean) < 1e-5 and rel(coef[3], rstd) < 1e-5
    torch.testing.assert_close(bn.running_mean, 0.1 * mean, rtol=1e-4, atol=1e-5)


def bwd_parts(dh, x, coef, rows):
    """(sum g, sum g xhat) per chunk, g = dh * elu'(x * scale + shift), as a dgrad epilogue (stats_mode LVAE_STATS_BN_BWD) writes them"""
    u = x.double() * coef[0].double() + coef[1].double()
    g = dh.double() * torch.where(u > 0, torch.ones_like(u), torch.exp(u))
    xh = (x.double() - coef[2].double()) * coef[3].double()
    parts = torch.stack([torch.stack([a.sum(0), b.sum(0)]) for a, b in zip(g.reshape(-1, C).tensor_split(rows), (g * xh).reshape(-1, C).tensor_split(rows))])
    dx = (g - g.reshape(-1, C).mean(0) - xh * (g * xh).reshape(-1, C).mean(0)) * coef[0].double()
    return parts.float().contiguous(), dx, g.reshape(-1, C).sum(0), (g * xh).reshape(-1, C).sum(0)


@pytest.mark.parametrize('rows', [1, 3, 130])
@pytest.mark.parametrize('shape', [(64, 4, 4), (7, 2, 2), (64, 16, 16), (256, 8, 8)])
def test_batchnorm_backward_apply_from_few_partial_rows(K, shape, rows):
    """lvae_affine_act_bwd_parts_f32, and the same apply inside the fused launches (PRO_BN_APPLY; deferred in front of the gate backward)."""
    N, H, W = shape
    g = torch.Generator().manual_seed(3 * N + H + rows)
    rn = lambda *s: torch.randn(*s, generator=g).cuda()
    x, dh, add = rn(N, H, W, C), rn(N, H, W, C), rn(N, H, W, C)
    bn = make_bn(g)
    coef = K.bn_stats(x, bn.weight, bn.bias, None, None)
    parts, dx_ref, sg, sgx = bwd_parts(dh, x, coef, rows)
    dgam, dbet = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    dx = K.affine_act_bwd_parts(parts, dh, x, coef[0], coef[1], 'elu', coef[2], coef[3], dgam, dbet, add=add)
    torch.cuda.synchronize()
    assert rel(dx, dx_ref + add.double()) < 2e-6
    torch.testing.assert_close(dbet.double(), sg, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(dgam.double(), sgx, rtol=1e-4, atol=1e-3)
    if H * W > 64:
        return
    # the fused launches of the <= 8x8 levels: BatchNorm-apply + dgrad, and the deferred apply + gate backward + dgrad
    w = packed_weight(torch.randn(C, C, 3, 3, generator=g) / 24)
    geom = K.ConvGeom(w, 1, 1)
    xb = rn(N, H, W, C)
    coefb = K.bn_stats(xb, None, None, None, None)
    drop = ((torch.rand(N, C, generator=g) < 0.8).float() / 0.8).cuda()
    dg2, db2 = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    dy1, dh1, _ = K.rb_apply_dgrad(parts, dh, x, coef[0], 'elu', dg2, db2, drop, w, geom, bn_bwd=(xb, coefb[0], 'elu'))
    torch.cuda.synchronize()
    want_dy1 = (dx_ref.float() * drop.view(N, 1, 1, C))
    assert rel(dy1, want_dy1) < 3e-6
    assert rel(dh1, K.conv2d_dgrad(want_dy1.contiguous(), w, geom, (H, W))) < 5e-6
    torch.testing.assert_close(db2.double(), sg, rtol=1e-4, atol=1e-3)
    wg = packed_weight(torch.randn(2 * C, C, 1, 1, generator=g) / 8)
    geg = K.ConvGeom(wg, 1, 0)
    ab = rn(N, H, W, 2 * C)
    out = torch.full_like(x, float('nan'))
    dg3, db3 = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    pend = K.PendingApply(parts, dh, x, coef[0], 'elu', dg3, db3, add, out)
    got = K.rb_gate_dgrad(out, ab, wg, geg, 'elu', drop, w, geom, bn_bwd=(xb, coefb[0], 'elu'), apply=pend)
    ref = K.rb_gate_dgrad(dx, ab, wg, geg, 'elu', drop, w, geom, bn_bwd=(xb, coefb[0], 'elu'))
    torch.cuda.synchronize()
    assert rel(out, dx) < 1e-6
    for a, b in zip(got, ref):
        assert rel(a, b) < 3e-6
    torch.testing.assert_close(dg3, dgam, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('rows', [1, 3])
def test_forward_fold_of_the_fused_whole_image_launch(K, rows):
    """rb_conv (PRO_AFFINE) with its BatchNorm finalized from 1 / 3 partial rows in the prologue."""
    N, H, W = 37, 4, 4
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(N, H, W, C, generator=g) + 0.2).cuda()
    w = packed_weight(torch.randn(C, C, 3, 3, generator=g) / 24)
    geom = K.ConvGeom(w, 1, 1)
    b = torch.randn(C, generator=g).cuda()
    bn = make_bn(g)
    pivot = x[0, 0, 0].clone()
    sc, sh, mean, rstd = ref_coef(x, bn)
    block = torch.stack([sc, sh, mean, rstd]).contiguous()
    want, _, _ = K.rb_conv(x, w, geom, b, 'elu', None, coef=(block[0], block[1], block[2], block[3]))
    got, _, coef = K.rb_conv(x, w, geom, b, 'elu', None, in_bn=(fwd_parts(K, x, rows, pivot), pivot, bn))
    torch.cuda.synchronize()
    assert rel(got, want) < 2e-5
    assert rel(coef[2], mean) < 1e-5 and rel(coef[3], rstd) < 1e-5


@pytest.mark.parametrize('rows', [1, 3])
def test_deferred_apply_of_the_persistent_gate_backward_from_few_rows(K, rows):
    N, H, W = 64, 16, 16
    g = torch.Generator().manual_seed(50 + rows)
    rn = lambda *s: torch.randn(*s, generator=g).cuda()
    x, dh, add, ab, y2 = rn(N, H, W, C), rn(N, H, W, C), rn(N, H, W, C), rn(N, H, W, 2 * C), rn(N, H, W, C)
    coef = K.bn_stats(x, None, None, None, None)
    parts, dx_ref, sg, sgx = bwd_parts(dh, x, coef, rows)
    wg = packed_weight(torch.randn(2 * C, C, 1, 1, generator=g) / 8)
    geg = K.ConvGeom(wg, 1, 0)
    dgam, dbet = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    out = torch.full_like(x, float('nan'))
    pend = K.PendingApply(parts, dh, x, coef[0], 'elu', dgam, dbet, add, out)
    dw, db = torch.zeros_like(wg), torch.zeros(2 * C, device='cuda')
    dx = K.conv1x1_gate_bwd_wgrad(out, ab, y2, wg, geg, 'elu', dw, db, apply=pend)
    assert dx is not None
    torch.cuda.synchronize()
    assert rel(out, dx_ref + add.double()) < 2e-6
    torch.testing.assert_close(dbet.double(), sg, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(dgam.double(), sgx, rtol=1e-4, atol=1e-3)
