"""Fused residual-block kernels of the low-resolution levels (csrc/resblock_img.hip, lvae_resblock_conv_f32) against a plain torch
float64 statement of the reference's gated 'bacdbacd' block (lib/nn.py:78-99, 118-126) and its autograd: forward (two launches),
backward (two launches + the existing BatchNorm-1 apply), all intermediate tensors, statistics and parameter-gradient inputs."""
import math
import types

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def K():
    import lvae_amd  # noqa: F401
    from lvae_amd import kernels
    return kernels


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().float().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).cpu().double()


def packed_weight(w):
    return w.float().permute(2, 3, 1, 0).contiguous().cuda().permute(3, 2, 0, 1)


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


ACTS = {'elu': F.elu, 'relu': F.relu, 'leakyrelu': lambda t: F.leaky_relu(t, 0.01), 'selu': F.selu}


def make_block(N, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    C = 64
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    p = types.SimpleNamespace()
    p.x = r(N, C, H, W).requires_grad_(True)
    p.w1 = (r(C, C, 3, 3) / (3 * math.sqrt(C))).requires_grad_(True)
    p.w2 = (r(C, C, 3, 3) / (3 * math.sqrt(C))).requires_grad_(True)
    p.b1, p.b2 = (0.1 * r(C)).requires_grad_(True), (0.1 * r(C)).requires_grad_(True)
    p.wg = (r(2 * C, C, 1, 1) / math.sqrt(C)).requires_grad_(True)
    p.bg = (0.1 * r(2 * C)).requires_grad_(True)
    p.g1, p.g2 = (1 + 0.1 * r(C)).requires_grad_(True), (1 + 0.1 * r(C)).requires_grad_(True)
    p.be1, p.be2 = (0.1 * r(C)).requires_grad_(True), (0.1 * r(C)).requires_grad_(True)
    p.m1 = (torch.rand(N, C, generator=g) < 0.8).double() / 0.8
    p.m2 = (torch.rand(N, C, generator=g) < 0.8).double() / 0.8
    p.dout = r(N, C, H, W)
    return p


def reference(p, act='elu'):
    """float64 torch statement of the block and of everything the kernels hand to each other"""
    o = types.SimpleNamespace()
    fa = ACTS[act]
    bn = lambda t, ga, be: F.batch_norm(t, None, None, ga, be, True, 0.1, 1e-5)
    o.h1 = fa(bn(p.x, p.g1, p.be1))
    o.y1 = (F.conv2d(o.h1, p.w1, p.b1, padding=1) * p.m1[:, :, None, None])
    o.h2 = fa(bn(o.y1, p.g2, p.be2))
    o.y2 = (F.conv2d(o.h2, p.w2, p.b2, padding=1) * p.m2[:, :, None, None])
    o.ab = F.conv2d(o.y2, p.wg, p.bg)
    a, b = o.ab.chunk(2, 1)
    o.out = fa(a) * torch.sigmoid(b) + p.x
    for t in (o.h1, o.y1, o.h2, o.y2, o.ab):
        t.retain_grad()
    o.out.backward(p.dout)
    return o


@pytest.mark.parametrize('shape,prec,act', [(s, pr, 'elu') for pr in ('f32', 'bf16') for s in
                                            [(256, 4, 4), (64, 8, 8), (37, 2, 2), (7, 8, 8), (130, 4, 4), (1, 2, 2), (300, 8, 8)]] +
                         [((40, 4, 4), 'f32', 'selu'), ((9, 8, 8), 'f32', 'leakyrelu'), ((33, 2, 2), 'f32', 'relu')] +   # the run-time-activation build
                         # tile sizes (round 5): 32-pixel tiles where 64-pixel tiles would fill at most half the CUs (every 4x4 / 2x2 shape above),
                         # 64-pixel tiles beyond that (600 x 4x4 = 150 tiles), non-square images of 8 pixels (4 per 32-pixel tile)
                         [((600, 4, 4), 'f32', 'elu'), ((5, 2, 4), 'f32', 'elu'), ((5, 2, 4), 'bf16', 'elu'), ((2100, 2, 2), 'f32', 'elu')])
def test_fused_block_forward_and_backward(K, shape, prec, act):
    N, H, W = shape
    C = 64
    p = make_block(N, H, W, 7 * N + H)
    o = reference(p, act)
    tol = 1.0 if prec == 'f32' else 4000.0   # bf16 operands: 2^-9 relative per product instead of 2^-24
    K.set_precision(prec)
    try:
        dev = 'cuda'
        f = lambda t: t.detach().float().to(dev)
        x = nhwc(p.x.detach())
        w1, w2, wg = packed_weight(p.w1.detach()), packed_weight(p.w2.detach()), packed_weight(p.wg.detach())
        ge1, ge2, geg = K.ConvGeom(w1, 1, 1), K.ConvGeom(w2, 1, 1), K.ConvGeom(wg, 1, 0)
        assert K.rb_rows(x, w1, ge1) > 0
        mk_bn = lambda ga, be: types.SimpleNamespace(weight=f(ga), bias=f(be), running_mean=torch.zeros(C, device=dev),
                                                     running_var=torch.ones(C, device=dev), eps=1e-5, momentum=0.1)
        bn1, bn2 = mk_bn(p.g1, p.be1), mk_bn(p.g2, p.be2)
        m1, m2 = p.m1.float().to(dev), p.m2.float().to(dev)
        # ---- forward: BN1 statistics by the stand-alone kernel (first block of a chain), then the two fused launches
        coef1 = K.bn_stats(x, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, bn1.eps, bn1.momentum)
        y1, parts2, _ = K.rb_conv(x, w1, ge1, f(p.b1), act, m1, coef=coef1, stats_pivot=bn2.running_mean)
        pivot = coef1[2]
        y2, ab, out, oparts, coef2 = K.rb_conv_gate(y1, w2, ge2, f(p.b2), act, m2, wg, geg, f(p.bg), x, act,
                                                    in_bn=(parts2, bn2.running_mean, bn2), stats_pivot=pivot)
        torch.cuda.synchronize()
        assert rel(nchw(y1), o.y1.detach()) < 2e-6 * tol
        assert rel(nchw(y2), o.y2.detach()) < 3e-6 * tol
        assert rel(nchw(ab), o.ab.detach()) < 3e-6 * tol
        assert rel(nchw(out), o.out.detach()) < 3e-6 * tol
        # folded BatchNorm-2 finalize: coefficients and running statistics as nn.BatchNorm2d would leave them
        y1r = o.y1.detach()
        mean2, var2 = y1r.mean((0, 2, 3)), y1r.var((0, 2, 3), unbiased=False)
        stol = dict(rtol=1e-5 * tol, atol=1e-5 * tol)
        torch.testing.assert_close(coef2[2].cpu().double(), mean2, **stol)
        torch.testing.assert_close(coef2[3].cpu().double(), 1 / torch.sqrt(var2 + 1e-5), **stol)
        M = N * H * W
        torch.testing.assert_close(bn2.running_mean.cpu().double(), 0.1 * mean2, **stol)
        torch.testing.assert_close(bn2.running_var.cpu().double(), 0.9 + 0.1 * var2 * M / max(M - 1, 1), **stol)
        # statistics of `out` for the next block: partial rows around the pivot, and the pivot behind them
        s = oparts.rows_view().double().sum(0).cpu()
        dl = o.out.detach() - pivot.cpu().double().view(1, -1, 1, 1)
        torch.testing.assert_close(s[0], dl.sum((0, 2, 3)), rtol=1e-4 * tol, atol=1e-3 * tol)
        torch.testing.assert_close(s[1], (dl * dl).sum((0, 2, 3)), rtol=1e-4 * tol, atol=1e-3 * tol)
        assert torch.equal(oparts.buf[oparts.rows, 0], pivot)
        # ---- backward
        dout = nhwc(p.dout)
        assert K.bn_coef_block(*coef2) and K.bn_coef_block(*coef1)
        dab, dy2, dh2, bparts2 = K.rb_gate_dgrad(dout, ab, wg, geg, act, m2, w2, ge2, bn_bwd=(y1, coef2[0], act))
        dg2, db2 = torch.full((C,), 0.5, device=dev), torch.full((C,), -0.25, device=dev)   # accumulated into
        dy1, dh1, bparts1 = K.rb_apply_dgrad(bparts2, dh2, y1, coef2[0], act, dg2, db2, m1, w1, ge1, bn_bwd=(x, coef1[0], act))
        dg1, db1 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dx = K.affine_act_bwd_parts(bparts1, dh1, x, coef1[0], coef1[1], act, coef1[2], coef1[3], dg1, db1, add=dout)
        torch.cuda.synchronize()
        btol = tol * 3
        assert rel(nchw(dab), o.ab.grad) < 2e-6 * btol
        # gradient w.r.t. conv2's output (before the Dropout2d mask) = y2.grad * m2
        assert rel(nchw(dy2), o.y2.grad * p.m2[:, :, None, None]) < 3e-6 * btol
        assert rel(nchw(dh2), o.h2.grad) < 4e-6 * btol
        assert rel(nchw(dy1), o.y1.grad * p.m1[:, :, None, None]) < 6e-6 * btol
        assert rel(nchw(dh1), o.h1.grad) < 8e-6 * btol
        assert rel(nchw(dx), p.x.grad) < 1e-5 * btol
        gt = dict(rtol=2e-4 * btol, atol=2e-4 * btol)
        torch.testing.assert_close(dg2.cpu().double() - 0.5, p.g2.grad, **gt)
        torch.testing.assert_close(db2.cpu().double() + 0.25, p.be2.grad, **gt)
        torch.testing.assert_close(dg1.cpu().double(), p.g1.grad, **gt)
        torch.testing.assert_close(db1.cpu().double(), p.be1.grad, **gt)
    finally:
        K.set_precision('f32')


@pytest.mark.parametrize('shape,prec', [((256, 4, 4), 'f32'), ((37, 2, 2), 'f32'), ((64, 8, 8), 'f32'), ((1, 2, 2), 'f32'), ((130, 4, 4), 'bf16'),
                                        ((256, 16, 16), 'f32'), ((70, 16, 16), 'f32'), ((20, 32, 32), 'f32'), ((257, 16, 16), 'bf16')])
def test_deferred_batchnorm_apply_in_the_next_blocks_first_backward_launch(K, shape, prec):
    """The BatchNorm-1 apply that ends a block's backward, dx = BN1'(dh; x) + add, formed instead by the first backward launch of the block
    that consumes dx (kernels.PendingApply): the whole-image gate-backward + dgrad launch (<= 8x8 levels, lvae_rb_ext.ap_*) and the persistent
    fused gate-backward kernel (>= 16 k pixels, lvae_bn_apply). Against the two-launch form on the same operands: the deferred launch must
    write the same dout (to 1e-6: the partial rows are summed in another order), the same dgamma / dbeta, and everything downstream of dout."""
    N, H, W = shape
    C = 64
    dev = 'cuda'
    g = torch.Generator().manual_seed(11 * N + H)
    rn = lambda *s_: torch.randn(*s_, generator=g).to(dev)
    K.set_precision(prec)
    try:
        # the producer block's side: dh (gradient w.r.t. act(BN(x))), its partial BatchNorm-backward sums, x, the coefficient block, add
        x, dh, add = rn(N, H, W, C), rn(N, H, W, C), rn(N, H, W, C)
        coef = K.bn_stats(x, torch.rand(C, generator=g).to(dev) + 0.5, rn(C) * 0.1, None, None)
        xh = (x - coef[2]) * coef[3]
        u = x * coef[0] + coef[1]
        gg = dh * torch.where(u > 0, torch.ones_like(u), torch.exp(u))
        rows = 7 if N * H * W < 4096 else 256
        chunks_g = gg.reshape(-1, C).tensor_split(rows)
        chunks_x = (gg * xh).reshape(-1, C).tensor_split(rows)
        parts = torch.stack([torch.stack([a.sum(0), b.sum(0)]) for a, b in zip(chunks_g, chunks_x)]).contiguous()   # [rows][2][C]
        # the consumer block's side
        wg = packed_weight(torch.randn(2 * C, C, 1, 1, generator=g) / 8)
        geg = K.ConvGeom(wg, 1, 0)
        ab, y2 = rn(N, H, W, 2 * C), rn(N, H, W, C)
        m2 = ((torch.rand(N, C, generator=g) < 0.8).float() / 0.8).to(dev)
        dg_a, db_a = torch.full((C,), 0.5, device=dev), torch.full((C,), -1.0, device=dev)
        dg_b, db_b = dg_a.clone(), db_a.clone()
        dout = K.affine_act_bwd_parts(parts, dh, x, coef[0], coef[1], 'elu', coef[2], coef[3], dg_a, db_a, add=add)
        out = torch.full_like(x, float('nan'))   # what the producer would return unwritten
        pend = K.PendingApply(parts, dh, x, coef[0], 'elu', dg_b, db_b, add, out)
        if H * W <= 64:
            w2 = packed_weight(torch.randn(C, C, 3, 3, generator=g) / 24)
            ge2 = K.ConvGeom(w2, 1, 1)
            y1 = rn(N, H, W, C)
            coef2 = K.bn_stats(y1, None, None, None, None)
            # with L2 warm-up ranges, as inside a step (the first whole-step run of this variant died of late-landing warm-up loads: profiles/r05_faults/)
            pf = K.rb_weight_ranges(dout, w2, ge2, True, gate=(wg, geg), gate_bwd=True)
            ref = K.rb_gate_dgrad(dout, ab, wg, geg, 'elu', m2, w2, ge2, bn_bwd=(y1, coef2[0], 'elu'), prefetch=pf)
            got = K.rb_gate_dgrad(out, ab, wg, geg, 'elu', m2, w2, ge2, bn_bwd=(y1, coef2[0], 'elu'), apply=pend, prefetch=pf)
        else:
            assert K.gate_bwd_fused_ok(x, wg, geg)
            dw_a, dbias_a = torch.zeros_like(wg), torch.zeros(2 * C, device=dev)
            dw_b, dbias_b = torch.zeros_like(wg), torch.zeros(2 * C, device=dev)
            ref = (K.conv1x1_gate_bwd_wgrad(dout, ab, y2, wg, geg, 'elu', dw_a, dbias_a, out_scale=m2), dw_a, dbias_a)
            got = (K.conv1x1_gate_bwd_wgrad(out, ab, y2, wg, geg, 'elu', dw_b, dbias_b, out_scale=m2, apply=pend), dw_b, dbias_b)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out).all())
        assert rel(out.cpu(), dout.cpu()) < 1e-6
        torch.testing.assert_close(dg_b, dg_a, rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(db_b, db_a, rtol=1e-5, atol=1e-4)
        for a, b in zip(got, ref):
            assert rel(a.float().cpu(), b.float().cpu()) < (2e-6 if prec == 'f32' else 2e-3)
    finally:
        K.set_precision('f32')


def test_fused_block_shape_gate(K):
    """lvae_resblock_conv_rows: only 64 -> 64 channel 3x3 / stride 1 / pad 1 layers whose images divide a 64-pixel tile."""
    mk = lambda co, ci, k: packed_weight(torch.randn(co, ci, k, k))
    w = mk(64, 64, 3)
    ge = K.ConvGeom(w, 1, 1)
    for hw, want in (((8, 8), True), ((4, 4), True), ((2, 2), True), ((16, 16), False), ((6, 6), False), ((4, 8), True)):
        x = torch.zeros(5, hw[0], hw[1], 64, device='cuda')
        assert (K.rb_rows(x, w, ge) > 0) == want, hw
    w32 = mk(64, 32, 3)
    assert K.rb_rows(torch.zeros(5, 4, 4, 32, device='cuda'), w32, K.ConvGeom(w32, 1, 1)) == 0
    assert K.rb_rows(torch.zeros(5, 4, 4, 64, device='cuda'), w, K.ConvGeom(w, 2, 1)) == 0


@pytest.mark.parametrize('shape', [(256, 16, 16), (300, 16, 16), (64, 32, 32), (70, 32, 32)])
def test_conv_gate_fusion_behind_the_winograd_kernel(K, shape):
    """lvae_resblock_conv_f32 / LVAE_RB_EPI_GATE at the >= 16x16 levels (fp32): the GateLayer2d and the residual add run behind the 256-pixel
    six-product Winograd kernel's epilogue. y2, ab, out, the folded BatchNorm finalize of the input and the BatchNorm partials of `out`
    against the float64 statement of lib/nn.py:80-99,118-126."""
    N, H, W = shape
    C = 64
    p = make_block(N, H, W, N + 3 * H)
    o = reference(p)
    dev = 'cuda'
    f = lambda t: t.detach().float().to(dev)
    x = nhwc(p.x.detach())
    w1, w2, wg = packed_weight(p.w1.detach()), packed_weight(p.w2.detach()), packed_weight(p.wg.detach())
    ge1, ge2, geg = K.ConvGeom(w1, 1, 1), K.ConvGeom(w2, 1, 1), K.ConvGeom(wg, 1, 0)
    assert K.rb_rows(x, w2, ge2) == 0 and K.rb_gate_rows(x, w2, ge2) > 0
    mk_bn = lambda ga, be: types.SimpleNamespace(weight=f(ga), bias=f(be), running_mean=torch.zeros(C, device=dev),
                                                 running_var=torch.ones(C, device=dev), eps=1e-5, momentum=0.1)
    bn1, bn2 = mk_bn(p.g1, p.be1), mk_bn(p.g2, p.be2)
    m1, m2 = p.m1.float().to(dev), p.m2.float().to(dev)
    coef1 = K.bn_stats(x, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, bn1.eps, bn1.momentum)
    y1, parts2 = K.conv2d(x, w1, ge1, bias=f(p.b1), in_scale=coef1[0], in_shift=coef1[1], in_act='elu', out_scale=m1, stats_pivot=bn2.running_mean)
    assert parts2 is not None
    pivot = coef1[2]
    y2, ab, out, oparts, coef2 = K.rb_conv_gate(y1, w2, ge2, f(p.b2), 'elu', m2, wg, geg, f(p.bg), x, 'elu',
                                                in_bn=(parts2, bn2.running_mean, bn2), stats_pivot=pivot)
    torch.cuda.synchronize()
    assert rel(nchw(y1), o.y1.detach()) < 3e-6
    assert rel(nchw(y2), o.y2.detach()) < 4e-6
    assert rel(nchw(ab), o.ab.detach()) < 4e-6
    assert rel(nchw(out), o.out.detach()) < 4e-6
    y1r = o.y1.detach()
    mean2, var2 = y1r.mean((0, 2, 3)), y1r.var((0, 2, 3), unbiased=False)
    torch.testing.assert_close(coef2[2].cpu().double(), mean2, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(coef2[3].cpu().double(), 1 / torch.sqrt(var2 + 1e-5), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(bn2.running_mean.cpu().double(), 0.1 * mean2, rtol=1e-5, atol=1e-5)
    s = oparts.rows_view().double().sum(0).cpu()
    dl = o.out.detach() - pivot.cpu().double().view(1, -1, 1, 1)
    torch.testing.assert_close(s[0], dl.sum((0, 2, 3)), rtol=1e-4, atol=2e-2)
    torch.testing.assert_close(s[1], (dl * dl).sum((0, 2, 3)), rtol=1e-4, atol=2e-2)
    assert torch.equal(oparts.buf[oparts.rows, 0], pivot)
    # the same launch with given coefficients (no partial sums: e.g. the first block behind a resampling convolution)
    y2b, abb, outb, _, _ = K.rb_conv_gate(y1, w2, ge2, f(p.b2), 'elu', m2, wg, geg, f(p.bg), x, 'elu', coef=coef2, stats_pivot=None)
    torch.cuda.synchronize()
    assert torch.equal(y2b, y2) and torch.equal(abb, ab) and torch.equal(outb, out)
