"""bf16 storage of the tensors inside a residual block under compute_dtype = 'bf16' (BASELINE configs[1], [3], [4]; include/lvae_hip.h
LVAE_DT_BF16): conv outputs y1 / y2, the gate pre-activations ab and their gradients live in HBM as bfloat16 — what torch.autocast(bfloat16)
stores for nn.Conv2d outputs — while the residual stream, statistics and parameters stay fp32.

Kernel level (exact): a kernel fed bf16-stored tensors must produce exactly what it produces from the same values stored as fp32, and a
bf16-stored output must be the round-to-nearest-even of the fp32-stored one. Block level: forward + backward of a whole ResidualBlock
with bf16-stored internals against the same block with fp32-stored internals (bf16 rounding tolerance) and the model-level parity tests of
tests/test_fullsize_gpu.py (ELBO within north_star's 1e-2 of the fp32 oracle)."""
import ctypes
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def K():
    import lvae_amd  # noqa: F401
    from lvae_amd import kernels
    kernels._C.load()
    kernels.set_precision('bf16')
    yield kernels
    kernels.set_precision('f32')


def packed(co, ci, k):
    return (torch.randn(k, k, ci, co, device='cuda') / (ci * k * k) ** 0.5).permute(3, 2, 0, 1)


def bfr(t):
    """values representable in bf16, stored as fp32"""
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize('shape', [(256, 16, 16), (70, 32, 32), (512, 8, 8)])   # (8x8 at batch 256 keeps fp32 storage: its weight gradient is the grouped fp32 one)
def test_conv3x3_bf16_storage_is_exact(K, shape):
    N, H, W = shape
    C = 64
    torch.manual_seed(N + H)
    x = bfr(torch.randn(N, H, W, C, device='cuda'))
    w = packed(C, C, 3)
    g = K.ConvGeom(w, 1, 1)
    b = torch.randn(C, device='cuda')
    sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.3
    drop = (torch.rand(N, C, device='cuda') < 0.8).float() / 0.8
    piv = torch.randn(C, device='cuda') * 0.1
    assert K.resblock_bf16_storage(x, w, g)
    kw = dict(bias=b, in_scale=sc, in_shift=sh, in_act='elu', out_scale=drop, stats_pivot=piv)
    y32, p32 = K.conv2d(x, w, g, **kw)
    y16, p16 = K.conv2d(x.to(torch.bfloat16), w, g, out_bf16=True, **kw)
    assert y16.dtype == torch.bfloat16 and torch.equal(y16, y32.to(torch.bfloat16))
    # statistics come from the fp32 accumulators either way (the 8-channel epilogue sums them in another order)
    torch.testing.assert_close(p16.rows_view().sum(0), p32.rows_view().sum(0), rtol=2e-5, atol=1e-2)
    ymix = K.conv2d(x.to(torch.bfloat16), w, g, **kw)[0]      # bf16 in, fp32 out
    assert torch.equal(ymix, y32)
    # dgrad with the BatchNorm-backward sums in its epilogue: dy bf16, stats_x bf16 or fp32, dh bf16
    dy = bfr(torch.randn(N, H, W, C, device='cuda'))
    xb = bfr(torch.randn(N, H, W, C, device='cuda'))
    coef = torch.stack([sc, sh, torch.randn(C, device='cuda') * 0.1, torch.rand(C, device='cuda') + 0.5]).contiguous()
    d32, q32 = K.conv2d_dgrad(dy, w, g, (H, W), bn_bwd=(xb, coef[0], 'elu'))
    for xs in (xb, xb.to(torch.bfloat16)):
        d16, q16 = K.conv2d_dgrad(dy.to(torch.bfloat16), w, g, (H, W), bn_bwd=(xs, coef[0], 'elu'), out_bf16=True)
        assert torch.equal(d16, d32.to(torch.bfloat16))
        torch.testing.assert_close(q16.sum(0), q32.sum(0), rtol=2e-5, atol=1e-2)
    # weight gradient: x fp32 or bf16, dy bf16 (and x bf16, dy fp32, which the step does not produce): the same MFMA operands, so the
    # storage forms agree exactly with each other; layers of >= 512 tiles take the half-slab kernel for them and keep the whole-slab one
    # when both operands are fp32-stored, whose fp32 sums run in another order (1e-6)
    dw32, db32 = torch.zeros_like(w), torch.zeros(C, device='cuda')
    K.conv2d_wgrad(x, dy, w, g, dw32, db32, in_scale=sc, in_shift=sh, in_act='elu')
    first = None
    for xs, ds in ((x, dy.to(torch.bfloat16)), (x.to(torch.bfloat16), dy.to(torch.bfloat16)), (x.to(torch.bfloat16), dy)):
        dw16, db16 = torch.zeros_like(w), torch.zeros(C, device='cuda')
        K.conv2d_wgrad(xs, ds, w, g, dw16, db16, in_scale=sc, in_shift=sh, in_act='elu')
        first = dw16 if first is None else first
        assert torch.equal(dw16, first)
        assert float((dw16 - dw32).norm() / dw32.norm()) < 1e-6
        torch.testing.assert_close(db16, db32, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize('shape', [(256, 16, 16), (129, 16, 16), (300, 8, 8)])
def test_gate_kernels_bf16_storage_is_exact(K, shape):
    N, H, W = shape
    C = 64
    torch.manual_seed(N)
    x = bfr(torch.randn(N, H, W, C, device='cuda'))
    res = torch.randn(N, H, W, C, device='cuda')
    w = packed(2 * C, C, 1)
    g = K.ConvGeom(w, 1, 0)
    b = torch.randn(2 * C, device='cuda')
    piv = torch.randn(C, device='cuda') * 0.1
    ab32, out32, p32 = K.conv1x1_gate(x, w, g, b, res, 'elu', stats_pivot=piv)
    ab16, out16, p16 = K.conv1x1_gate(x.to(torch.bfloat16), w, g, b, res, 'elu', stats_pivot=piv)
    assert ab16.dtype == torch.bfloat16 and out16.dtype == torch.float32
    assert torch.equal(ab16, ab32.to(torch.bfloat16)) and torch.equal(out16, out32)
    assert torch.equal(p16.rows_view(), p32.rows_view()) and torch.equal(p16.buf[p16.rows, 0], p32.buf[p32.rows, 0])   # partial rows + the pivot row
    # fused backward: ab, y stored bf16; dx stored bf16
    dout = torch.randn(N, H, W, C, device='cuda')
    ab = bfr(torch.randn(N, H, W, 2 * C, device='cuda'))
    y = bfr(torch.randn(N, H, W, C, device='cuda'))
    mask = (torch.rand(N, C, device='cuda') < 0.8).float() / 0.8
    dw32, db32 = torch.zeros_like(w), torch.zeros(2 * C, device='cuda')
    dx32 = K.conv1x1_gate_bwd_wgrad(dout, ab, y, w, g, 'elu', dw32, db32, out_scale=mask)
    dw16, db16 = torch.zeros_like(w), torch.zeros(2 * C, device='cuda')
    dx16 = K.conv1x1_gate_bwd_wgrad(dout, ab.to(torch.bfloat16), y.to(torch.bfloat16), w, g, 'elu', dw16, db16, out_scale=mask, out_bf16=True)
    assert dx32 is not None and dx16 is not None and dx16.dtype == torch.bfloat16
    assert torch.equal(dx16, dx32.to(torch.bfloat16)) and torch.equal(dw16, dw32)
    torch.testing.assert_close(db16, db32, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize('shape', [(256, 16, 16, 512), (256, 8, 8, 128), (70, 32, 32, 560)])
def test_bn_backward_apply_bf16_storage_is_exact(K, shape):
    N, H, W, rows = shape
    C = 64
    torch.manual_seed(rows)
    dh = bfr(torch.randn(N, H, W, C, device='cuda'))
    x = bfr(torch.randn(N, H, W, C, device='cuda'))
    add = torch.randn(N, H, W, C, device='cuda')
    parts = torch.randn(rows, 2, C, device='cuda')
    coef = torch.stack([torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.3, torch.randn(C, device='cuda') * 0.1,
                        torch.rand(C, device='cuda') + 0.5]).contiguous()
    drop = (torch.rand(N, C, device='cuda') < 0.8).float() / 0.8

    def run(dh_, x_, out_bf16, **kw):
        dg, db = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
        return K.affine_act_bwd_parts(parts, dh_, x_, coef[0], coef[1], 'elu', coef[2], coef[3], dg, db, out_bf16=out_bf16, **kw), dg, db

    ref, dg0, db0 = run(dh, x, False, drop=drop)
    for xs in (x, x.to(torch.bfloat16)):
        got, dg, db = run(dh.to(torch.bfloat16), xs, True, drop=drop)
        assert got.dtype == torch.bfloat16 and torch.equal(got, ref.to(torch.bfloat16)) and torch.equal(dg, dg0) and torch.equal(db, db0)
    ref, _, _ = run(dh, x, False, add=add)                 # BatchNorm 1 of a block: fp32 x, + the residual gradient, fp32 result
    got, _, _ = run(dh.to(torch.bfloat16), x, False, add=add)
    assert got.dtype == torch.float32 and torch.equal(got, ref)


def test_kernels_without_a_bf16_storage_form_refuse_bf16_tensors(K):
    x = torch.randn(256, 4, 4, 64, device='cuda').to(torch.bfloat16)      # a 4x4 level: position-major fp32 kernel
    w = packed(64, 64, 3)
    g = K.ConvGeom(w, 1, 1)
    assert not K.resblock_bf16_storage(x.float(), w, g)
    with pytest.raises(K._C.LvaeHipError):
        K.conv2d(x, w, g)
    K.set_precision('f32')                                               # fp32 arithmetic never stores bf16
    xl = torch.randn(256, 16, 16, 64, device='cuda')
    assert not K.resblock_bf16_storage(xl, w, g)
    with pytest.raises(K._C.LvaeHipError):
        K.conv2d(xl, w, g, out_bf16=True)


@pytest.mark.parametrize('shape', [(128, 16, 16), (512, 8, 8)])
def test_residual_block_with_bf16_stored_internals(K, shape, monkeypatch):
    """Whole gated 'bacdbacd' block, training mode: bf16-stored internals against fp32-stored internals (same bf16-operand kernels, same
    dropout masks): outputs and every gradient agree to the bf16 rounding of the stored tensors."""
    from lvae_amd.lib.nn import ResidualGatedBlock
    from lvae_amd.noise import PhiloxNoise
    N, H, W = shape
    # (the 8x8 level of a training step takes the fused whole-image launches of resblock_img.hip, which keep fp32 storage; this test pins
    # the one-kernel-per-op bf16-storage path, which stays the path of the >= 16x16 levels and of every shape those kernels do not take)
    monkeypatch.setattr(K, '_RB_FWD_MIN_HW', 0)
    monkeypatch.setattr(K, '_RB_BWD_MIN_HW', 0)
    torch.manual_seed(2)
    from lvae_amd.arena import ParamArena
    blk = ResidualGatedBlock(64, 'elu', batchnorm=True, block_type='bacdbacd', dropout=0.2).cuda().train()
    arena = ParamArena(blk, torch.device('cuda'))      # packed weights ([KH][KW][Cin][Cout]) as in a model: what the fused kernels take
    x0 = torch.randn(N, H, W, 64, device='cuda')
    dout = torch.randn(N, H, W, 64, device='cuda')
    res = []
    used = []
    real = K.resblock_bf16_storage
    for storage in (False, True):
        monkeypatch.setattr(K, 'resblock_bf16_storage', (lambda *a: used.append(real(*a)) or used[-1]) if storage else (lambda *a: False))
        arena.zero_grad()
        x = x0.clone().requires_grad_(True)
        out = blk(x, PhiloxNoise(seed=3))
        out.backward(dout)
        torch.cuda.synchronize()
        res.append((out.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()}))
    assert used == [True], 'the shape is meant to take the bf16-storage kernels'
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-20))
    (o0, dx0, g0), (o1, dx1, g1) = res
    assert rel(o1, o0) < 4e-3 and rel(dx1, dx0) < 8e-3
    for k in g0:
        assert rel(g1[k], g0[k]) < 2e-2, (k, rel(g1[k], g0[k]))
