"""Kernel forms that must agree to fp32 accuracy, on random shapes (formerly tools/fuzz_forms.py), and their documented behaviour on
non-finite / extreme operands (include/lvae_hip.h, `precision` and `form`):

 * Winograd position GEMMs on the fp32 MFMA (form = LVAE_FORM_F32_MFMA) vs the six-product form on the bf16 MFMA (default), forward + dgrad
 * persistent gate forward vs a float64 reference, incl. its BatchNorm partials
 * fused gate backward: six-product (default) vs fp32 MFMA
 * BatchNorm finalize folded into the consuming convolution's prologue (lvae_bn_fold) vs the finalize launch + in_scale / in_shift
 * +-inf, NaN, values beyond the bf16 range (3.39e38 .. FLT_MAX) and denormals through the six-product kernels
"""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def K():
    import lvae_amd  # noqa: F401
    from lvae_amd import kernels
    kernels._C.load()
    return kernels


def packed(co, ci, k, gen=None):
    return (torch.randn(k, k, ci, co, device='cuda') / (ci * k * k) ** 0.5).permute(3, 2, 0, 1)


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-20))


def _both_forms(K, fn):
    outs = []
    for f in (K._C.FORM_F32_MFMA, K._C.FORM_AUTO):
        with K.use_form(f):
            K.prepared.entries.clear()
            K.prepared.table = None
            outs.append(fn().clone())
    return outs


@pytest.mark.parametrize('seed', [0, 1])
def test_winograd_forms_agree_on_random_shapes(K, seed):
    random.seed(seed)
    torch.manual_seed(seed)
    worst = 0.0
    for it in range(20):
        H = random.choice([8, 16, 32])
        W = random.choice([8, 16, 32])
        N = random.randint(max(1, 16384 // (H * W)), max(2, 70000 // (H * W)))
        Co = random.choice([32, 64, 100, 128])
        C = 64
        x = torch.randn(N, H, W, C, device='cuda')
        w = packed(Co, C, 3)
        g = K.ConvGeom(w, 1, 1)
        b = torch.randn(Co, device='cuda')
        sc = torch.rand(C, device='cuda') + 0.5
        sh = torch.randn(C, device='cuda') * 0.3
        drop = (torch.rand(N, Co, device='cuda') < 0.8).float() / 0.8
        o = _both_forms(K, lambda: K.conv2d(x, w, g, bias=b, in_scale=sc, in_shift=sh, in_act='elu', out_scale=drop, out_act='elu'))
        e = rel(o[1], o[0])
        worst = max(worst, e)
        assert e < 4e-6, ('wino fwd', N, H, W, Co, e)
        if Co in (64, 128):
            dy = torch.randn(N, H, W, Co, device='cuda')
            o = _both_forms(K, lambda: K.conv2d_dgrad(dy, w, g, (H, W)))
            e = rel(o[1], o[0])
            worst = max(worst, e)
            assert e < 4e-6, ('wino dgrad', N, H, W, Co, e)
    print('winograd forms: worst relative difference %.2e' % worst)


def test_gate_forward_on_random_shapes(K):
    random.seed(3)
    torch.manual_seed(3)
    for it in range(30):
        H = random.choice([2, 3, 4, 8, 16])
        N = random.randint(1, max(2, 70000 // (H * H)))
        C = 64
        x = torch.randn(N, H, H, C, device='cuda')
        res = torch.randn(N, H, H, C, device='cuda')
        w = packed(2 * C, C, 1)
        g = K.ConvGeom(w, 1, 0)
        b = torch.randn(2 * C, device='cuda')
        piv = torch.randn(C, device='cuda')
        ab, out, parts = K.conv1x1_gate(x, w, g, b, res, 'elu', stats_pivot=piv)
        abr = (x.reshape(-1, C).double() @ w[:, :, 0, 0].t().double() + b.double())
        outr = (torch.nn.functional.elu(abr[:, :C]) * torch.sigmoid(abr[:, C:]) + res.reshape(-1, C).double())
        e = max(rel(ab.reshape(-1, 2 * C).double(), abr), rel(out.reshape(-1, C).double(), outr))
        assert e < 3e-6, ('gate fwd', N, H, e)
        d = out.reshape(-1, C).double() - piv.double()
        pr = parts.rows_view().double()
        e = max(rel(pr[:, 0].sum(0), d.sum(0)), rel(pr[:, 1].sum(0), (d * d).sum(0)))
        assert e < 1e-5, ('gate stats', N, H, e)


def test_fused_gate_backward_forms_agree_on_random_shapes(K):
    random.seed(5)
    torch.manual_seed(5)
    for it in range(20):
        H = random.choice([8, 16, 32])
        N = random.randint(max(1, 16384 // (H * H)) + 1, max(3, 70000 // (H * H)))
        C = 64
        dout = torch.randn(N, H, H, C, device='cuda')
        ab = torch.randn(N, H, H, 2 * C, device='cuda')
        y = torch.randn(N, H, H, C, device='cuda')
        w = packed(2 * C, C, 1)
        g = K.ConvGeom(w, 1, 0)
        res = []
        for f in (K._C.FORM_F32_MFMA, K._C.FORM_AUTO):
            with K.use_form(f):
                dw, db = torch.zeros_like(w), torch.zeros(2 * C, device='cuda')
                dx = K.conv1x1_gate_bwd_wgrad(dout, ab, y, w, g, 'elu', dw, db)
                assert dx is not None
                res.append((dx.clone(), dw.clone(), db.clone()))
        e = max(rel(res[1][0], res[0][0]), rel(res[1][1], res[0][1]), rel(res[1][2], res[0][2]))
        assert e < 5e-6, ('gate bwd', N, H, e)

def test_folded_bn_finalize_agrees_with_the_finalize_launch_on_random_shapes(K):
    """lvae_bn_fold (position-major and Winograd kernels) against lvae_bn_finalize_parts_f32 + in_scale / in_shift on the same partial
    sums: output, published coefficients and running statistics (nn.BatchNorm2d of lib/nn.py:80-81; both are restatements of one
    formula with different summation orders, so they agree to fp32 rounding, not bitwise)."""
    import types
    random.seed(7)
    torch.manual_seed(7)
    lib = K._C.load()
    folded = 0
    for it in range(24):
        H = W = random.choice([2, 4, 8, 16, 24])
        C = random.choice([36, 48, 64]) if H >= 8 else random.choice([32, 64])
        lo = max(1, (16384 + H * W - 1) // (H * W)) if H >= 8 else 8
        N = random.randint(lo, max(lo + 1, 70000 // (H * W)))
        Co = random.choice([32, 64])
        x = torch.randn(N, H, W, C, device='cuda')
        w1, w2 = packed(C, C, 3), packed(Co, C, 3)
        g1, g2 = K.ConvGeom(w1, 1, 1), K.ConvGeom(w2, 1, 1)
        gamma, beta = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.2
        rm0, rv0 = torch.randn(C, device='cuda') * 0.1, torch.rand(C, device='cuda') + 0.5

        def bn():
            return types.SimpleNamespace(weight=gamma, bias=beta, running_mean=rm0.clone(), running_var=rv0.clone(), eps=1e-5, momentum=0.1)
        pivot = rm0.clone()
        y1, parts = K.conv2d(x, w1, g1, stats_pivot=pivot)
        if parts is None:
            continue
        # the ABI's allocation rule: lvae_conv2d_stats_buffer_rows = partial rows (+ the pivot row where the kernel stores one)
        assert parts.buf.shape[0] == parts.rows + int(parts.has_pivot) and parts.buf[parts.rows:].shape[0] == int(parts.has_pivot)
        if parts.has_pivot:
            assert torch.equal(parts.buf[parts.rows, 0], pivot)
        M = N * H * W
        bn_a, bn_b = bn(), bn()
        ya, _, coef_a = K.conv2d(y1, w2, g2, in_act='elu', in_bn=(parts, pivot, bn_a))
        coef_b = K.bn_finalize_parts(parts.rows_view(), M, pivot, gamma, beta, bn_b.running_mean, bn_b.running_var, 1e-5, 0.1)
        yb = K.conv2d(y1, w2, g2, in_scale=coef_b[0], in_shift=coef_b[1], in_act='elu')
        folded += int(parts.has_pivot)
        assert rel(ya, yb) < 2e-6, ('fold output', N, H, C, Co)
        for a, b, what in zip(coef_a, coef_b, ('scale', 'shift', 'mean', 'rstd')):
            torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-7, msg=lambda m: '%s %s: %s' % (what, (N, H, C, Co), m))
        torch.testing.assert_close(bn_a.running_mean, bn_b.running_mean, rtol=2e-6, atol=2e-7)
        torch.testing.assert_close(bn_a.running_var, bn_b.running_var, rtol=2e-6, atol=2e-7)
    assert folded >= 12, folded   # most of the drawn shapes take the folded path (has_pivot <=> lvae_conv2d_folds_bn_finalize)



# ------------------------------------------------------------------------------------------------------------------------------------
# Documented behaviour on special operands (include/lvae_hip.h, lvae_conv_desc.precision):
#   * Winograd F(2x2,3x3) — either form — forms +-combinations of the 4x4 input block of a tile before multiplying, so ONE non-finite
#     input makes every output of every tile whose 4x4 block contains it non-finite (inf - inf = NaN already in the fp32 transform);
#     outputs of all other tiles are bit-identical to the run without it.
#   * the six-product split turns an INFINITE operand into NaN (inf - inf in the remainder), where an fp32 product would be +-inf;
#     finite operands of any magnitude keep their fp32 result, also beyond the bf16 range (3.39e38 .. FLT_MAX: measured on gfx950, the
#     f32 -> bf16 conversion does not overflow there and the remainder pieces carry the rest).
#   * denormal operands: products of pieces far below FLT_MIN; results stay finite and within 1e-37 (absolute) of the clean result.
# ------------------------------------------------------------------------------------------------------------------------------------
SPECIALS = [float('inf'), float('-inf'), float('nan')]


@pytest.mark.parametrize('form', ['f32_mfma', 'six'])
@pytest.mark.parametrize('special', SPECIALS)
def test_winograd_non_finite_operand_stays_inside_its_tiles(K, form, special):
    torch.manual_seed(9)
    N, H, W, C = 70, 16, 16, 64
    x = torch.randn(N, H, W, C, device='cuda')
    w = packed(C, C, 3)
    g = K.ConvGeom(w, 1, 1)
    n0, h0, w0, c0 = 5, 7, 9, 11
    with K.use_form(K._C.FORM_F32_MFMA if form == 'f32_mfma' else K._C.FORM_AUTO):
        K.prepared.entries.clear()
        K.prepared.table = None
        clean = K.conv2d(x, w, g).clone()
        xs = x.clone()
        xs[n0, h0, w0, c0] = special
        out = K.conv2d(xs, w, g).clone()
    # tiles (2x2 outputs at even offsets) whose 4x4 input block (rows 2ty-1 .. 2ty+2) contains pixel (h0, w0)
    touched = torch.zeros(N, H, W, dtype=torch.bool, device='cuda')
    for ty in range(H // 2):
        for tx in range(W // 2):
            if 2 * ty - 1 <= h0 <= 2 * ty + 2 and 2 * tx - 1 <= w0 <= 2 * tx + 2:
                touched[n0, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = True
    assert torch.equal(out[~touched], clean[~touched]), 'a special operand leaked outside the Winograd tiles that contain it'
    field = torch.zeros(N, H, W, dtype=torch.bool, device='cuda')
    field[n0, max(0, h0 - 1):h0 + 2, max(0, w0 - 1):w0 + 2] = True      # the 3x3 receptive fields that really use the pixel
    assert not torch.isfinite(out[field]).all(dim=-1).any(), 'outputs that use the special operand must be non-finite'


@pytest.mark.parametrize('form', ['f32_mfma', 'six'])
def test_winograd_large_finite_and_denormal_operands(K, form):
    """Finite operands beyond the bf16 range (bf16 max = 3.3895e38 < 3.4e38 < FLT_MAX): both forms must give the fp32 result (weights
    small enough that no fp32 sum overflows); denormal inputs give finite results equal to float64 to fp32 accuracy, and do not disturb
    any other tile."""
    torch.manual_seed(10)
    N, H, W, C = 70, 16, 16, 64
    x = torch.randn(N, H, W, C, device='cuda')
    w = packed(C, C, 3) * 1e-3
    g = K.ConvGeom(w, 1, 1)
    with K.use_form(K._C.FORM_F32_MFMA if form == 'f32_mfma' else K._C.FORM_AUTO):
        K.prepared.entries.clear()
        K.prepared.table = None
        xs = x.clone()
        xs[3, 8, 8, 5] = 3.4e38               # one huge entry per 4x4 block: the +-combinations of the input transform stay below FLT_MAX
        out = K.conv2d(xs, w, g)
        ref = torch.nn.functional.conv2d(xs.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
        assert torch.isfinite(out).all()
        big = ref[3, 7:10, 7:10].abs().max()
        assert float((out[3, 7:10, 7:10].double() - ref[3, 7:10, 7:10]).abs().max()) < 4e-6 * float(big)
        clean = K.conv2d(x, w, g).clone()
        xd = x.clone()
        xd[4, 2:6, 2:6, :8] = 1e-40             # denormals
        outd = K.conv2d(xd, w, g)
        refd = torch.nn.functional.conv2d(xd.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
        assert torch.isfinite(outd).all()
        assert rel(outd.double(), refd) < 4e-6
        far = torch.ones(N, H, W, dtype=torch.bool, device='cuda')
        far[4, 0:8, 0:8] = False
        assert torch.equal(outd[far], clean[far])


@pytest.mark.parametrize('special', SPECIALS)
def test_fused_gate_backward_non_finite_operand(K, special):
    """Six-product fused gate backward: a non-finite dout element makes its own pixel row of dx non-finite and (through the weight
    gradient, a sum over all pixels) the weight-gradient entries of its channel; every other dx row is bit-identical to the clean run.
    The fp32-MFMA form must behave the same way except that it may produce +-inf where the six-product form produces NaN."""
    torch.manual_seed(11)
    N, H, C = 70, 16, 64
    dout = torch.randn(N, H, H, C, device='cuda')
    ab = torch.randn(N, H, H, 2 * C, device='cuda')
    y = torch.randn(N, H, H, C, device='cuda')
    w = packed(2 * C, C, 1)
    g = K.ConvGeom(w, 1, 0)
    for f in (K._C.FORM_F32_MFMA, K._C.FORM_AUTO):
        with K.use_form(f):
            dw, db = torch.zeros_like(w), torch.zeros(2 * C, device='cuda')
            clean = K.conv1x1_gate_bwd_wgrad(dout, ab, y, w, g, 'elu', dw, db).clone()
            ds = dout.clone()
            ds[6, 3, 4, 17] = special
            dw2, db2 = torch.zeros_like(w), torch.zeros(2 * C, device='cuda')
            dx = K.conv1x1_gate_bwd_wgrad(ds, ab, y, w, g, 'elu', dw2, db2)
            rows = torch.ones(N, H, H, dtype=torch.bool, device='cuda')
            rows[6, 3, 4] = False
            assert torch.equal(dx[rows], clean[rows])
            assert not torch.isfinite(dx[6, 3, 4]).all()
            assert not torch.isfinite(db2[17]) or not torch.isfinite(db2[17 + C])


def test_fused_gate_backward_operand_beyond_bf16_range(K):
    """3.4e38 in dout (finite in fp32, beyond bf16 max): both forms give finite dx rows equal to the float64 product to fp32 accuracy."""
    torch.manual_seed(12)
    N, H, C = 70, 16, 64
    dout = torch.randn(N, H, H, C, device='cuda')
    ab = torch.randn(N, H, H, 2 * C, device='cuda')
    y = torch.randn(N, H, H, C, device='cuda')
    w = packed(2 * C, C, 1) * 1e-2
    g = K.ConvGeom(w, 1, 0)
    dout[6, 3, 4, 17] = 3.4e38
    a_, b_ = ab[6, 3, 4, :C].double(), ab[6, 3, 4, C:].double()
    sig = torch.sigmoid(b_)
    da = dout[6, 3, 4].double() * sig * torch.where(a_ > 0, torch.ones_like(a_), a_.exp())
    db_ = dout[6, 3, 4].double() * torch.nn.functional.elu(a_) * sig * (1 - sig)
    ref = torch.cat([da, db_]) @ w[:, :, 0, 0].double()       # dx[m, ci] = sum_co dab[m, co] * W[co, ci]
    for f in (K._C.FORM_F32_MFMA, K._C.FORM_AUTO):
        with K.use_form(f):
            dw, db = torch.zeros_like(w), torch.zeros(2 * C, device='cuda')
            dx = K.conv1x1_gate_bwd_wgrad(dout, ab, y, w, g, 'elu', dw, db)
            assert torch.isfinite(dx).all()
            assert rel(dx[6, 3, 4].double(), ref) < 1e-5
