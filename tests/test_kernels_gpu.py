"""Per-kernel parity on the GPU: each C-ABI entry point against a plain PyTorch fp32 CPU op of the same maths."""
import ctypes
import math

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
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).cpu()


def packed_weight(w, transposed=False):
    """the arena layout: physical [KH][KW][Cin][Cout], logical torch shape"""
    if transposed:  # (Cin,Cout,KH,KW)
        return w.permute(2, 3, 0, 1).contiguous().cuda().permute(2, 3, 0, 1)
    return w.permute(2, 3, 1, 0).contiguous().cuda().permute(3, 2, 0, 1)


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-20))


CONV_CASES = [
    # N, Cin, Cout, H, W, k, stride, pad
    (4, 64, 64, 16, 16, 3, 1, 1),
    (3, 64, 128, 8, 8, 1, 1, 0),
    (2, 32, 64, 4, 4, 3, 1, 1),
    (5, 64, 64, 16, 16, 3, 2, 1),
    (3, 3, 64, 32, 32, 5, 2, 2),
    (2, 1, 16, 28, 28, 5, 2, 2),
    (40, 3, 64, 32, 32, 5, 2, 2),    # stem: thin-input weight-gradient kernel (one workgroup per image)
    (33, 1, 32, 28, 28, 5, 2, 2),
    (2, 64, 100, 32, 32, 3, 1, 1),
    (3, 64, 1, 28, 28, 3, 1, 1),
    (2, 16, 6, 16, 16, 3, 1, 1),
    (70, 64, 64, 16, 16, 3, 1, 1),   # > 128 tiles: BM = 128 path
    (40, 64, 128, 16, 16, 1, 1, 0),  # BM = 128, BN = 128
    (2, 8, 8, 2, 2, 3, 1, 1),
    (3, 64, 64, 8, 8, 3, 1, 1),      # halo kernel: 2 images per tile, ragged last tile
    (37, 64, 64, 4, 4, 3, 1, 1),     # halo kernel: 4 or 8 images per tile
    (50, 32, 64, 2, 2, 3, 1, 1),
    (9, 64, 64, 32, 32, 3, 1, 1),    # 4-row tiles of a 32-wide image
    (2, 64, 64, 12, 12, 3, 1, 1),    # width that does not divide the tile
    (130, 64, 64, 16, 16, 3, 1, 1),
    (40, 128, 64, 16, 16, 1, 1, 0),  # 1x1 tile wgrad with two ci blocks
    (5, 64, 128, 4, 4, 1, 1, 0),
    (130, 64, 128, 16, 16, 1, 1, 0),  # direct (no LDS staging) 1x1 weight gradient, 4 co blocks
    (129, 64, 64, 16, 16, 1, 1, 0),   # ... 2 co blocks, ragged pixel ranges
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_dgrad_wgrad(K, case):
    N, Ci, Co, H, W, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)
    b = torch.randn(Co, generator=g)
    x.requires_grad_(True); w.requires_grad_(True); b.requires_grad_(True)
    y = F.conv2d(x, w, b, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    wp = packed_weight(w.detach())
    geom = K.ConvGeom(wp, stride=s, pad=p)
    yd = K.conv2d(nhwc(x.detach()), wp, geom, bias=b.detach().cuda())
    assert rel(nchw(yd), y.detach()) < 2e-6
    dx = K.conv2d_dgrad(nhwc(dy), wp, geom, (H, W))
    assert rel(nchw(dx), x.grad) < 2e-6
    dw = torch.zeros_like(wp)
    db = torch.zeros(Co, device='cuda')
    K.conv2d_wgrad(nhwc(x.detach()), nhwc(dy), wp, geom, dw, db)
    assert rel(dw.cpu(), w.grad) < 3e-6
    assert rel(db.cpu(), b.grad) < 3e-6
    # accumulate semantics
    K.conv2d_wgrad(nhwc(x.detach()), nhwc(dy), wp, geom, dw, db)
    assert rel(dw.cpu(), 2 * w.grad) < 3e-6
    # default (Cout,Cin,KH,KW)-contiguous layout goes through the scalar weight path
    wc = w.detach().cuda().contiguous()
    yd2 = K.conv2d(nhwc(x.detach()), wc, K.ConvGeom(wc, stride=s, pad=p), bias=b.detach().cuda())
    assert rel(nchw(yd2), y.detach()) < 2e-6


WINO_CASES = [
    (200, 64, 16, 16),   # one image per workgroup
    (49, 64, 32, 32),    # 8-row tiles of a 32-wide image
    (770, 64, 8, 8),     # four images per workgroup, ragged last group
    (193, 100, 16, 16),  # two output-channel tiles, the second ragged
    (260, 32, 24, 8),    # 192-pixel tile: 16 of the 64 Winograd tile slots stay unused
    (97, 128, 32, 16),   # four 32-channel blocks in the weight-gradient kernel, ranges of unequal length
    (64, 64, 16, 16),    # fewer than 256 pixel tiles: the 32-channel-per-workgroup variant with the pinned weight ring
    (257, 100, 8, 8),    # ... ragged last channel block, two images per tile, ragged last tile
    (300, 32, 8, 8),     # ... a single channel block
    (257, 64, 16, 16),   # ragged batch
    (65, 64, 32, 32),    # 8-row tiles of a 32-wide image
    (130, 100, 16, 16),  # two output-channel tiles, the second ragged
    (300, 32, 24, 16),   # 192-pixel tile: 16 of the 64 Winograd tile slots stay unused
]


@pytest.mark.parametrize('form', ['winograd', 'winograd6', 'split'])
@pytest.mark.parametrize('case', WINO_CASES)
def test_conv3x3_winograd(K, case, form):
    """Large 3x3 stride-1 layers in fp32 run as Winograd F(2x2,3x3) with the 16 position GEMMs either on the fp32 MFMA ('winograd',
    lvae_conv_desc.form = LVAE_FORM_F32_MFMA) or as six exact bf16-piece products per fp32 product on the bf16 MFMA ('winograd6', the
    default), or as a direct convolution in the six-product form ('split', LVAE_FORM_SIX_PRODUCT_DIRECT); all must agree with the direct
    sum to a few ulp of fp32, including the fused BN/activation prologue and dropout/activation epilogue."""
    with K.use_form({'winograd': K._C.FORM_F32_MFMA, 'winograd6': K._C.FORM_AUTO, 'split': K._C.FORM_SIX_PRODUCT_DIRECT}[form]):
        _conv3x3_winograd_case(K, case, form)


def _conv3x3_winograd_case(K, case, form):
    N, Co, H, W = case
    C = 64
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, C, H, W, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    w = torch.randn(Co, C, 3, 3, generator=g) / 24
    b = torch.randn(Co, generator=g)
    drop = (torch.rand(N, Co, generator=g) < 0.8).float() / 0.8
    xin = F.elu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    y = F.elu(F.conv2d(xin.double(), w.double(), b.double(), padding=1) * drop.view(N, Co, 1, 1).double()).float()
    wp = packed_weight(w)
    geom = K.ConvGeom(wp, 1, 1)
    d = K._desc(geom, wp, nhwc(x), None, N, H, W, H, W, Co, geom.s_ci, geom.s_co, K.GATHER_CONV)
    need = K._C.load().lvae_conv2d_workspace(ctypes.byref(d))
    if form != 'split':
        assert need > 0, "case is meant to exercise the Winograd path"
    if need:
        scratch = torch.empty(need, dtype=torch.uint8, device='cuda')
        d.workspace, d.workspace_bytes = scratch.data_ptr(), need
        var = K._C.load().lvae_conv2d_variant(ctypes.byref(d))
        V = K._C
        if form == 'winograd':
            assert var == V.VARIANT_WINO_F32
        elif form == 'winograd6':   # the six-product form needs 64 reduction channels and at least 256 pixel tiles
            assert var in (V.VARIANT_WINO_SIX, V.VARIANT_WINO_F32)
            if case in ((200, 64, 16, 16), (257, 64, 16, 16), (65, 64, 32, 32)):
                assert var == V.VARIANT_WINO_SIX
        else:
            assert var == V.VARIANT_SIX_DIRECT
    yd = K.conv2d(nhwc(x), wp, geom, bias=b.cuda(), in_scale=sc.cuda(), in_shift=sh.cuda(), in_act='elu',
                  out_scale=drop.cuda(), out_act='elu')
    assert rel(nchw(yd), y) < 4e-6
    dy = torch.randn(N, Co, H, W, generator=g)
    if Co >= 64:  # dgrad reduces over Cout: 64 -> the 64-channel kernel, 100 / 128 -> the 128-channel (zero padded) variant
        dx_ref = F.conv_transpose2d(dy.double(), w.double(), padding=1).float()
        mask = (torch.rand(N, C, generator=g) < 0.8).float() / 0.8
        dx = K.conv2d_dgrad(nhwc(dy), wp, geom, (H, W), out_scale=mask.cuda())
        assert rel(nchw(dx), dx_ref * mask.view(N, C, 1, 1)) < 4e-6
    # weight / bias gradient (Winograd-domain kernel when Cout is a multiple of 32 and W is 8, 16 or 32), fused prologue,
    # accumulating into a non-zero gradient
    xin64 = xin.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    F.conv2d(xin64, w64, b64, padding=1).backward(dy.double())
    dw0 = torch.randn(Co, C, 3, 3, generator=g) * 0.1
    dw = packed_weight(dw0)
    db0 = torch.randn(Co, generator=g)
    db = db0.cuda()
    K.conv2d_wgrad(nhwc(x), nhwc(dy), wp, geom, dw, db, in_scale=sc.cuda(), in_shift=sh.cuda(), in_act='elu')
    assert rel(dw.cpu() - dw0, w64.grad.float()) < 5e-6
    assert rel(db.cpu() - db0, b64.grad.float()) < 5e-6


def test_conv_wgrad_grouped(K):
    """lvae_conv2d_wgrad_grouped_f32 == the same gradients one by one (mixed 3x3 / 1x1 / large / odd shapes in one call)."""
    g = torch.Generator().manual_seed(33)
    specs = [(37, 64, 64, 4, 4, 3, 1), (50, 64, 64, 2, 2, 3, 1), (16, 64, 64, 4, 4, 3, 1), (20, 64, 128, 4, 4, 1, 0), (9, 128, 64, 2, 2, 1, 0),
             (8, 32, 64, 4, 4, 3, 1), (3, 3, 16, 8, 8, 5, 2), (12, 64, 64, 8, 8, 3, 1),
             (260, 64, 64, 8, 8, 3, 1), (300, 64, 64, 8, 8, 3, 1), (257, 64, 128, 8, 8, 3, 1)] + [(30 + i, 64, 64, 2, 2, 3, 1) for i in range(14)]
    items, refs = [], []
    for (N, Ci, Co, H, W, k, p) in specs:
        x = torch.randn(N, Ci, H, W, generator=g)
        w = torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)
        sc, sh = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.3
        dy = torch.randn(N, Co, H, W, generator=g)
        wp = packed_weight(w)
        geom = K.ConvGeom(wp, 1, p)
        kw = dict(in_scale=sc.cuda(), in_shift=sh.cuda(), in_act='elu') if Ci % 4 == 0 else {}
        dw0 = torch.randn(Co, Ci, k, k, generator=g) * 0.1
        db0 = torch.randn(Co, generator=g)
        one_w, one_b = packed_weight(dw0), db0.cuda()
        K.conv2d_wgrad(nhwc(x), nhwc(dy), wp, geom, one_w, one_b, **kw)
        refs.append((one_w, one_b))
        items.append((nhwc(x), nhwc(dy), wp, geom, packed_weight(dw0), db0.cuda(), kw))
    K.conv2d_wgrad_grouped(items)
    for (x, dy, wp, geom, dw, db, kw), (rw, rb) in zip(items, refs):
        assert torch.equal(dw, rw) and torch.equal(db, rb)   # same kernels, same summation order: bitwise equal


def test_prepared_weights_cache(K):
    """lvae_conv2d_prepare_weights: one batched transform serves later convolutions; any write to the weights (torch in-place
    op or a raw-pointer kernel announced through weights_written) makes the convolution transform them itself again."""
    g = torch.Generator().manual_seed(21)
    N, C, H, W = 200, 64, 16, 16
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / 24
    dy = torch.randn(N, C, H, W, generator=g)
    wp = packed_weight(w)
    geom = K.ConvGeom(wp, 1, 1)
    xd, dyd = nhwc(x), nhwc(dy)
    K.prepared.entries.clear()
    K.prepared.table = None
    y0, dx0 = K.conv2d(xd, wp, geom), K.conv2d_dgrad(dyd, wp, geom, (H, W))      # registers both orientations
    assert len(K.prepared.entries) == 2 and all(e['stamp'] is None for e in K.prepared.entries.values())
    assert K.prepared.prepare_all() == 2
    for e in K.prepared.entries.values():
        assert e['stamp'] == K.prepared.stamp(wp)
        e['U'].zero_()                                                           # a launch that re-transformed would not care
    K.prepared.prepare_all()
    y1, dx1 = K.conv2d(xd, wp, geom), K.conv2d_dgrad(dyd, wp, geom, (H, W))      # served from the prepared buffers
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
    wp.mul_(2.0)                                                                  # torch in-place write: version counter moves
    y2 = K.conv2d(xd, wp, geom)
    assert rel(y2, 2 * y0) < 1e-6
    K.prepared.prepare_all()
    K.prepared.weights_written()                                                  # e.g. the Adamax kernel ran
    for e in K.prepared.entries.values():
        e['U'].zero_()                                                           # stale garbage must not be read
    assert rel(K.conv2d(xd, wp, geom), 2 * y0) < 1e-6
    K.prepared.entries.clear()
    K.prepared.table = None


@pytest.mark.parametrize('case', [(3, 64, 64, 4, 4), (2, 16, 8, 8, 8), (40, 64, 64, 8, 8)])
def test_conv_transpose(K, case):
    N, Ci, Co, H, W = case
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, Ci, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Ci, Co, 3, 3, generator=g) / math.sqrt(Ci * 9)).requires_grad_(True)
    b = torch.randn(Co, generator=g, requires_grad=True)
    y = F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    wp = packed_weight(w.detach(), transposed=True)
    geom = K.ConvGeom(wp, stride=2, pad=1, transposed=True, output_padding=1)
    yd = K.conv2d(nhwc(x.detach()), wp, geom, bias=b.detach().cuda())
    assert tuple(yd.shape) == (N, 2 * H, 2 * W, Co)
    assert rel(nchw(yd), y.detach()) < 2e-6
    assert rel(nchw(K.conv2d_dgrad(nhwc(dy), wp, geom, (H, W))), x.grad) < 2e-6
    dw, db = torch.zeros_like(wp), torch.zeros(Co, device='cuda')
    K.conv2d_wgrad(nhwc(x.detach()), nhwc(dy), wp, geom, dw, db)
    assert rel(dw.cpu(), w.grad) < 3e-6 and rel(db.cpu(), b.grad) < 3e-6


def test_conv_fused_prologue_epilogue_and_cat(K):
    g = torch.Generator().manual_seed(3)
    N, C, H, W = 6, 64, 8, 8
    x1, x2 = torch.randn(N, C, H, W, generator=g), torch.randn(N, C, H, W, generator=g)
    sc, sh = torch.rand(2 * C, generator=g) + 0.5, torch.randn(2 * C, generator=g)
    w = torch.randn(C, 2 * C, 3, 3, generator=g) / 30
    b = torch.randn(C, generator=g)
    drop = (torch.rand(N, C, generator=g) < 0.8).float() / 0.8
    xin = F.elu(torch.cat((x1, x2), 1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y = F.elu(F.conv2d(xin, wr, b, padding=1) * drop.view(N, C, 1, 1))
    wp = packed_weight(w)
    geom = K.ConvGeom(wp, 1, 1)
    yd = K.conv2d(nhwc(x1), wp, geom, bias=b.cuda(), x2=nhwc(x2), in_scale=sc.cuda(), in_shift=sh.cuda(), in_act='elu',
                  out_scale=drop.cuda(), out_act='elu')
    assert rel(nchw(yd), y.detach()) < 2e-6
    dy = torch.randn(y.shape, generator=g)
    pre = F.conv2d(xin, wr, b, padding=1)
    pre.backward(dy)
    dw = torch.zeros_like(wp)
    K.conv2d_wgrad(nhwc(x1), nhwc(dy), wp, geom, dw, None, x2=nhwc(x2), in_scale=sc.cuda(), in_shift=sh.cuda(), in_act='elu')
    assert rel(dw.cpu(), wr.grad) < 3e-6
    dx = K.conv2d_dgrad(nhwc(dy), wp, geom, (H, W))
    assert rel(nchw(dx), xin.grad) < 2e-6



@pytest.mark.parametrize('case', [(256, 64, 64, 64, 4, 4), (33, 64, 64, 64, 16, 16), (5, 32, 96, 64, 8, 8), (70, 64, 64, 128, 2, 2)])
def test_conv1x1_dgrad_of_a_channel_concat_in_one_launch(K, case):
    """lvae_conv1x1_dgrad_cat_f32: both halves of the input gradient of MergeLayer's 1x1 convolution over cat(x, x2) == the two launches with a
    weight offset (bitwise: same kernel, same reduction order), and == F.conv_transpose2d."""
    N, C1, C2, Co, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    w = torch.randn(Co, C1 + C2, 1, 1, generator=g) / math.sqrt(C1 + C2)
    dy = torch.randn(N, Co, H, W, generator=g)
    wp = packed_weight(w)
    geom = K.ConvGeom(wp, 1, 0)
    both = K.conv1x1_dgrad_cat(nhwc(dy), wp, geom, C1)
    assert both is not None
    dx, dx2 = both
    a = K.conv2d_dgrad(nhwc(dy), wp, geom, (H, W), ci_range=(0, C1))
    b = K.conv2d_dgrad(nhwc(dy), wp, geom, (H, W), ci_range=(C1, C1 + C2))
    assert torch.equal(dx, a) and torch.equal(dx2, b)
    ref = F.conv_transpose2d(dy.double(), w.double())
    assert rel(nchw(dx).double(), ref[:, :C1]) < 3e-6 and rel(nchw(dx2).double(), ref[:, C1:]) < 3e-6


@pytest.mark.parametrize('shape', [(8, 64, 16, 16), (3, 8, 5, 7), (4, 3, 6, 6), (64, 64, 32, 32)])
def test_bn_stats_and_affine_bwd(K, shape):
    g = torch.Generator().manual_seed(5)
    N, C, H, W = shape
    x = (torch.randn(shape, generator=g) * 2 + 3).requires_grad_(True)   # large mean: exercises the pivoted variance
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(C, generator=g).requires_grad_(True)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    h = F.elu(F.batch_norm(x, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5))
    dh = torch.randn(shape, generator=g)
    h.backward(dh)
    xd = nhwc(x.detach())
    rmd, rvd = rm.cuda(), rv.cuda()
    sc, sh, mean, rstd = K.bn_stats(xd, gamma.detach().cuda(), beta.detach().cuda(), rmd, rvd)
    torch.testing.assert_close(rmd.cpu(), rm_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rvd.cpu(), rv_ref, rtol=1e-5, atol=1e-6)
    hd = K.affine_act(xd, sc, sh, 'elu')
    assert rel(nchw(hd), h.detach()) < 2e-6
    dg, db = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    dx = K.affine_act_bwd(nhwc(dh), xd, sc, sh, 'elu', True, mean, rstd, dg, db)
    assert rel(nchw(dx), x.grad) < 1e-5
    assert rel(dg.cpu(), gamma.grad) < 1e-5 and rel(db.cpu(), beta.grad) < 1e-5


@pytest.mark.parametrize('form', ['split', 'mfma_f32'])
@pytest.mark.parametrize('shape', [(40, 64, 16, 16), (3, 64, 4, 4), (2, 8, 2, 2), (70, 32, 8, 8),
                                   (129, 64, 16, 16),   # persistent kernel: 516 tiles over 512 workgroups (four of them take two)
                                   (37, 64, 3, 3)])     # ... ragged last tile
def test_conv1x1_gate_fused(K, shape, form, monkeypatch):
    """GateLayer2d forward fused with its 1x1 convolution; both fp32 forms of the persistent kernel (six exact bf16-piece products per
    fp32 product on the bf16 MFMA, lvae_conv_desc.form = LVAE_FORM_SIX_PRODUCT, and the fp32 MFMA, the default)."""
    monkeypatch.setattr(K, 'form', K._C.FORM_F32_MFMA if form == 'mfma_f32' else K._C.FORM_SIX_PRODUCT)
    N, C, H, W = shape
    g = torch.Generator().manual_seed(11)
    x, res = torch.randn(N, C, H, W, generator=g), torch.randn(N, C, H, W, generator=g)
    w, b = torch.randn(2 * C, C, 1, 1, generator=g) / math.sqrt(C), torch.randn(2 * C, generator=g)
    ab = F.conv2d(x, w, b)
    a_, b_ = ab.chunk(2, 1)
    out = F.elu(a_) * torch.sigmoid(b_) + res
    wp = packed_weight(w)
    abd, outd = K.conv1x1_gate(nhwc(x), wp, K.ConvGeom(wp, 1, 0), b.cuda(), nhwc(res), 'elu')
    assert rel(nchw(abd), ab) < 2e-6 and rel(nchw(outd), out) < 2e-6
    # without the pre-activations (evaluation) and without a residual
    abn, outn = K.conv1x1_gate(nhwc(x), wp, K.ConvGeom(wp, 1, 0), b.cuda(), None, 'elu', need_ab=False)
    assert abn is None and rel(nchw(outn), out - res) < 2e-6


@pytest.mark.parametrize('shape', [(40, 64, 16, 16), (129, 64, 16, 16), (37, 64, 3, 3)])
def test_conv1x1_gate_fused_bf16_operands(K, shape):
    """GateLayer2d forward at precision bf16: x and W rounded to bf16 at the matrix-core inputs, fp32 accumulate, fp32 gate / residual /
    statistics. Against the fp64 result over operands rounded beforehand."""
    N, C, H, W = shape
    g = torch.Generator().manual_seed(12)
    x, res = torch.randn(N, C, H, W, generator=g), torch.randn(N, C, H, W, generator=g)
    w, b = torch.randn(2 * C, C, 1, 1, generator=g) / math.sqrt(C), torch.randn(2 * C, generator=g)
    bf = lambda t_: t_.bfloat16().double()
    ab = F.conv2d(bf(x), bf(w), b.double())
    a_, b_ = ab.chunk(2, 1)
    out = F.elu(a_) * torch.sigmoid(b_) + res.double()
    wp = packed_weight(w)
    pivot = torch.randn(C, generator=g).cuda()
    K.set_precision('bf16')
    try:
        abd, outd, parts = K.conv1x1_gate(nhwc(x), wp, K.ConvGeom(wp, 1, 0), b.cuda(), nhwc(res), 'elu', stats_pivot=pivot)
    finally:
        K.set_precision('f32')
    assert rel(nchw(abd), ab.float()) < 2e-6 and rel(nchw(outd), out.float()) < 2e-6
    d = outd.reshape(-1, C).double() - pivot.double()
    pr = parts.rows_view().double()
    assert rel(pr[:, 0].sum(0), d.sum(0)) < 1e-5 and rel(pr[:, 1].sum(0), (d * d).sum(0)) < 1e-5


@pytest.mark.parametrize('shape', [(40, 64, 16, 16), (3, 64, 4, 4), (70, 32, 8, 8), (5, 8, 2, 2)])
def test_conv1x1_gate_bwd_fused(K, shape):
    """gate backward + 1x1 dgrad in one kernel == lvae_gate_bwd_f32 followed by lvae_conv2d_f32 (and both == autograd)."""
    N, C, H, W = shape
    g = torch.Generator().manual_seed(12)
    x = torch.randn(N, C, H, W, generator=g, requires_grad=True)
    w = (torch.randn(2 * C, C, 1, 1, generator=g) / math.sqrt(C))
    b = torch.randn(2 * C, generator=g)
    ab = F.conv2d(x, w, b)
    a_, b_ = ab.chunk(2, 1)
    out = F.elu(a_) * torch.sigmoid(b_)
    dout = torch.randn(out.shape, generator=g)
    mask = (torch.rand(N, C, generator=g) < 0.8).float() / 0.8
    out.backward(dout)
    wp = packed_weight(w)
    geom = K.ConvGeom(wp, 1, 0)
    abd, doutd = nhwc(ab.detach()), nhwc(dout)
    dab, dx = K.conv1x1_gate_bwd(doutd, abd, wp, geom, 'elu', out_scale=mask.cuda())
    dab_ref = K.gate_bwd(doutd, abd, 'elu')
    assert rel(dab, dab_ref) < 1e-6
    assert rel(dx, K.conv2d_dgrad(dab_ref, wp, geom, (H, W), out_scale=mask.cuda())) < 2e-6
    assert rel(nchw(dx), x.grad * mask.view(N, C, 1, 1)) < 3e-6


@pytest.mark.parametrize('shape', [(256, 256), (7, 70), (1, 3), (300, 1)])
def test_colsum(K, shape):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(shape, generator=g)
    out0 = torch.randn(shape[1], generator=g)
    out = out0.cuda()
    K.colsum(x.cuda(), out, True)
    torch.testing.assert_close(out.cpu(), out0 + x.sum(0), rtol=1e-5, atol=1e-5)
    K.colsum(x.cuda(), out, False)
    torch.testing.assert_close(out.cpu(), x.sum(0), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('case', [(200, 64, 64, 16, 16), (37, 64, 64, 4, 4), (70, 32, 64, 8, 8), (9, 64, 100, 32, 32), (258, 64, 64, 16, 16),
                                  (66, 64, 64, 16, 16), (259, 64, 100, 8, 8)])   # last two: 32-channel Winograd workgroups
def test_bn_statistics_from_conv_epilogue(K, case):
    """conv2d(..., stats_pivot) + bn_finalize_parts == conv2d followed by bn_stats on its output (Winograd and tile kernels)."""
    N, Ci, Co, H, W = case
    g = torch.Generator().manual_seed(44)
    x = nhwc(torch.randn(N, Ci, H, W, generator=g))
    wp = packed_weight(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci))
    b = (torch.randn(Co, generator=g) + 2.0).cuda()          # output mean far from the pivot
    drop = ((torch.rand(N, Co, generator=g) < 0.8).float() / 0.8).cuda()
    gamma, beta = (torch.rand(Co, generator=g) + 0.5).cuda(), torch.randn(Co, generator=g).cuda()
    rm, rv = (torch.randn(Co, generator=g) * 0.1).cuda(), (torch.rand(Co, generator=g) + 0.5).cuda()
    rm2, rv2 = rm.clone(), rv.clone()
    geom = K.ConvGeom(wp, 1, 1)
    y, parts = K.conv2d(x, wp, geom, bias=b, out_scale=drop, stats_pivot=rm)
    assert parts is not None, "this shape is meant to take a kernel with the statistics epilogue"
    got = K.bn_finalize_parts(parts.rows_view(), N * H * W, rm, gamma, beta, rm, rv)
    ref = K.bn_stats(y, gamma, beta, rm2, rv2)
    for a_, b_ in zip(got, ref):
        torch.testing.assert_close(a_, b_, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(rm, rm2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv, rv2, rtol=1e-5, atol=1e-6)
    assert torch.equal(y, K.conv2d(x, wp, geom, bias=b, out_scale=drop))


@pytest.mark.parametrize('shape', [(40, 64, 16, 16), (3, 64, 4, 4), (70, 32, 8, 8), (129, 64, 16, 16), (37, 64, 3, 3)])
def test_bn_statistics_from_gate_epilogue(K, shape):
    """conv1x1_gate(..., stats_pivot) + bn_finalize_parts == bn_stats on the gate output (the next block's BatchNorm input)."""
    N, C, H, W = shape
    g = torch.Generator().manual_seed(45)
    x, res = nhwc(torch.randn(N, C, H, W, generator=g)), nhwc(torch.randn(N, C, H, W, generator=g) + 1.5)
    wp = packed_weight(torch.randn(2 * C, C, 1, 1, generator=g) / math.sqrt(C))
    b = torch.randn(2 * C, generator=g).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    rm, rv = (torch.randn(C, generator=g) * 0.1).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    rm2, rv2 = rm.clone(), rv.clone()
    pivot = torch.randn(C, generator=g).cuda()
    ab, out, parts = K.conv1x1_gate(x, wp, K.ConvGeom(wp, 1, 0), b, res, 'elu', stats_pivot=pivot)
    assert parts is not None
    assert parts.has_pivot and torch.equal(parts.buf[parts.rows, 0], pivot)      # the pivot travels behind the partial rows
    got = K.bn_finalize_parts(parts.rows_view(), N * H * W, pivot, gamma, beta, rm, rv)
    ref = K.bn_stats(out, gamma, beta, rm2, rv2)
    for a_, b_ in zip(got, ref):
        torch.testing.assert_close(a_, b_, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(rm, rm2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv, rv2, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('case', [(200, 64, 64, 16, 16), (37, 64, 64, 4, 4), (70, 64, 32, 8, 8), (66, 64, 64, 16, 16), (259, 64, 64, 8, 8), (258, 64, 64, 16, 16)])
def test_bn_backward_sums_from_dgrad_epilogue(K, case):
    """conv2d_dgrad(..., bn_bwd) + affine_act_bwd_parts == conv2d_dgrad followed by affine_act_bwd (Winograd and tile kernels)."""
    N, Ci, Co, H, W = case
    g = torch.Generator().manual_seed(46)
    xb = nhwc(torch.randn(N, Ci, H, W, generator=g) * 2 + 1)
    dy = nhwc(torch.randn(N, Co, H, W, generator=g))
    wp = packed_weight(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci))
    gamma, beta = (torch.rand(Ci, generator=g) + 0.5).cuda(), torch.randn(Ci, generator=g).cuda()
    sc, sh, mean, rstd = K.bn_stats(xb, gamma, beta, None, None)
    assert K.bn_coef_block(sc, sh, mean, rstd)
    drop = ((torch.rand(N, Ci, generator=g) < 0.8).float() / 0.8).cuda()
    add = nhwc(torch.randn(N, Ci, H, W, generator=g))
    geom = K.ConvGeom(wp, 1, 1)
    dh, parts = K.conv2d_dgrad(dy, wp, geom, (H, W), bn_bwd=(xb, sc, 'elu'))
    assert parts is not None, "this shape is meant to take a kernel with the statistics epilogue"
    assert torch.equal(dh, K.conv2d_dgrad(dy, wp, geom, (H, W)))
    dg, db = torch.zeros(Ci, device='cuda'), torch.zeros(Ci, device='cuda')
    dg2, db2 = torch.zeros(Ci, device='cuda'), torch.zeros(Ci, device='cuda')
    got = K.affine_act_bwd_parts(parts, dh, xb, sc, sh, 'elu', mean, rstd, dg, db, drop=drop, add=add)
    ref = K.affine_act_bwd(dh, xb, sc, sh, 'elu', True, mean, rstd, dg2, db2, drop=drop, add=add)
    assert rel(got, ref) < 2e-6
    assert rel(dg, dg2) < 1e-5 and rel(db, db2) < 1e-5


def test_gate(K):
    g = torch.Generator().manual_seed(6)
    ab = torch.randn(5, 128, 4, 4, generator=g, requires_grad=True)
    res = torch.randn(5, 64, 4, 4, generator=g)
    a, b = ab.chunk(2, 1)
    out = F.elu(a) * torch.sigmoid(b) + res
    do = torch.randn(out.shape, generator=g)
    out.backward(do)
    abd = nhwc(ab.detach())
    assert rel(nchw(K.gate_fwd(abd, nhwc(res), 'elu')), out.detach()) < 1e-6
    assert rel(nchw(K.gate_bwd(nhwc(do), abd, 'elu')), ab.grad) < 1e-6


@pytest.mark.parametrize('analytical', [False, True])
@pytest.mark.parametrize('top', [False, True])
def test_normal_stochastic(K, analytical, top):
    from oracle import lvae_ref as R
    g = torch.Generator().manual_seed(8)
    N, Z, H, W = 6, 32, 4, 4
    p = (torch.randn(1 if top else N, 2 * Z, H, W, generator=g) * 0.5).requires_grad_(True)
    q = (torch.randn(N, 2 * Z, H, W, generator=g) * 0.5).requires_grad_(True)
    eps = torch.randn(N, Z, H, W, generator=g)
    pmu, plv = p.chunk(2, 1)
    qmu, qlv = q.chunk(2, 1)
    z = qmu + (qlv / 2).exp() * eps
    lp = R.normal_log_prob(z, pmu, plv).sum((1, 2, 3))
    lq = R.normal_log_prob(z, qmu, qlv).sum((1, 2, 3))
    kan = R.normal_kl(qmu, qlv, pmu, plv)
    kl = kan.sum((1, 2, 3)) if analytical else (R.normal_log_prob(z, qmu, qlv) - R.normal_log_prob(z, pmu, plv)).sum((1, 2, 3))
    ks = kan.sum(1)
    dz, glp, glq, gkl, gks = (torch.randn(z.shape, generator=g), torch.randn(N, generator=g), torch.randn(N, generator=g),
                              torch.randn(N, generator=g), torch.randn(ks.shape, generator=g))
    ((z * dz).sum() + (lp * glp).sum() + (lq * glq).sum() + (kl * gkl).sum() + (ks * gks).sum()).backward()
    pd, qd, ed = nhwc(p.detach()), nhwc(q.detach()), nhwc(eps)
    zd, lpd, lqd, kld, ksd = K.normal_stochastic_fwd(pd, qd, ed, 0, analytical, Z, N)
    torch.testing.assert_close(nchw(zd), z.detach(), rtol=1e-5, atol=1e-5)
    for a, b in ((lpd, lp), (lqd, lq), (kld, kl)):
        torch.testing.assert_close(a.cpu(), b.detach(), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(ksd.cpu(), ks.detach(), rtol=1e-5, atol=1e-4)
    dp, dq = K.normal_stochastic_bwd(pd, qd, ed, zd, nhwc(dz), glp.cuda(), glq.cuda(), gkl.cuda(), gks.cuda(), 0, analytical, Z)
    dpc = nchw(dp).sum(0, keepdim=True) if top else nchw(dp)
    assert rel(dpc, p.grad) < 1e-5 and rel(nchw(dq), q.grad) < 1e-5


def test_likelihood_golden_vectors(K):
    from conftest import load_golden
    g = load_golden('ops')
    mean, xb = g.t('bern.mean'), g.t('bern.x')
    logits = torch.log(mean.clamp(1e-30)) - torch.log1p(-mean.clamp(max=1 - 1e-7))
    # Bernoulli: compare through logits that reproduce the (unsaturated) means; saturation is tested below
    m, mode, smp, ll, dll = K.bernoulli_fwd(nhwc(logits), nhwc(xb), nhwc(torch.rand_like(xb)), True)
    keep = (mean > 1e-6) & (mean < 1 - 1e-6)
    torch.testing.assert_close(nchw(m)[keep], mean[keep], rtol=1e-5, atol=1e-6)
    big = torch.tensor([[40., -40., 200., -200.]]).view(1, 1, 2, 2)
    xs = torch.tensor([[0., 1., 0., 1.]]).view(1, 1, 2, 2)
    _, _, _, ll_s, _ = K.bernoulli_fwd(nhwc(big), nhwc(xs), nhwc(torch.rand_like(xs)), False)
    ref = -F.binary_cross_entropy(torch.sigmoid(big), xs, reduction='none').sum()
    torch.testing.assert_close(ll_s.cpu()[0], ref, rtol=1e-6, atol=1e-4)
    assert float(ref) < -150
    # DMoL log-likelihood, gradient and sampler against the reference's outputs
    l, xd = g.t('dmol.l'), g.t('dmol.x')
    ll, dl = K.dmol_ll_fwd(nhwc(l), nhwc(xd), True)
    torch.testing.assert_close(ll.cpu(), g.t('dmol.ll'), rtol=1e-5, atol=1e-4)
    # Gradient: this stress vector sits where cdf_delta ~ 1e-5 (sigmoid(plus) - sigmoid(min) cancels to 2-3 digits in fp32), so the
    # REFERENCE's own fp32 gradient is only good to ~1e-2 there. Measured, not assumed: the same formula in float64 is the
    # yardstick; the kernel must be at least as close to it as the reference's fp32 result is (and close to the reference within
    # the reference's own error).
    from oracle import lvae_ref as R
    l64 = l.double().requires_grad_(True)
    R.discretized_mix_logistic_ll(xd.double() * 2 - 1, l64).sum().backward()
    ref_err = float((g.t('dmol.dl').double() - l64.grad).abs().max())
    our_err = float((nchw(dl).double() - l64.grad).abs().max())
    assert ref_err > 1e-4, ref_err            # the vector really is ill-conditioned in fp32
    assert our_err <= 2.0 * ref_err + 1e-5, (our_err, ref_err)
    torch.testing.assert_close(nchw(dl), g.t('dmol.dl'), rtol=1e-4, atol=3.0 * ref_err)
    tape = g.seq('dmol.tape')
    s = K.dmol_sample(nhwc(l), tape[0].cuda().contiguous(), tape[1].cuda().contiguous())
    torch.testing.assert_close(nchw(s) * 2 - 1, g.t('dmol.sample'), rtol=1e-5, atol=1e-5)


def test_discretized_logistic_golden_vector(K):
    from conftest import load_golden
    g = load_golden('ops')
    mean, ls, x = g.t('dlog.mean'), g.t('dlog.ls'), g.t('dlog.x')
    raw = torch.cat((mean - 0.5, ls + 1.0), dim=1)   # the kernel applies mean + 0.5 and logscale - 1 itself
    m, l, smp, ll, dll = K.discr_logistic_fwd(nhwc(raw), nhwc(x), nhwc(torch.rand_like(x).clamp(1e-6, 1 - 1e-6)), True)
    torch.testing.assert_close(ll.cpu(), g.t('dlog.ll'), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(nchw(l), ls.clamp(min=-7.), rtol=1e-6, atol=1e-6)


def test_upsample_pad_crop(K):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 8, 5, 6, generator=g, requires_grad=True)
    y = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    assert rel(nchw(K.upsample2x_fwd(nhwc(x.detach()))), y.detach()) < 1e-6
    assert rel(nchw(K.upsample2x_bwd(nhwc(dy))), x.grad) < 1e-6
    from oracle import lvae_ref as R
    xi = torch.randn(2, 3, 28, 27, generator=g)
    pad = K.pad_crop(xi.cuda(), True, (32, 32), False)
    torch.testing.assert_close(nchw(pad), R.pad_img_tensor(xi, (32, 32)))
    crop = K.pad_crop(pad, False, (28, 27), True)
    torch.testing.assert_close(crop.cpu(), xi)


def test_kl_bookkeeping_elbo_adamax_l2(K):
    from oracle import lvae_ref as R
    g = torch.Generator().manual_seed(10)
    L, N = 5, 37
    kl = (torch.rand(N, L, generator=g) * 2).requires_grad_(True)
    ll = (-torch.rand(N, generator=g) * 100).requires_grad_(True)
    for fb in (0.0, 0.7):
        kl.grad = ll.grad = None
        kl_loss = R.free_bits_kl(kl, fb).sum()
        kl_sep = kl.sum(1)
        loss = (-ll).mean() + 0.3 * kl_loss
        (loss * 1.7).backward()
        kl_ln = kl.detach().t().contiguous().cuda()
        ksep, kavg, scal = K.kl_bookkeeping_fwd(kl_ln, fb)
        torch.testing.assert_close(ksep.cpu(), kl_sep.detach())
        torch.testing.assert_close(kavg.cpu(), kl.detach().mean(0))
        torch.testing.assert_close(scal.cpu(), torch.stack((kl_loss.detach(), kl_sep.detach().mean())))
        esep, s3 = K.elbo_loss_fwd(ll.detach().cuda(), ksep, scal[0:1], 0.3)
        torch.testing.assert_close(s3.cpu()[0], loss.detach())
        torch.testing.assert_close(esep.cpu(), (ll - kl_sep).detach())
        d_ll, d_kll = K.elbo_loss_bwd(torch.tensor([1.7], device='cuda'), 0.3, N)
        torch.testing.assert_close(d_ll.cpu(), ll.grad)
        gs = torch.cat((d_kll, torch.zeros(1, device='cuda')))
        dkl = K.kl_bookkeeping_bwd(kl_ln, fb, None, None, gs)
        torch.testing.assert_close(dkl.t().cpu(), kl.grad)
    # Adamax against torch.optim.Adamax, three steps
    n = 1000
    p0, grads = torch.randn(n, generator=g), [torch.randn(n, generator=g) for _ in range(3)]
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adamax([pr], lr=3e-4)
    pd, m, u = p0.cuda(), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    step = torch.zeros(1, dtype=torch.int64, device='cuda')
    for gr in grads:
        pr.grad = gr.clone()
        opt.step()
        K.adamax_step(pd, gr.cuda(), m, u, None, 3e-4, 0.9, 0.999, 1e-8, 0.0, None, step)
        K.counter_advance(step)
    torch.testing.assert_close(pd.cpu(), pr.detach(), rtol=1e-6, atol=1e-7)
    assert int(step.item()) == 3
    torch.testing.assert_close(K.l2norm(pd).cpu()[0], pr.detach().norm(), rtol=1e-6, atol=0)


def test_rng_statistics(K):
    n = 1 << 20
    off = torch.zeros(1, dtype=torch.int64, device='cuda')
    a = K.rng_fill(torch.empty(n, device='cuda'), 'normal', 0, 0, 1234, off, 1)
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.std()) - 1) < 5e-3
    u = K.rng_fill(torch.empty(n, device='cuda'), 'uniform', 1e-5, 1 - 1e-5, 1234, off, 2)
    assert 0 < float(u.min()) and float(u.max()) < 1 and abs(float(u.mean()) - 0.5) < 2e-3
    b = K.rng_fill(torch.empty(n, device='cuda'), 'bernoulli', 0.8, 1.25, 1234, off, 3)
    assert abs(float((b > 0).float().mean()) - 0.8) < 2e-3 and float(b.max()) == 1.25
    a2 = K.rng_fill(torch.empty(n, device='cuda'), 'normal', 0, 0, 1234, off, 1)
    assert torch.equal(a, a2)              # same (seed, step, call site) -> same numbers
    K.counter_advance(off)
    a3 = K.rng_fill(torch.empty(n, device='cuda'), 'normal', 0, 0, 1234, off, 1)
    assert abs(float((a * a3).mean())) < 5e-3  # next step: fresh, uncorrelated


POS_CASES = [
    # N, Cin, Cout, H, W
    (256, 64, 64, 4, 4), (256, 64, 64, 2, 2), (37, 64, 64, 4, 4), (5, 32, 64, 2, 2), (70, 64, 64, 3, 4), (33, 64, 100, 1, 1),
    (64, 64, 32, 4, 2),
]


@pytest.mark.parametrize('case', POS_CASES)
def test_conv3x3_position_major_small_images(K, case):
    """Low-resolution levels (H*W <= 16): the position-major kernel — forward and dgrad with fused BN+ELU prologue, bias, dropout
    scale, statistics epilogues — against torch's direct convolution (lib/nn.py:83-89 call sites)."""
    N, Ci, Co, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)
    b = torch.randn(Co, generator=g) * 0.1
    sc, sh = 1 + 0.1 * torch.randn(Ci, generator=g), 0.1 * torch.randn(Ci, generator=g)
    drop = (torch.rand(N, Co, generator=g) < 0.8).float() / 0.8
    piv = 0.05 * torch.randn(Co, generator=g)
    wp = packed_weight(w)
    geom = K.ConvGeom(wp, 1, 1)
    ref = F.conv2d(F.elu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), w, b, padding=1) * drop.view(N, Co, 1, 1)
    y, parts = K.conv2d(nhwc(x), wp, geom, bias=b.cuda(), in_scale=sc.cuda(), in_shift=sh.cuda(), in_act='elu', out_scale=drop.cuda(),
                        stats_pivot=piv.cuda())
    assert parts is not None and parts.has_pivot and parts.rows == ((N + 31) // 32) * H * W     # the position-major kernel ran
    assert rel(nchw(y), ref) < 2e-6
    s = parts.rows_view().sum(0).cpu().double()
    dl = (ref.double() - piv.double().view(1, -1, 1, 1))
    torch.testing.assert_close(s[0], dl.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(s[1], (dl * dl).sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    assert torch.equal(parts.buf[parts.rows, 0].cpu(), piv)
    # dgrad (transposed gather, k-contiguous weights) with the Dropout2d scale of the producer
    dy = torch.randn(N, Co, H, W, generator=g)
    dref = F.conv_transpose2d(dy, w, padding=1)
    dx = K.conv2d_dgrad(nhwc(dy), wp, geom, (H, W))
    assert rel(nchw(dx), dref) < 2e-6


@pytest.mark.parametrize('case', [(256, 64, 4, 4), (96, 64, 2, 2), (40, 32, 4, 4),                   # position-major kernel
                                  (256, 64, 16, 16), (256, 64, 8, 8), (64, 64, 24, 24), (96, 48, 16, 16)])  # Winograd kernels: 256-pixel
                                  # six-product, 32-channel fp32, 128-pixel six-product, 48 of 64 padded channels
def test_bn_finalize_folded_into_the_consuming_convolution(K, case, monkeypatch):
    """lvae_bn_fold: conv1 writes BatchNorm partials (+ its pivot) of its output; conv2 finalizes them in its prologue, publishes
    (scale, shift, mean, rstd) and updates the running statistics — against nn.BatchNorm2d semantics (lib/nn.py:80-81)."""
    import types
    N, Cc, H, W = case

    def no_finalize_launch(*a, **k):
        raise AssertionError('the consuming convolution did not take the folded finalize')
    monkeypatch.setattr(K, 'bn_finalize_parts', no_finalize_launch)
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cc, H, W, generator=g)
    w1 = torch.randn(Cc, Cc, 3, 3, generator=g) / (3 * Cc ** 0.5)
    w2 = torch.randn(64, Cc, 3, 3, generator=g) / (3 * Cc ** 0.5)
    gamma, beta = 1 + 0.1 * torch.randn(Cc, generator=g), 0.1 * torch.randn(Cc, generator=g)
    rm0, rv0 = 0.1 * torch.randn(Cc, generator=g), 1 + 0.2 * torch.rand(Cc, generator=g)
    wp1, wp2 = packed_weight(w1), packed_weight(w2)
    g1, g2 = K.ConvGeom(wp1, 1, 1), K.ConvGeom(wp2, 1, 1)
    bn = types.SimpleNamespace(weight=gamma.cuda(), bias=beta.cuda(), running_mean=rm0.cuda(), running_var=rv0.cuda(), eps=1e-5, momentum=0.1)
    y1, parts = K.conv2d(nhwc(x), wp1, g1, stats_pivot=bn.running_mean)
    assert parts is not None and parts.has_pivot
    y2, _, (sc, sh, mean, rstd) = K.conv2d(y1, wp2, g2, in_act='elu', in_bn=(parts, bn.running_mean, bn))
    r1 = F.conv2d(x, w1, None, padding=1)
    rmr, rvr = rm0.clone(), rv0.clone()
    h = F.batch_norm(r1, rmr, rvr, gamma, beta, True, 0.1, 1e-5)
    r2 = F.conv2d(F.elu(h), w2, None, padding=1)
    assert rel(nchw(y2), r2) < (5e-6 if H * W <= 16 else 2e-5)
    torch.testing.assert_close(mean.cpu(), r1.mean((0, 2, 3)), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rstd.cpu(), 1 / torch.sqrt(r1.var((0, 2, 3), unbiased=False) + 1e-5), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(sc.cpu(), gamma * rstd.cpu(), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(bn.running_mean.cpu(), rmr, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn.running_var.cpu(), rvr, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('case', [(200, 64, 64, 16, 16), (49, 64, 64, 32, 32), (300, 32, 64, 8, 8), (70, 64, 100, 16, 16)])
def test_conv3x3_bf16_operands(K, case):
    """precision = bf16 (lvae_conv2d_bf16): operands rounded to bf16, exact products, fp32 accumulation — against torch's fp32
    convolution on inputs that were rounded to bf16 beforehand (then the only difference is summation order), and within bf16
    rounding of the unrounded fp32 result. Forward with the fused BN+ELU prologue, dgrad with the Dropout2d scale."""
    N, Ci, Co, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)
    b = torch.randn(Co, generator=g) * 0.1
    sc, sh = 1 + 0.1 * torch.randn(Ci, generator=g), 0.1 * torch.randn(Ci, generator=g)
    wp = packed_weight(w)
    geom = K.ConvGeom(wp, 1, 1)
    bf = lambda t: t.bfloat16().float()
    xin = F.elu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    ref_rounded = F.conv2d(bf(xin).double(), bf(w).double(), b.double(), padding=1).float()
    ref_full = F.conv2d(xin, w, b, padding=1)
    K.set_precision('bf16')
    try:
        y, parts = K.conv2d(nhwc(x), wp, geom, bias=b.cuda(), in_scale=sc.cuda(), in_shift=sh.cuda(), in_act='elu',
                            stats_pivot=torch.zeros(Co, device='cuda'))
        dy = torch.randn(N, Co, H, W, generator=g)
        dx = K.conv2d_dgrad(nhwc(dy), wp, geom, (H, W)) if Co <= 64 else None
    finally:
        K.set_precision('f32')
    # the fused transform is evaluated in fp32 on the GPU (fast exp): a value that lands within 1e-6 of a bf16 rounding boundary may
    # round the other way than on the CPU, so the "same rounded inputs" comparison holds to ~1e-4 rather than to fp32 ulps
    assert rel(nchw(y), ref_rounded) < 2e-4
    assert rel(nchw(y), ref_full) < 6e-3
    assert parts is not None
    s = parts.rows_view().sum(0).cpu()
    torch.testing.assert_close(s[0], nchw(y).sum((0, 2, 3)), rtol=1e-4, atol=1e-2)
    if dx is not None:
        dref = F.conv_transpose2d(bf(dy).double(), bf(w).double(), padding=1).float()
        assert rel(nchw(dx), dref) < 1e-5


@pytest.mark.parametrize('form', ['split', 'mfma_f32'])
@pytest.mark.parametrize('shape', [(256, 16, 16), (70, 16, 16), (33, 32, 32), (257, 8, 8)])
def test_conv1x1_gate_bwd_with_fused_weight_gradient(K, shape, form, monkeypatch):
    """lvae_conv1x1_gate_bwd_wgrad_f32 (gate derivative + dgrad + weight / bias gradient of the gate convolution, one persistent
    kernel) == autograd of lib/nn.py:118-126, accumulating into non-zero gradient buffers. Both fp32 forms: six exact bf16-piece
    products per fp32 product on the bf16 MFMA (default) and the fp32 MFMA (lvae_conv_desc.form = LVAE_FORM_F32_MFMA)."""
    monkeypatch.setattr(K, 'form', K._C.FORM_F32_MFMA if form == 'mfma_f32' else K._C.FORM_AUTO)
    N, H, W = shape
    C = 64
    g = torch.Generator().manual_seed(N + H)
    x = torch.randn(N, C, H, W, generator=g, requires_grad=True)
    w = (torch.randn(2 * C, C, 1, 1, generator=g) / math.sqrt(C)).requires_grad_(True)
    b = torch.randn(2 * C, generator=g).requires_grad_(True)
    ab = F.conv2d(x, w, b)
    a_, b_ = ab.chunk(2, 1)
    out = F.elu(a_) * torch.sigmoid(b_)
    dout = torch.randn(out.shape, generator=g)
    mask = (torch.rand(N, C, generator=g) < 0.8).float() / 0.8
    out.backward(dout)
    wp = packed_weight(w.detach())
    geom = K.ConvGeom(wp, 1, 0)
    dw0, db0 = torch.randn(2 * C, C, 1, 1, generator=g) * 0.1, torch.randn(2 * C, generator=g)
    dw, db = packed_weight(dw0), db0.cuda()
    dx = K.conv1x1_gate_bwd_wgrad(nhwc(dout), nhwc(ab.detach()), nhwc(x.detach()), wp, geom, 'elu', dw, db, out_scale=mask.cuda())
    assert dx is not None, "shape is meant to take the fused kernel"
    assert rel(nchw(dx), x.grad * mask.view(N, C, 1, 1)) < 3e-6
    assert rel(dw.cpu() - dw0, w.grad) < 5e-6
    assert rel(db.cpu() - db0, b.grad) < 5e-6
    # small layers are not taken (the caller composes gate_bwd + wgrad)
    small = K.conv1x1_gate_bwd_wgrad(nhwc(dout[:4]), nhwc(ab.detach()[:4]), nhwc(x.detach()[:4]), wp, geom, 'elu', dw, db) if H * W * 4 < 16384 else None
    assert small is None


@pytest.mark.parametrize('shape', [(256, 16, 16), (67, 16, 16), (300, 8, 8), (19, 32, 32)])
def test_conv1x1_gate_bwd_fused_bf16_operands(K, shape):
    """The fused GateLayer2d backward at precision bf16: dab, the saved convolution input and W are rounded to bf16 at the matrix-core
    inputs (fp32 accumulate), the gate derivative and the bias gradient stay fp32. Against fp64 sums over operands rounded beforehand."""
    N, H, W = shape
    C = 64
    g = torch.Generator().manual_seed(N + H + 1)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(2 * C, C, 1, 1, generator=g) / math.sqrt(C)
    b = torch.randn(2 * C, generator=g)
    ab = F.conv2d(x, w, b)
    a_, b_ = ab.double().chunk(2, 1)
    dout = torch.randn(N, C, H, W, generator=g)
    mask = (torch.rand(N, C, generator=g) < 0.8).float() / 0.8
    sg = torch.sigmoid(b_)
    dab = torch.cat([dout.double() * sg * torch.where(a_ > 0, torch.ones_like(a_), a_.exp()),
                     dout.double() * F.elu(a_) * sg * (1 - sg)], 1)                       # fp64 gate derivative (lib/nn.py:118-126)
    bf = lambda t_: t_.float().bfloat16().double()
    w2 = bf(w[:, :, 0, 0])                                                                  # [2C][C]
    dx_ref = torch.einsum('nkhw,kc->nchw', bf(dab), w2) * mask.view(N, C, 1, 1).double()
    dw_ref = torch.einsum('nkhw,nchw->kc', bf(dab), bf(x))
    db_ref = dab.sum((0, 2, 3))
    wp = packed_weight(w)
    geom = K.ConvGeom(wp, 1, 0)
    dw0, db0 = torch.randn(2 * C, C, 1, 1, generator=g) * 0.1, torch.randn(2 * C, generator=g)
    dw, db = packed_weight(dw0), db0.cuda()
    K.set_precision('bf16')
    try:
        dx = K.conv1x1_gate_bwd_wgrad(nhwc(dout), nhwc(ab), nhwc(x), wp, geom, 'elu', dw, db, out_scale=mask.cuda())
    finally:
        K.set_precision('f32')
    assert dx is not None, "shape is meant to take the fused kernel"
    # 3e-4: an operand that sits on a bf16 rounding boundary may round the other way on the GPU (fast exp in the gate derivative)
    assert rel(nchw(dx), dx_ref.float()) < 3e-4
    assert rel((dw.cpu() - dw0)[:, :, 0, 0], dw_ref.float()) < 3e-4
    assert rel(db.cpu() - db0, db_ref.float()) < 5e-6


@pytest.mark.parametrize('case', [(256, 64, 64, 16, 16), (33, 64, 64, 32, 32), (300, 64, 64, 8, 8), (70, 32, 64, 16, 16), (40, 64, 100, 32, 32),
                                  (301, 64, 64, 16, 16), (1025, 64, 64, 8, 8), (67, 64, 64, 32, 32)])   # quadrant form: ragged ranges, a half-empty last tile
def test_conv3x3_wgrad_bf16_operands(K, case):
    """Weight / bias gradient with bf16 matrix-core operands (transposed LDS reads, persistent accumulators): against the fp64 sum
    over operands that were rounded to bf16 beforehand, accumulating into non-zero gradient buffers; fused BN+ELU prologue."""
    N, Ci, Co, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Ci, H, W, generator=g)
    dy = torch.randn(N, Co, H, W, generator=g)
    sc, sh = 1 + 0.1 * torch.randn(Ci, generator=g), 0.1 * torch.randn(Ci, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / 24
    wp = packed_weight(w)
    geom = K.ConvGeom(wp, 1, 1)
    bf = lambda t: t.bfloat16().double()
    xin = F.elu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    xin64 = bf(xin).requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    F.conv2d(xin64, w64, None, padding=1).backward(bf(dy))
    dw0, db0 = torch.randn(Co, Ci, 3, 3, generator=g) * 0.1, torch.randn(Co, generator=g)
    dw, db = packed_weight(dw0), db0.cuda()
    K.set_precision('bf16')
    try:
        d = K._desc(geom, wp, nhwc(x), None, N, H, W, H, W, Co, geom.s_ci, geom.s_co, K.GATHER_CONV)
        assert K._C.load().lvae_conv2d_wgrad_workspace(ctypes.byref(d)) > 0
        K.conv2d_wgrad(nhwc(x), nhwc(dy), wp, geom, dw, db, in_scale=sc.cuda(), in_shift=sh.cuda(), in_act='elu')
    finally:
        K.set_precision('f32')
    assert rel(dw.cpu() - dw0, w64.grad.float()) < 3e-4     # bf16 rounding boundaries of the GPU's fast-exp ELU vs the CPU's
    assert rel(db.cpu() - db0, dy.double().sum((0, 2, 3)).float()) < 1e-5    # the bias gradient sums the unrounded dy


IMG_WGRAD_CASES = [
    # N, Cin, Cout, H, W, k: the <= 8x8 levels' weight gradients on whole-image tiles (csrc/conv_wgrad_img.hip)
    (256, 64, 64, 8, 8, 3), (256, 64, 64, 4, 4, 3), (256, 64, 64, 2, 2, 3),      # as in the CIFAR-15 step: 64 / 16 / 4 slabs
    (37, 64, 64, 4, 4, 3), (50, 32, 64, 2, 2, 3), (3, 64, 64, 8, 8, 3), (1, 64, 64, 2, 2, 3),   # ragged last tile, thin input, one image
    (21, 64, 48, 4, 8, 3), (19, 24, 64, 8, 4, 3),                                # non-square images, channel counts below 64
    (256, 64, 128, 8, 8, 1), (77, 64, 128, 4, 4, 1), (130, 64, 128, 2, 2, 1), (9, 48, 100, 4, 4, 1),   # GateLayer2d's 1x1 (64 -> 128)
]


@pytest.mark.parametrize('prec', ['f32', 'bf16'])
@pytest.mark.parametrize('case', IMG_WGRAD_CASES)
def test_conv_wgrad_whole_image_tiles(K, case, prec):
    """lvae_conv2d_wgrad_f32 on the whole-image-tile kernel (transposed LDS reads, six-product / bf16 operands, persistent accumulators,
    fixed-order slab reduce) against the float64 sum: fused BatchNorm + ELU prologue (3x3), accumulation into non-zero buffers."""
    N, Ci, Co, H, W, k = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Ci, H, W, generator=g)
    dy = torch.randn(N, Co, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)
    wp = packed_weight(w)
    geom = K.ConvGeom(wp, 1, k // 2)
    kw = {}
    xin = x
    if k == 3:
        sc, sh = 1 + 0.1 * torch.randn(Ci, generator=g), 0.1 * torch.randn(Ci, generator=g)
        kw = dict(in_scale=sc.cuda(), in_shift=sh.cuda(), in_act='elu')
        xin = F.elu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    rnd = (lambda t: t.bfloat16().double()) if prec == 'bf16' else (lambda t: t.double())
    xin64 = rnd(xin).requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    F.conv2d(xin64, w64, None, padding=k // 2).backward(rnd(dy))
    dw0, db0 = torch.randn(Co, Ci, k, k, generator=g) * 0.1, torch.randn(Co, generator=g)
    dw, db = packed_weight(dw0), db0.cuda()
    K.set_precision(prec)
    try:
        d = K._desc(geom, wp, nhwc(x), None, N, H, W, H, W, Co, geom.s_ci, geom.s_co, K.GATHER_CONV)
        assert K._C.load().lvae_conv2d_wgrad_variant(ctypes.byref(d)) == K._C.WGRAD_VARIANT_IMG
        K.conv2d_wgrad(nhwc(x), nhwc(dy), wp, geom, dw, db, **kw)
    finally:
        K.set_precision('f32')
    tol = 3e-4 if prec == 'bf16' else 3e-6   # bf16: rounding boundaries of the GPU's fast-exp ELU vs the CPU's
    assert rel(dw.cpu() - dw0, w64.grad.float()) < tol
    assert rel(db.cpu() - db0, dy.double().sum((0, 2, 3)).float()) < 3e-6   # the bias gradient sums the unrounded dy


@pytest.mark.parametrize('shape', [(256, 16, 16), (70, 16, 16), (130, 16, 16), (64, 32, 32), (19, 32, 32)])
def test_weight_gradient_that_absorbs_the_batchnorm_apply_in_front_of_it(K, shape):
    """lvae_conv2d_wgrad_apply_f32 == lvae_affine_act_bwd_parts_f32 followed by lvae_conv2d_wgrad_f32: the stored dy (to 1e-6: the partial rows
    are summed in another order), dgamma / dbeta, and the weight / bias gradient."""
    N, H, W = shape
    C = 64
    g = torch.Generator().manual_seed(N + 7 * H)
    rn = lambda *s_: torch.randn(*s_, generator=g).cuda()
    x, dh, xbn = rn(N, H, W, C), rn(N, H, W, C), rn(N, H, W, C)
    w = packed_weight(torch.randn(C, C, 3, 3, generator=g) / 24)
    geom = K.ConvGeom(w, 1, 1)
    assert K.conv2d_wgrad_apply_ok(x, w, geom)
    coef_in = K.bn_stats(x, None, None, None, None)
    coef = K.bn_stats(xbn, torch.rand(C, generator=g).cuda() + 0.5, rn(C) * 0.1, None, None)
    u = xbn * coef[0] + coef[1]
    gg = dh * torch.where(u > 0, torch.ones_like(u), torch.exp(u))
    xh = (xbn - coef[2]) * coef[3]
    rows = 256
    parts = torch.stack([torch.stack([a.sum(0), b.sum(0)]) for a, b in zip(gg.reshape(-1, C).tensor_split(rows), (gg * xh).reshape(-1, C).tensor_split(rows))]).contiguous()
    drop = ((torch.rand(N, C, generator=g) < 0.8).float() / 0.8).cuda()
    dg_a, db_a = torch.full((C,), 0.5, device='cuda'), torch.full((C,), -1.0, device='cuda')
    dg_b, db_b = dg_a.clone(), db_a.clone()
    dy_ref = K.affine_act_bwd_parts(parts, dh, xbn, coef[0], coef[1], 'elu', coef[2], coef[3], dg_a, db_a, drop=drop)
    dw_a, dbias_a = torch.zeros_like(w), torch.zeros(C, device='cuda')
    K.conv2d_wgrad(x, dy_ref, w, geom, dw_a, dbias_a, in_scale=coef_in[0], in_shift=coef_in[1], in_act='elu')
    dw_b, dbias_b = torch.zeros_like(w), torch.zeros(C, device='cuda')
    dy = K.conv2d_wgrad_apply(x, w, geom, dw_b, dbias_b, parts, dh, xbn, coef[0], 'elu', dg_b, db_b, drop=drop, in_scale=coef_in[0], in_shift=coef_in[1], in_act='elu')
    torch.cuda.synchronize()
    assert rel(dy.cpu(), dy_ref.cpu()) < 1e-6
    torch.testing.assert_close(dg_b, dg_a, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(db_b, db_a, rtol=1e-5, atol=1e-4)
    assert rel(dw_b.cpu(), dw_a.cpu()) < 2e-6 and rel(dbias_b.cpu(), dbias_a.cpu()) < 2e-6


@pytest.mark.parametrize('nmix', [1, 5, 10, 16, 20])
def test_dmol_any_component_count_matches_oracle(K, nmix):
    """DiscretizedLogisticMixLikelihood(n_components) of lib/likelihoods.py:183-202 takes any count (the reference only ever builds 10):
    log-likelihood, its gradient and the sampler for the counts the kernels are instantiated for, against the oracle's restatement of
    lib/likelihoods.py:291-382 and lib/stochastic.py:141-206 (whose 10-component case is pinned by the reference's own vectors)."""
    from oracle import lvae_ref as R
    g = torch.Generator().manual_seed(100 + nmix)
    N, H, W = 3, 8, 8
    l = torch.randn(N, 10 * nmix, H, W, generator=g)
    x01 = torch.floor(256 * torch.rand(N, 3, H, W, generator=g)) / 255
    x01[0, :, 0, 0], x01[0, :, 0, 1] = 0.0, 1.0            # the two edge bins
    l64 = l.double().requires_grad_(True)
    ll_ref = R.discretized_mix_logistic_ll((x01 * 2 - 1).double(), l64)
    ll_ref.sum().backward()
    ll, dl = K.dmol_ll_fwd(nhwc(l), nhwc(x01), True)
    torch.testing.assert_close(ll.cpu().double(), ll_ref.detach(), rtol=1e-5, atol=1e-3)
    assert rel(nchw(dl).double(), l64.grad) < 2e-4
    tape = R.Tape(gen=torch.Generator().manual_seed(7))
    s_ref = R.sample_discretized_mix_logistic(l, tape)
    u_mix, u_log = [torch.as_tensor(e).float().cuda().contiguous() for e in tape.entries]
    s = K.dmol_sample(nhwc(l), u_mix, u_log)
    torch.testing.assert_close(nchw(s) * 2 - 1, s_ref, rtol=1e-5, atol=1e-5)
    with pytest.raises(K._C.LvaeHipError):
        K.dmol_ll_fwd(nhwc(torch.randn(1, 70, 4, 4)), nhwc(torch.rand(1, 3, 4, 4)), False)      # 7 components: not instantiated
