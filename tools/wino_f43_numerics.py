"""Numerics of Winograd F(4x4,3x3) against F(2x2,3x3) for the fp32 path (candidate 0): transforms in fp32 (as the kernels
do them), products and channel sums exact (float64 here; the kernels' six-product MFMAs are exact per product, fp32 accumulate), against a
float64 direct convolution. CPU only, numpy."""
import numpy as np

rng = np.random.default_rng(0)
C, K, H = 64, 64, 16


def winograd(x, w, BT, G, AT, m):
    """x [C][H][H] zero-padded conv 3x3, output tiles m x m; transforms rounded to fp32 after every matrix product."""
    r = 3
    a = m + r - 1
    f32 = np.float32
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1))).astype(f32)
    U = np.einsum('ij,kcjl,ml->kcim', G.astype(f32), w.astype(f32), G.astype(f32), optimize=False)   # exact in the kernels (weights are prepared in higher precision)
    U = (G.astype(np.float64) @ w.astype(np.float64) @ G.T.astype(np.float64)).astype(f32)
    y = np.zeros((K, H, H), np.float64)
    for ty in range(0, H, m):
        for tx in range(0, H, m):
            d = xp[:, ty:ty + a, tx:tx + a]
            t1 = np.einsum('ij,cjl->cil', BT.astype(f32), d).astype(f32)
            V = np.einsum('cil,ml->cim', t1, BT.astype(f32)).astype(f32)
            M = np.einsum('kcij,cij->kij', U.astype(np.float64), V.astype(np.float64))     # exact products, exact sums
            M = M.astype(f32)                                                              # fp32 accumulators
            t2 = np.einsum('ij,kjl->kil', AT.astype(f32), M).astype(f32)
            y[:, ty:ty + m, tx:tx + m] = np.einsum('kil,ml->kim', t2, AT.astype(f32)).astype(f32)
    return y


def direct(x, w):
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1)))
    y = np.zeros((K, H, H))
    for dy in range(3):
        for dx in range(3):
            y += np.einsum('kc,chw->khw', w[:, :, dy, dx].astype(np.float64), xp[:, dy:dy + H, dx:dx + H])
    return y


BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], float)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], float)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], float)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], float)

for trial in range(3):
    x = rng.standard_normal((C, H, H)).astype(np.float32)
    w = (rng.standard_normal((K, C, 3, 3)) / 24).astype(np.float32)
    ref = direct(x, w)
    # a plain fp32 direct convolution (fp32 accumulate in channel order) for scale
    acc = np.zeros((K, H, H), np.float32)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    for dy in range(3):
        for dx in range(3):
            for c in range(C):
                acc += w[:, c, dy, dx][:, None, None] * xp[c, dy:dy + H, dx:dx + H][None]
    rel = lambda y: np.linalg.norm(y - ref) / np.linalg.norm(ref)
    print('trial %d  fp32 direct %.2e   F(2x2,3x3) %.2e   F(4x4,3x3) %.2e' % (trial, rel(acc), rel(winograd(x, w, BT2, G2, AT2, 2)), rel(winograd(x, w, BT4, G4, AT4, 4))))
