"""Fused residual-block kernels (resblock_img.hip) against the one-kernel-per-op launches they replace, per level at batch 256:
HIP-event time of 50 back-to-back repetitions of each chain. Profiling helper."""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvae_amd  # noqa: F401
from lvae_amd import kernels as K


def packed(co, ci, k):
    return torch.randn(k, k, ci, co, device='cuda').permute(3, 2, 0, 1) * 0.05


def timeit(fn, n=20, reps=10):
    """fn captured n times back to back in one hipGraph (the host cannot keep up with 5-10 us kernels in eager mode), replayed reps times"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


def main():
    B, C = int(os.environ.get('RB_BATCH', '256')), 64
    dev = 'cuda'
    for H in (8, 4, 2):
        x = torch.randn(B, H, H, C, device=dev)
        dout = torch.randn(B, H, H, C, device=dev)
        w1, w2, wg = packed(C, C, 3), packed(C, C, 3), packed(2 * C, C, 1)
        g1, g2, gg = K.ConvGeom(w1, 1, 1), K.ConvGeom(w2, 1, 1), K.ConvGeom(wg, 1, 0)
        b1, b2, bg = torch.randn(C, device=dev), torch.randn(C, device=dev), torch.randn(2 * C, device=dev)
        m1 = (torch.rand(B, C, device=dev) < 0.8).float() / 0.8
        m2 = (torch.rand(B, C, device=dev) < 0.8).float() / 0.8
        mk = lambda: types.SimpleNamespace(weight=torch.ones(C, device=dev), bias=torch.zeros(C, device=dev), running_mean=torch.zeros(C, device=dev),
                                           running_var=torch.ones(C, device=dev), eps=1e-5, momentum=0.1)
        bn1, bn2 = mk(), mk()
        coef1 = K.bn_stats(x, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var)
        K.prepared.prepare_all()
        # a previous gate kernel's partials of x (so that conv1 can fold its finalize, as inside a chain of blocks)
        _, _, xparts = K.conv1x1_gate(x, wg, gg, bg, x, 'elu', stats_pivot=coef1[2])
        st = {}

        def old_fwd():
            y1, p2, c1 = K.conv2d(x, w1, g1, bias=b1, in_act='elu', out_scale=m1, in_bn=(xparts, coef1[2], bn1), stats_pivot=bn2.running_mean)
            y2, _, c2 = K.conv2d(y1, w2, g2, bias=b2, in_act='elu', out_scale=m2, in_bn=(p2, bn2.running_mean, bn2))
            ab, out, op = K.conv1x1_gate(y2, wg, gg, bg, x, 'elu', stats_pivot=c1[2])
            st.update(y1=y1, y2=y2, ab=ab, c1=c1, c2=c2)

        def new_fwd():
            y1, p2, c1 = K.rb_conv(x, w1, g1, b1, 'elu', m1, in_bn=(xparts, coef1[2], bn1), stats_pivot=bn2.running_mean)
            y2, ab, out, op, c2 = K.rb_conv_gate(y1, w2, g2, b2, 'elu', m2, wg, gg, bg, x, 'elu', in_bn=(p2, bn2.running_mean, bn2), stats_pivot=c1[2])
            st.update(y1=y1, y2=y2, ab=ab, c1=c1, c2=c2)

        def new_fwd_convs_only():
            y1, p2, c1 = K.rb_conv(x, w1, g1, b1, 'elu', m1, in_bn=(xparts, coef1[2], bn1), stats_pivot=bn2.running_mean)
            y2, _, c2 = K.rb_conv(y1, w2, g2, b2, 'elu', m2, in_bn=(p2, bn2.running_mean, bn2))

        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)

        def old_bwd():
            y1, ab, c1, c2 = st['y1'], st['ab'], st['c1'], st['c2']
            dab, dy2 = K.conv1x1_gate_bwd(dout, ab, wg, gg, 'elu', out_scale=m2)
            dh2, p2 = K.conv2d_dgrad(dy2, w2, g2, (H, H), bn_bwd=(y1, c2[0], 'elu'))
            dy1 = K.affine_act_bwd_parts(p2, dh2, y1, c2[0], c2[1], 'elu', c2[2], c2[3], dg, db, drop=m1)
            dh1, p1 = K.conv2d_dgrad(dy1, w1, g1, (H, H), bn_bwd=(x, c1[0], 'elu'))
            K.affine_act_bwd_parts(p1, dh1, x, c1[0], c1[1], 'elu', c1[2], c1[3], dg, db, add=dout)

        def new_bwd():
            y1, ab, c1, c2 = st['y1'], st['ab'], st['c1'], st['c2']
            dab, dy2, dh2, p2 = K.rb_gate_dgrad(dout, ab, wg, gg, 'elu', m2, w2, g2, bn_bwd=(y1, c2[0], 'elu'))
            dy1, dh1, p1 = K.rb_apply_dgrad(p2, dh2, y1, c2[0], 'elu', dg, db, m1, w1, g1, bn_bwd=(x, c1[0], 'elu'))
            K.affine_act_bwd_parts(p1, dh1, x, c1[0], c1[1], 'elu', c1[2], c1[3], dg, db, add=dout)

        old_fwd(); new_fwd(); new_bwd(); old_bwd()
        K.prepared.prepare_all()   # the pre-split / pre-transformed weights of every descriptor met above (a training step does this once per step)
        t_of, t_nf, t_nc = timeit(old_fwd), timeit(new_fwd), timeit(new_fwd_convs_only)
        old_fwd()
        t_ob, t_nb = timeit(old_bwd), timeit(new_bwd)
        print('%dx%d B%d: forward old %6.1f us (3 launches) -> fused %6.1f us (2)  [two plain fused convs %6.1f] | backward old %6.1f us (5) -> fused %6.1f us (3)' %
              (H, H, B, t_of, t_nf, t_nc, t_ob, t_nb), flush=True)


if __name__ == '__main__':
    main()
