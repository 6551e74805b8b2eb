"""Do independent small kernels on several streams overlap inside a hipGraph? (profiling helper)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvae_amd  # noqa: F401
from lvae_amd import kernels as K
from conv_bench import packed


def run(nstreams, H, reps=64):
    B, C = 256, 64
    xs = [torch.randn(B, H, H, C, device='cuda') for _ in range(4)]
    dys = [torch.randn(B, H, H, C, device='cuda') for _ in range(4)]
    ws = [packed(C, C, 3) for _ in range(reps)]
    dws = [torch.zeros_like(w) for w in ws]
    g = K.ConvGeom(ws[0], 1, 1)
    streams = [torch.cuda.Stream() for _ in range(nstreams)]

    def body():
        cur = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(cur)
        for i in range(reps):
            st = streams[i % nstreams]
            with torch.cuda.stream(st):
                K.conv2d_wgrad(xs[i % 4], dys[i % 4], ws[i], g, dws[i], None)
        for st in streams:
            cur.wait_stream(st)

    body()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        body()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 / reps * 1e3


if __name__ == '__main__':
    for H in (2, 4, 8):
        print('H=%d: ' % H + '  '.join('%d streams %.1f us/wgrad' % (n, run(n, H)) for n in (1, 2, 4, 8)))
