"""Cost of a dependent kernel boundary inside a replayed hipGraph on this box: chains of trivial launches of this library, HIP-event
timed per launch. Profiling helper (answers: what is the floor of one more launch in the captured training step?)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvae_amd  # noqa: F401
from lvae_amd import kernels as K


def timeit(fn, n=200, reps=10):
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


a4 = torch.zeros(4, device='cuda')
print('fill, 1 workgroup of 4 floats:        %.2f us per launch' % timeit(lambda: K.fill(a4, 1.0)))
for n in (1 << 10, 1 << 16, 1 << 20, 1 << 22):
    a, b, o = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    import ctypes as C
    from lvae_amd._C import call, ptr, stream_ptr
    print('add of %8d floats (%5.1f MB moved): %.2f us per launch' % (n, 12 * n / 1e6, timeit(lambda: call('lvae_add_f32', ptr(a), ptr(b), n, ptr(o), stream_ptr()))))
x = torch.randn(256, 4, 4, 64, device='cuda')
sc = torch.ones(64, device='cuda')
print('affine_act on 256x4x4x64:             %.2f us per launch' % timeit(lambda: K.affine_act(x, sc, sc, 'elu')))
