"""Debug: persistent gate forward, statistics vs its own output (write-through epilogue)."""
import importlib, math, sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
pkg = importlib.import_module('ladder-vae-pytorch_amd')
K = importlib.import_module('ladder-vae-pytorch_amd.kernels')
g = torch.Generator().manual_seed(45)
N, C, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 40, 64, 16, 16
x = torch.randn(N, H, W, C, generator=g).cuda()
res = (torch.randn(N, H, W, C, generator=g) + 1.5).cuda()
wl = torch.randn(2 * C, C, 1, 1, generator=g) / math.sqrt(C)
w = wl.permute(2, 3, 1, 0).contiguous().cuda().permute(3, 2, 0, 1)   # arena layout: physical [KH][KW][Cin][Cout], logical torch shape
b = torch.randn(2 * C, generator=g).cuda()
pivot = torch.randn(C, generator=g).cuda()
geom = K.ConvGeom(w, 1, 0)
for it in range(3):
    ab, out, parts = K.conv1x1_gate(x, w, geom, b, res, 'elu', stats_pivot=pivot)
    torch.cuda.synchronize()
    ab_ref = x.reshape(-1, C).double() @ wl[:, :, 0, 0].t().double().cuda() + b.double()
    out_ref = (F.elu(ab_ref[:, :C]) * torch.sigmoid(ab_ref[:, C:]) + res.reshape(-1, C).double()).float()
    print('it', it, 'rows', parts.rows, 'ab err', (ab.reshape(-1, 2 * C) - ab_ref.float()).abs().max().item(),
          'out err', (out.reshape(-1, C) - out_ref).abs().max().item())
    pr = parts.buf[:parts.rows]
    s1 = pr[:, 0].double().sum(0)
    s2 = pr[:, 1].double().sum(0)
    d = out.reshape(-1, C).double() - pivot.double()
    e1 = (s1 - d.sum(0)).abs()
    e2 = (s2 - (d * d).sum(0)).abs()
    print('   s1 err max', e1.max().item(), 'bad ch', (e1 > 1e-2).nonzero().flatten().tolist())
    print('   s2 err max', e2.max().item(), 'bad ch', (e2 > 1e-1).nonzero().flatten().tolist())
    # per-row check: each workgroup's tile
    tile = out.reshape(-1, 64, C).double() - pivot.double()
    ts = tile.sum(1)
    if parts.rows < ts.shape[0]:   # persistent: workgroup b owns tiles b, b + rows, ...
        extra = ts[parts.rows:]
        ts = ts[:parts.rows].clone()
        ts[:extra.shape[0]] += extra
    if True:
        r1 = (pr[:, 0].double() - ts).abs()
        bad = (r1 > 1e-3).nonzero()
        print('   bad (row, ch) count', bad.shape[0], bad[:10].tolist())
