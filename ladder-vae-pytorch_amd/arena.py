"""Flat parameter / gradient arenas.

All parameters of a model live in ONE float32 buffer and all gradients in another of the same layout, so that
the optimiser, the L2 norm and the data-parallel all-reduce are single passes over contiguous memory instead of
1,689 per-tensor launches (experiment_manager.py:346-350, torch.optim.Adamax). Each nn.Parameter keeps the
reference's logical shape (state_dict compatible) but is re-bound as a strided view of the arena; convolution
weights are stored physically as [KH][KW][Cin][Cout] — the layout the implicit-GEMM kernels read with 16-byte
loads in forward (Cout contiguous) and dgrad (same memory, k = Cout contiguous).

Trainable parameters come first ([0, n_train)), frozen ones (e.g. a non-learned top prior) after, so Adamax and the
all-reduce work on the prefix and the L2 norm on everything.

Within the trainable prefix the parameters are laid out by GRADIENT-COMPLETION ORDER (`segment_of`: the model's
`grad_segments()`, i.e. reverse execution order: likelihood head, final top-down blocks, top-down layers 0..L-1, bottom-up
layers L-1..0, stem) — not by registration order (models/lvae.py:74,87-88,156,159-167 of the reference registers bottom-up and
top-down layers interleaved). A data-parallel bucket is then a contiguous slice of the gradient arena that is complete as soon
as backward has left its last segment (`segments` = [(segment id, lo, hi)] in that order), so its all-reduce can start while backward
continues (dist.GradAllReduce). The state_dict order is untouched (it follows module registration).
"""
import torch

from .lib.nn import Conv2dParams

ALIGN = 64  # floats; keeps every slot 256-byte aligned


def _round(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


class ParamArena:
    def __init__(self, model, device, segment_of=None):
        conv_w = {}
        for mod in model.modules():
            if isinstance(mod, Conv2dParams):
                conv_w[id(mod.weight)] = mod
        named = [(k, p) for k, p in model.named_parameters()]
        train = [(k, p) for k, p in named if p.requires_grad]
        frozen = [(k, p) for k, p in named if not p.requires_grad]
        seg_ids = [0] * len(train)
        if segment_of is not None:
            order = sorted(range(len(train)), key=lambda i: segment_of(train[i][0]))   # stable: registration order inside a segment
            train = [train[i] for i in order]
            seg_ids = [segment_of(k) for k, _ in train]
        self.names = [k for k, _ in train + frozen]
        sizes = [_round(p.numel()) for _, p in train + frozen]
        self.n_train = sum(_round(p.numel()) for _, p in train)
        self.n_total = sum(sizes)
        self.params = torch.zeros(self.n_total, dtype=torch.float32, device=device)
        self.grads = torch.zeros(self.n_train, dtype=torch.float32, device=device)
        self.slots = {}
        # [(segment id, lo, hi)]: element ranges of the gradient arena in completion order; the id is what the model's segment
        # markers report (a segment that owns no trainable parameter has no entry here, so ids may have gaps)
        self.segments = []
        lo = 0
        for i, sz in enumerate(sizes[:len(train)]):
            if i + 1 == len(train) or seg_ids[i + 1] != seg_ids[i]:
                hi = sum(sizes[:i + 1])
                self.segments.append((seg_ids[i], lo, hi))
                lo = hi
        off = 0
        for (k, p), sz in zip(train + frozen, sizes):
            n = p.numel()
            mod = conv_w.get(id(p))
            with torch.no_grad():
                src = p.detach().to(device=device, dtype=torch.float32)
                if mod is not None:
                    # logical (Cout,Cin,KH,KW) [or (Cin,Cout,KH,KW) transposed] -> physical [KH][KW][Cin][Cout]
                    perm = (2, 3, 0, 1) if mod.transposed else (2, 3, 1, 0)
                    inv = (2, 3, 0, 1) if mod.transposed else (3, 2, 0, 1)
                    phys_shape = tuple(src.shape[i] for i in perm)
                    view = self.params[off:off + n].view(phys_shape).permute(inv)
                    view.copy_(src)
                    gview = self.grads[off:off + n].view(phys_shape).permute(inv) if p.requires_grad else None
                elif src.dim() == 4:
                    # feature-map shaped parameter (top_prior_params): physical NHWC, logical NCHW
                    phys_shape = (src.shape[0], src.shape[2], src.shape[3], src.shape[1])
                    view = self.params[off:off + n].view(phys_shape).permute(0, 3, 1, 2)
                    view.copy_(src)
                    gview = self.grads[off:off + n].view(phys_shape).permute(0, 3, 1, 2) if p.requires_grad else None
                else:
                    view = self.params[off:off + n].view(src.shape)
                    view.copy_(src)
                    gview = self.grads[off:off + n].view(src.shape) if p.requires_grad else None
            p.data = view
            p.grad = gview
            self.slots[k] = (off, n)
            off += sz

    def zero_grad(self):
        from . import kernels as K
        if self.grads.is_cuda:
            K.fill(self.grads, 0.0)     # our own kernel: no torch fill inside the (captured) step
        else:
            self.grads.zero_()           # CPU arenas exist only in the host-logic tests

    def owns(self, p):
        a = self.params
        return a.data_ptr() <= p.data_ptr() < a.data_ptr() + 4 * a.numel()
