"""Tensor-level wrappers over the C ABI (one Python function per kernel entry point).

Activations are 4-D torch tensors stored NHWC: shape (N, H, W, C), contiguous, float32, on the GPU. No arithmetic
happens in Python or in torch ops here: every function launches hand-written HIP kernels on torch's current
stream. Buffers (outputs, workspaces) come from torch's caching allocator, which is graph-capture safe.
"""
import ctypes as C
import weakref

import torch

from . import _C
from ._C import ACT, BnFold, ConvDesc, RbExt, GATHER_CONV, GATHER_TRANSPOSED, PREC_BF16, PREC_F32, call, ptr, ptr_dt, stream_ptr

_ws_cache = {}

# Arithmetic of the matrix-core kernels for every convolution descriptor built from here on: PREC_F32 (results as an fp32
# multiply-add chain) or PREC_BF16 (operands rounded to bf16 at the MFMA input, fp32 accumulate). LadderVAE sets it from its
# `compute_dtype` at the start of every pass; backward launches of that pass follow before the next forward.
precision = PREC_F32


# Arithmetic-form request (lvae_conv_desc.form) of every descriptor built from here on: FORM_AUTO in the product; the parity tests
# walk the other forms (`with kernels.use_form(_C.FORM_F32_MFMA): ...`).
form = _C.FORM_AUTO


class use_form:
    def __init__(self, f):
        self.f = f

    def __enter__(self):
        global form
        self.prev, form = form, self.f

    def __exit__(self, *exc):
        global form
        form = self.prev


BF16_MODE_NOTE = ('forward, dgrad and weight gradient of the 3x3 convolutions of the 8x8..32x32 levels and the GateLayer2d 1x1 family on '
                  'v_mfma_f32_32x32x16_bf16 with bf16 operands; the tensors inside those residual blocks (conv outputs, gate pre-activations '
                  'and their gradients) stored as bf16 in HBM, as torch.autocast(bfloat16) stores conv outputs; residual stream, fp32 '
                  'accumulate / statistics / KL / likelihood / parameters / Adamax; merge 1x1 and <=4x4 convolutions fp32')


def set_precision(dtype):
    global precision
    precision = {'f32': PREC_F32, 'fp32': PREC_F32, 'float32': PREC_F32, 'bf16': PREC_BF16, 'bfloat16': PREC_BF16}[str(dtype).replace('torch.', '')]


def workspace(nbytes, device):
    """A per-device scratch buffer, grown on demand. Kernels run in stream order, so one buffer is enough."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)  # one scratch buffer per launch stream
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes or buf.device != device:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _chk_nhwc(t, name):
    if t.dim() != 4 or not t.is_contiguous() or t.dtype not in (torch.float32, torch.bfloat16):
        raise _C.LvaeHipError("%s must be a contiguous float32 (or, inside a residual block under compute_dtype bf16, bfloat16) (N,H,W,C) "
                              "tensor, got shape %s strides %s %s" % (name, tuple(t.shape), tuple(t.stride()), t.dtype))


def _dt(t):
    """lvae_conv_desc.*_dtype of a tensor (LVAE_DT_F32 for None)."""
    return _C.DT_BF16 if (t is not None and t.dtype == torch.bfloat16) else _C.DT_F32


class ConvGeom:
    """Geometry of one convolution call site plus the strides of its weight tensor.

    `weight` is the nn.Parameter in the reference's logical shape — (Cout,Cin,KH,KW) for nn.Conv2d,
    (Cin,Cout,KH,KW) for nn.ConvTranspose2d — in ANY tap-linear memory layout; the arena packs it as
    [KH][KW][Cin][Cout] so that forward reads are n-contiguous and dgrad reads k-contiguous float4s.
    """

    def __init__(self, weight, stride=1, pad=0, transposed=False, output_padding=0):
        self.transposed = bool(transposed)
        if transposed:
            self.Cin, self.Cout, self.KH, self.KW = weight.shape
            s_ci, s_co, s_kh, s_kw = weight.stride()
        else:
            self.Cout, self.Cin, self.KH, self.KW = weight.shape
            s_co, s_ci, s_kh, s_kw = weight.stride()
        if self.KH * self.KW > 1 and s_kh != self.KW * s_kw:
            raise _C.LvaeHipError("weight layout is not tap-linear (strides %s)" % (tuple(weight.stride()),))
        self.s_tap, self.s_ci, self.s_co = (s_kw if self.KH * self.KW > 1 else 0), s_ci, s_co
        self.stride, self.pad, self.output_padding = int(stride), int(pad), int(output_padding)

    def out_size(self, H, W):
        if self.transposed:
            return ((H - 1) * self.stride - 2 * self.pad + self.KH + self.output_padding,
                    (W - 1) * self.stride - 2 * self.pad + self.KW + self.output_padding)
        return ((H + 2 * self.pad - self.KH) // self.stride + 1, (W + 2 * self.pad - self.KW) // self.stride + 1)


def _desc(g, weight, x, x2, N, H, W, OH, OW, Cout, k_stride, n_stride, gather, bias=None, in_scale=None, in_shift=None,
          in_act=None, out_scale=None, out_act=None, y=None):
    d = ConvDesc()
    d.x, d.x2 = ptr_dt(x), ptr_dt(x2)   # element types travel in d.x_dtype / d.y_dtype below
    d.C1, d.C2 = x.shape[3], (x2.shape[3] if x2 is not None else 0)
    d.w, d.w_stap, d.w_sk, d.w_sn = ptr(weight), g.s_tap, k_stride, n_stride
    d.bias, d.in_scale, d.in_shift = ptr(bias), ptr(in_scale), ptr(in_shift)
    d.in_act, d.out_scale, d.out_act = ACT[in_act], ptr(out_scale), ACT[out_act]
    d.y = ptr_dt(y)
    d.N, d.H, d.W, d.OH, d.OW, d.Cout = N, H, W, OH, OW, Cout
    d.KH, d.KW, d.stride, d.pad, d.gather = g.KH, g.KW, g.stride, g.pad, gather
    d.precision = precision
    d.form = form
    d.x_dtype, d.y_dtype = _dt(x), _dt(y)
    if x2 is not None and x2.dtype != x.dtype:
        raise _C.LvaeHipError("x and x2 must have the same element type")
    return d


class _PreparedWeights:
    """Transformed-weight cache of the Winograd convolutions (lvae_conv2d_prepare_weights).

    Every eligible (weight view, orientation) gets a private scratch buffer; `prepare_all()` refreshes all of them in ONE
    launch (the training step calls it right after the optimizer has written the weights) and a convolution whose entry is
    current runs with workspace_ready = 1. An entry is current while neither the tensor's autograd version counter (any
    torch in-place op: load_state_dict, copy_, ...) nor `epoch` (bumped by every kernel of this package that writes weights
    through raw pointers: Adamax, broadcast, graph replays) has moved since it was written; otherwise the convolution falls
    back to transforming its weights itself, so a stale buffer is never read.
    """

    def __init__(self):
        self.entries = {}   # key -> dict(weight, U, desc bytes, stamp)
        self.epoch = 0
        self.table = None   # device table of the entries, rebuilt when the set changes
        self.enabled = True

    def stamp(self, weight):
        return (weight._version, self.epoch)

    @staticmethod
    def _alive(ent):
        """The entry's weight tensor still exists and still lives where the entry's raw pointers say (a model that was dropped, or
        re-packed by .to(), leaves entries behind that must neither pin its arena nor be transformed again)."""
        w = ent['weight']()
        return w is not None and w.data_ptr() == ent['base']

    def evict_dead(self):
        dead = [k for k, e in self.entries.items() if not self._alive(e)]
        for k in dead:
            del self.entries[k]
        if dead:
            self.table = None
        return len(dead)

    def attach(self, d, weight, device, need, entry_fn='lvae_conv2d_prepare_entry'):
        key = (d.w, d.w_sk, d.w_sn, d.gather, d.Cout, int(need), device.index, entry_fn)   # `need` / entry_fn tell the transform kinds apart
        ent = self.entries.get(key)
        if ent is not None and not self._alive(ent):   # the address was recycled for another tensor
            del self.entries[key]
            self.table = None
            ent = None
        if ent is None:
            ent = {'weight': weakref.ref(weight), 'base': weight.data_ptr(), 'U': torch.empty(int(need), dtype=torch.uint8, device=device), 'stamp': None, 'entry': None}
            d.workspace, d.workspace_bytes = ent['U'].data_ptr(), ent['U'].numel()
            raw = (C.c_char * _C.load().lvae_conv2d_prepare_entry_bytes())()
            call(entry_fn, C.byref(d), C.cast(raw, C.c_void_p))
            ent['entry'] = bytes(raw)
            ent['cout'] = max(d.Cout, d.C1) if entry_fn == 'lvae_resblock_gate_prepare_entry' else d.Cout   # threads per entry of the batched launch
            self.entries[key] = ent
            self.table = None
        d.workspace, d.workspace_bytes = ent['U'].data_ptr(), ent['U'].numel()
        d._ws_tensor = ent['U']   # (a Python attribute of the ctypes object: lets a caller keep the scratch buffer alive, rb_weight_ranges)
        d.workspace_ready = 1 if (self.enabled and ent['stamp'] == self.stamp(weight)) else 0

    def prepare_all(self):
        """One launch that (re)writes every registered buffer from the current weights. Returns the number of entries."""
        if not torch.cuda.is_current_stream_capturing():
            self.evict_dead()   # (a captured step keeps its table: the replayed launch reads the pointers it was captured with)
        if not self.entries or not self.enabled:
            return 0
        by_dev = {}
        for k, e in self.entries.items():
            by_dev.setdefault(k[6], []).append(e)
        if self.table is None:
            if torch.cuda.is_current_stream_capturing():
                return 0  # the table upload is a host copy: convolutions transform their own weights until an eager step built it
            self.table = {}
            for dev, ents in by_dev.items():
                blob = b''.join(e['entry'] for e in ents)
                t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(torch.device('cuda', dev))
                self.table[dev] = (t, len(ents), max(e['cout'] for e in ents))
        for dev, ents in by_dev.items():
            t, n, max_cout = self.table[dev]
            with torch.cuda.device(dev):
                call('lvae_conv2d_prepare_weights', t.data_ptr(), n, max_cout, stream_ptr())
            for e in ents:
                w = e['weight']()
                e['stamp'] = self.stamp(w) if w is not None else None
        return len(self.entries)

    def pin_current(self):
        """Called by a TrainStep right after it captured a graph containing prepare_all(): returns a handle that keeps the table the
        captured launch reads, and every scratch buffer its entries point to, alive. A later change of the entry set (another batch
        size, another model, evict_dead) builds a NEW table; the pinned one stays valid for the replays of that graph."""
        if self.table is None:
            return None
        return (self.table, [e for e in self.entries.values()])   # the caller (the graph's owner) holds it; nothing here does

    def invalidate(self):
        """Forget that any buffer is current (a step that stamped them was abandoned before its launches ran, e.g. a failed capture):
        every convolution transforms its own weights until the next prepare_all()."""
        for e in self.entries.values():
            e['stamp'] = None

    def weights_written(self):
        """Call after weights were modified through raw pointers (optimizer kernel, collective, graph replay)."""
        self.epoch += 1


prepared = _PreparedWeights()


def _conv_ws(d, weight, device):
    need = _C.load().lvae_conv2d_workspace(C.byref(d))
    if need:
        prepared.attach(d, weight, device, need)


def resblock_bf16_storage(x, weight, g):
    """True when a residual block whose 3x3 convolutions are (weight, g) on input x can keep its internal tensors (conv outputs, gate
    pre-activations and their gradients) in bf16: precision bf16 and every kernel of the block's forward and backward has the
    bf16-storage form for this shape (lvae_resblock_bf16_storage)."""
    if precision != PREC_BF16 or _ddi is not None or x.dtype != torch.float32:
        return False
    N, H, W, _ = x.shape
    d = _desc(g, weight, x, None, N, H, W, H, W, g.Cout, g.s_ci, g.s_co, GATHER_CONV)
    _conv_ws(d, weight, x.device)
    return bool(_C.load().lvae_resblock_bf16_storage(C.byref(d)))


_ddi = None  # set by init.data_dependent_init for the duration of its forward pass


def _ddi_conv(state, x, weight, g, bias, x2, in_scale, in_shift, in_act, out_scale, out_act):
    """One convolution of the data-dependent initialisation pass (init.py): plain output, per-channel statistics, parameter
    rescale in place, corrected output, then the caller's epilogue."""
    global _ddi
    _ddi = None
    try:
        y = conv2d(x, weight, g, bias=bias, x2=x2, in_scale=in_scale, in_shift=in_shift, in_act=in_act)
        _, _, mean, rstd = bn_stats(y, None, None, None, None, eps=1e-20)
    finally:
        _ddi = state
    scale = 1.0 / (1.0 / rstd + 1e-5)
    weight.mul_(scale.view(1, -1, 1, 1) if g.transposed else scale.view(-1, 1, 1, 1))
    if bias is not None:
        bias.sub_(mean).mul_(scale)
        y = affine_act(y, scale, -mean * scale, None)
    else:
        y = affine_act(y, scale, torch.zeros_like(scale), None)
    if out_scale is not None or out_act is not None:
        ones, zeros = torch.ones_like(scale), torch.zeros_like(scale)
        y = affine_act(y, ones, zeros, out_act, row_scale=out_scale)
    state['done'].add(weight.data_ptr())
    state['count'] += 1
    return y


class StatParts:
    """BatchNorm partial sums a kernel epilogue wrote for its OUTPUT: buf (rows [+1], 2, C) — (sum(y - pivot), sum((y - pivot)^2)) per
    workgroup row; has_pivot: the producer stored its pivot in row `rows` (such a buffer can be finalized inside the consuming
    convolution's prologue, see lvae_bn_fold)."""
    __slots__ = ('buf', 'rows', 'has_pivot')

    def __init__(self, buf, rows, has_pivot):
        self.buf, self.rows, self.has_pivot = buf, rows, has_pivot

    def rows_view(self):
        return self.buf[:self.rows]


def conv2d(x, weight, g, bias=None, x2=None, in_scale=None, in_shift=None, in_act=None, out_scale=None, out_act=None,
           stats_pivot=None, in_bn=None, out_bf16=False):
    """y = out_act((conv(in_act(x*in_scale+in_shift)) + bias) * out_scale). x (and x2) NHWC; returns NHWC.
    stats_pivot (Cout,): also ask the kernel's epilogue for BatchNorm partials of y around that pivot; returns (y, parts) with
    parts a StatParts, or (y, None) when the kernel variant chosen for this shape has no such epilogue.
    in_bn = (parts: StatParts, pivot, bn): training-mode BatchNorm of the INPUT whose statistics exist as partial sums (bn has
    weight, bias, running_mean, running_var, eps, momentum); the coefficients are finalized inside the convolution when the
    selected kernel can do that (lvae_conv2d_folds_bn_finalize) and by lvae_bn_finalize_parts_f32 otherwise. Returns
    (y, parts | None, (scale, shift, mean, rstd)).
    out_bf16: store y as bfloat16 (only inside a residual block whose shape passed resblock_bf16_storage)."""
    if _ddi is not None and weight.data_ptr() not in _ddi['done']:
        assert in_bn is None
        y = _ddi_conv(_ddi, x, weight, g, bias, x2, in_scale, in_shift, in_act, out_scale, out_act)
        return y if stats_pivot is None else (y, None)
    _chk_nhwc(x, 'x')
    N, H, W, C1 = x.shape
    if x2 is not None:
        _chk_nhwc(x2, 'x2')
        assert x2.shape[:3] == x.shape[:3]
    if C1 + (x2.shape[3] if x2 is not None else 0) != g.Cin:
        raise _C.LvaeHipError("conv2d: input has %d channels, weight expects %d" % (C1 + (x2.shape[3] if x2 is not None else 0), g.Cin))
    OH, OW = g.out_size(H, W)
    y = torch.empty((N, OH, OW, g.Cout), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    d = _desc(g, weight, x, x2, N, H, W, OH, OW, g.Cout, g.s_ci, g.s_co,
              GATHER_TRANSPOSED if g.transposed else GATHER_CONV, bias, in_scale, in_shift, in_act, out_scale, out_act, y)
    _conv_ws(d, weight, x.device)
    lib = _C.load()
    coef = fold = None
    if in_bn is not None:
        sp, pivot, bn = in_bn
        M = N * H * W
        if sp.has_pivot and x2 is None and lib.lvae_conv2d_folds_bn_finalize(C.byref(d)):
            coef = torch.empty((4, C1), dtype=torch.float32, device=x.device)
            fold = BnFold(ptr(sp.buf), sp.rows, M, ptr(bn.weight), ptr(bn.bias), bn.eps, bn.momentum, ptr(bn.running_mean),
                          ptr(bn.running_var), ptr(coef))
            d.in_fold = C.addressof(fold)
            coef = (coef[0], coef[1], coef[2], coef[3])
        else:
            coef = bn_finalize_parts(sp.rows_view(), M, pivot, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum)
            d.in_scale, d.in_shift = ptr(coef[0]), ptr(coef[1])
    parts = None
    if stats_pivot is not None:
        rows = lib.lvae_conv2d_stats_rows(C.byref(d))
        if rows > 0:
            buf_rows = lib.lvae_conv2d_stats_buffer_rows(C.byref(d))   # + 1 where the kernel stores its pivot behind the partial rows
            buf = torch.empty((buf_rows, 2, g.Cout), dtype=torch.float32, device=x.device)
            parts = StatParts(buf, rows, buf_rows > rows)
            d.stats_out, d.stats_pivot = ptr(buf), ptr(stats_pivot)
    call('lvae_conv2d_f32', C.byref(d), stream_ptr())
    del fold
    if in_bn is not None:
        return y, parts, coef
    return y if stats_pivot is None else (y, parts)


def conv1x1_gate(x, weight, g, bias, res, act, need_ab=True, stats_pivot=None):
    """GateLayer2d forward: ab = conv1x1(x) + bias (returned when need_ab), out = act(a) * sigmoid(b) + res, one kernel.
    Falls back to conv2d + gate_fwd when the fused kernel does not support the shape.
    stats_pivot (C,): also return BatchNorm partials of `out` (rows, 2, C) around that pivot, or None when unsupported:
    returns (ab, out, parts)."""
    _chk_nhwc(x, 'x')
    N, H, W, _ = x.shape
    Cn = g.Cout // 2
    fused_ok = (g.KH == 1 and g.KW == 1 and g.stride == 1 and g.pad == 0 and not g.transposed and g.Cin <= 128 and
                g.Cout <= 128 and g.Cin % 4 == 0 and g.Cout % 8 == 0 and (g.s_co == 1 or g.s_ci == 1))
    if not fused_ok or (_ddi is not None and weight.data_ptr() not in _ddi['done']):
        ab = conv2d(x, weight, g, bias=bias)
        out = gate_fwd(ab, res, act)
        return (ab, out) if stats_pivot is None else (ab, out, None)
    ab = torch.empty((N, H, W, g.Cout), dtype=x.dtype, device=x.device) if need_ab else None   # bf16-stored x: bf16-stored ab
    out = torch.empty((N, H, W, Cn), dtype=torch.float32, device=x.device)
    d = _desc(g, weight, x, None, N, H, W, H, W, g.Cout, g.s_ci, g.s_co, GATHER_CONV, bias, y=ab)
    parts = None
    if stats_pivot is not None:
        rows = _C.load().lvae_conv1x1_gate_stats_rows(C.byref(d))
        if rows > 0:
            buf = torch.empty((rows + 1, 2, Cn), dtype=torch.float32, device=x.device)   # last row: the pivot (written by the kernel)
            parts = StatParts(buf, rows, True)
            d.stats_out, d.stats_pivot = ptr(buf), ptr(stats_pivot)
    call('lvae_conv1x1_gate_f32', C.byref(d), ptr(res), ACT[act], ptr(out), stream_ptr())
    return (ab, out) if stats_pivot is None else (ab, out, parts)


def conv1x1_gate_bwd(dout, ab, weight, g, act, out_scale=None):
    """Backward of conv1x1_gate w.r.t. the convolution input and the pre-activations: returns (dab, dx). One kernel when the
    shape is supported (gate backward formed in the dgrad kernel's operand staging), else gate_bwd + conv2d_dgrad."""
    _chk_nhwc(dout, 'dout')
    N, H, W, Cn = dout.shape
    fused_ok = (g.KH == 1 and g.KW == 1 and g.stride == 1 and g.pad == 0 and not g.transposed and g.Cout == 2 * Cn and
                g.Cout <= 128 and g.Cin <= 128 and g.Cout % 8 == 0 and g.Cin % 4 == 0 and (g.s_co == 1 or g.s_ci == 1))
    if not fused_ok:
        dab = gate_bwd(dout, ab, act)
        return dab, conv2d_dgrad(dab, weight, g, (H, W), out_scale=out_scale)
    dab = torch.empty_like(ab)
    dx = torch.empty((N, H, W, g.Cin), dtype=torch.float32, device=dout.device)
    d = _desc(g, weight, ab, None, N, H, W, H, W, g.Cin, g.s_co, g.s_ci, GATHER_TRANSPOSED, out_scale=out_scale, y=dx)
    call('lvae_conv1x1_gate_bwd_f32', C.byref(d), ptr(dout), ptr(ab), ACT[act], ptr(dab), stream_ptr())
    return dab, dx


def gate_bwd_fused_ok(x_like, weight, g, dweight=None):
    """True when conv1x1_gate_bwd_wgrad takes the gate convolution (weight, g) on tensors shaped like x_like (N,H,W,C) — and can therefore
    also form its dout from a deferred BatchNorm-backward apply (PendingApply)."""
    N, H, W, Cn = x_like.shape
    if not (g.KH == 1 and g.KW == 1 and g.stride == 1 and g.pad == 0 and not g.transposed and g.Cout == 2 * Cn and g.s_co == 1):
        return False
    if dweight is not None and tuple(dweight.stride()) != tuple(weight.stride()):
        return False
    if form == _C.FORM_F32_MFMA and precision != PREC_BF16:
        return False
    d = _desc(g, weight, x_like, None, N, H, W, H, W, g.Cin, g.s_co, g.s_ci, GATHER_TRANSPOSED)
    d.C1 = g.Cout
    return bool(_C.load().lvae_conv1x1_gate_bwd_wgrad_workspace(C.byref(d)))


def conv1x1_gate_bwd_wgrad(dout, ab, y, weight, g, act, dweight, dbias, out_scale=None, out_bf16=False, apply=None):
    """conv1x1_gate_bwd and the weight / bias gradient of the gate convolution in one persistent kernel (dab never leaves LDS):
    returns dx, and accumulates into dweight / dbias. Returns None when the shape is not supported (caller composes the two).
    apply (PendingApply): dout does not exist yet; the kernel forms it from the deferred BatchNorm-backward apply and stores it to apply.out."""
    _chk_nhwc(dout, 'dout')
    N, H, W, Cn = dout.shape
    if not (g.KH == 1 and g.KW == 1 and g.stride == 1 and g.pad == 0 and not g.transposed and g.Cout == 2 * Cn and g.s_co == 1):
        return None
    if tuple(dweight.stride()) != tuple(weight.stride()):
        return None
    if ab.dtype != y.dtype:
        raise _C.LvaeHipError("conv1x1_gate_bwd_wgrad: ab and y must have the same element type")
    dx = torch.empty((N, H, W, g.Cin), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=dout.device)
    d = _desc(g, weight, ab, None, N, H, W, H, W, g.Cin, g.s_co, g.s_ci, GATHER_TRANSPOSED, out_scale=out_scale, y=dx)
    need = _C.load().lvae_conv1x1_gate_bwd_wgrad_workspace(C.byref(d))
    if not need or (apply is not None and form == _C.FORM_F32_MFMA and precision != PREC_BF16):
        return None
    ws = workspace(need, dout.device)
    ap = None
    if apply is not None:
        ap = _C.BnApply(ptr(apply.parts), apply.parts.shape[0], ACT[apply.act], N * H * W, ptr(apply.coef0), ptr_dt(apply.dh), ptr(apply.x),
                        ptr(apply.add), ptr(apply.dgamma), ptr(apply.dbeta), ptr(apply.out), int(apply.dh.dtype == torch.bfloat16), 0)
    call('lvae_conv1x1_gate_bwd_wgrad_f32', C.byref(d), ptr(dout), ptr_dt(ab), ptr_dt(y), ACT[act], ptr(dweight), g.s_ci, g.s_co, ptr(dbias),
         ws.data_ptr(), ws.numel(), C.byref(ap) if ap is not None else None, stream_ptr())
    return dx


# ----------------------------------------------------------------------------------------------------------------
# Fused residual-block launches of the low-resolution levels (csrc/resblock_img.hip, lvae_resblock_conv_f32)
def _rb_desc(x_like, weight, g, dgrad, y=None, bias=None, in_scale=None, in_shift=None, in_act=None, out_scale=None):
    """Descriptor of the block's 3x3 convolution (forward) or of its dgrad, with the pre-split weights attached."""
    N, H, W, _ = x_like.shape
    if dgrad:
        d = _desc(g, weight, x_like, None, N, H, W, H, W, g.Cin, g.s_co, g.s_ci, GATHER_TRANSPOSED, out_scale=out_scale, y=y)
    else:
        d = _desc(g, weight, x_like, None, N, H, W, H, W, g.Cout, g.s_ci, g.s_co, GATHER_CONV, bias, in_scale, in_shift, in_act, out_scale, None, y)
    need = _C.load().lvae_resblock_conv_workspace(C.byref(d))
    if need:
        prepared.attach(d, weight, x_like.device, need, 'lvae_resblock_conv_prepare_entry')
    return d, need


def _rb_gate_ws(e, gate_w, gate_g, device, bwd):
    """Attach the pre-split copy of the gate weight (direction: forward 64 -> 128, or its dgrad 128 -> 64) to the extension block."""
    gd = ConvDesc()
    gd.w, gd.precision = ptr(gate_w), precision
    if bwd:
        gd.C1, gd.Cout, gd.w_sk, gd.w_sn, gd.gather = gate_g.Cout, gate_g.Cin, gate_g.s_co, gate_g.s_ci, GATHER_TRANSPOSED
    else:
        gd.C1, gd.Cout, gd.w_sk, gd.w_sn, gd.gather = gate_g.Cin, gate_g.Cout, gate_g.s_ci, gate_g.s_co, GATHER_CONV
    need = _C.load().lvae_resblock_gate_workspace(C.byref(gd))
    if not need:
        raise _C.LvaeHipError("fused residual block: the gate convolution must be 1x1, 64 -> 128 channels")
    prepared.attach(gd, gate_w, device, need, 'lvae_resblock_gate_prepare_entry')
    e.gate_w, e.gate_w_sk, e.gate_w_sn = gd.w, gd.w_sk, gd.w_sn
    e.gate_ws, e.gate_ws_bytes, e.gate_ws_ready = gd.workspace, gd.workspace_bytes, gd.workspace_ready
    e._ws_tensor = gd._ws_tensor


def rb_rows(x, weight, g):
    """Workgroups (= statistics rows) of the fused residual-block kernels for the 3x3 convolution (weight, g) on x, 0 when the shape is
    not theirs (needs 3x3 / stride 1 / pad 1, 64 -> 64 channels, H*W a divisor of 64, fp32 NHWC x)."""
    if _ddi is not None or x.dtype != torch.float32 or x.dim() != 4 or g.transposed or g.KH != 3 or g.stride != 1 or g.pad != 1:
        return 0
    N, H, W, _ = x.shape   # (no scratch is attached here: asking must not register weights the caller may never run through these kernels)
    d = _desc(g, weight, x, None, N, H, W, H, W, g.Cout, g.s_ci, g.s_co, GATHER_CONV)
    return int(_C.load().lvae_resblock_conv_rows(C.byref(d)))


# Which residual blocks take the fused launches (measured per level at batch 256, tools/rb_bench.py: forward old -> fused 41 -> 31 us at
# 8x8, 33 -> 26 at 4x4, 22 -> 27 at 2x2, where the position-major kernels skip the taps outside the image; backward 60 -> 43, 46 -> 33,
# 37 -> 33): forward from 16 pixels per image up, backward everywhere the kernels exist. LVAE_RB_FWD_MIN_HW / LVAE_RB_BWD_MIN_HW
# (profiling only) move the thresholds; 0 pixels = never.
import os as _os
_RB_FWD_MIN_HW = int(_os.environ.get('LVAE_RB_FWD_MIN_HW', '16'))
_RB_BWD_MIN_HW = int(_os.environ.get('LVAE_RB_BWD_MIN_HW', '1'))
_RB_GATE_LARGE = _os.environ.get('LVAE_RB_GATE_LARGE', '1') != '0'   # conv2 + gate in one launch at the >= 16x16 levels (Winograd kernel)


def rb_policy(x, weight, g):
    """(fused forward?, fused backward?) for the residual block whose 3x3 convolutions look like (weight, g) on input x."""
    if rb_rows(x, weight, g) <= 0:
        return False, False
    hw = x.shape[1] * x.shape[2]
    return (_RB_FWD_MIN_HW > 0 and hw >= _RB_FWD_MIN_HW), (_RB_BWD_MIN_HW > 0 and hw >= _RB_BWD_MIN_HW)


def _rb_in_bn(d, in_bn, x, can_fold=True):
    """Training-mode BatchNorm of the convolution input from partial sums: folded into the prologue when the producer stored its pivot
    (and the kernel that will run can fold), by lvae_bn_finalize_parts_f32 otherwise. Returns (coef 4-tuple, keep-alive)."""
    sp, pivot, bn = in_bn
    N, H, W, C1 = x.shape
    M = N * H * W
    if sp.has_pivot and can_fold:
        coef = torch.empty((4, C1), dtype=torch.float32, device=x.device)
        fold = BnFold(ptr(sp.buf), sp.rows, M, ptr(bn.weight), ptr(bn.bias), bn.eps, bn.momentum, ptr(bn.running_mean),
                      ptr(bn.running_var), ptr(coef))
        d.in_fold = C.addressof(fold)
        return (coef[0], coef[1], coef[2], coef[3]), fold
    coef = bn_finalize_parts(sp.rows_view(), M, pivot, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum)
    d.in_scale, d.in_shift = ptr(coef[0]), ptr(coef[1])
    return coef, None


def rb_weight_ranges(x_like, weight, g, dgrad, gate=None, gate_bwd=False):
    """[(device pointer, bytes, buffer)] of the pre-split weights a later fused launch will stream: the 3x3 convolution (weight, g) in the
    given direction and optionally the gate (gate = (gate_w, gate_g)). For the `prefetch` argument of the launch that runs just before it.
    The third element is the scratch tensor itself: whoever remembers a range (ops' cross-block links, baked into captured graphs as raw
    addresses) thereby keeps the memory allocated, whatever happens to the transformed-weight cache in between (evict_dead, a cleared
    cache, a precision switch) — a stale range is then touched uselessly, never a freed one."""
    d, need = _rb_desc(x_like, weight, g, dgrad)
    out = [(int(d.workspace), int(need), d._ws_tensor)] if need else []
    if gate is not None:
        e = RbExt()
        _rb_gate_ws(e, gate[0], gate[1], x_like.device, gate_bwd)
        out.append((int(e.gate_ws), int(e.gate_ws_bytes), e._ws_tensor))
    return out


def _rb_prefetch(e, prefetch):
    for i, rng in enumerate((prefetch or [])[:2]):
        e.pf_ptr[i], e.pf_bytes[i] = rng[0], rng[1]


def rb_conv(x, weight, g, bias, in_act, out_scale, in_bn=None, coef=None, stats_pivot=None, prefetch=None):
    """y = (conv3x3(act(BN(x))) + bias) * out_scale with BatchNorm partials of y around stats_pivot (first half of a residual block).
    in_bn = (StatParts, pivot, bn) or coef = (scale, shift, ...) given. Returns (y, StatParts | None, coef)."""
    N, H, W, _ = x.shape
    y = torch.empty((N, H, W, g.Cout), dtype=torch.float32, device=x.device)
    d, _ = _rb_desc(x, weight, g, False, y, bias, None, None, in_act, out_scale)
    keep = None
    if in_bn is not None:
        coef, keep = _rb_in_bn(d, in_bn, x)
    else:
        d.in_scale, d.in_shift = ptr(coef[0]), ptr(coef[1])
    parts = None
    if stats_pivot is not None:
        rows = _C.load().lvae_resblock_conv_rows(C.byref(d))
        buf = torch.empty((rows + 1, 2, g.Cout), dtype=torch.float32, device=x.device)
        parts = StatParts(buf, rows, True)
        d.stats_out, d.stats_pivot = ptr(buf), ptr(stats_pivot)
    e = RbExt()
    e.prologue, e.epilogue = _C.RB_PRO_AFFINE, _C.RB_EPI_PLAIN
    _rb_prefetch(e, prefetch)
    call('lvae_resblock_conv_f32', C.byref(d), C.byref(e), stream_ptr())
    del keep
    return y, parts, coef


def rb_gate_rows(x, weight, g):
    """Workgroups (= statistics rows of `out`) of the forward conv + gate fusion for the 3x3 convolution (weight, g) on x: the whole-image
    kernels' shapes, or the 256-pixel six-product Winograd kernel's (fp32, 16x16 and 32x32 levels at batch 256). 0: not available."""
    if _ddi is not None or x.dtype != torch.float32 or x.dim() != 4 or g.transposed or g.KH != 3 or g.stride != 1 or g.pad != 1:
        return 0
    N, H, W, _ = x.shape
    d = _desc(g, weight, x, None, N, H, W, H, W, g.Cout, g.s_ci, g.s_co, GATHER_CONV)
    rows = int(_C.load().lvae_resblock_conv_rows(C.byref(d)))
    if rows > 0:
        return rows
    _conv_ws(d, weight, x.device)   # the Winograd kernel's transformed weights: the buffer the plain convolution of this layer uses too
    return int(_C.load().lvae_resblock_conv_gate_rows(C.byref(d)))


def rb_conv_gate(x, weight, g, bias, in_act, out_scale, gate_w, gate_g, gate_bias, res, act, in_bn=None, coef=None, stats_pivot=None,
                 prefetch=None):
    """Second half of a gated residual block in one launch: y2 = (conv3x3(act(BN(x))) + bias) * out_scale, ab = conv1x1(y2) + gate_bias,
    out = act(a) * sigmoid(b) + res, BatchNorm partials of out around stats_pivot. Returns (y2, ab, out, StatParts | None, coef)."""
    N, H, W, _ = x.shape
    dev = x.device
    y = torch.empty((N, H, W, g.Cout), dtype=torch.float32, device=dev)
    ab = torch.empty((N, H, W, gate_g.Cout), dtype=torch.float32, device=dev)
    out = torch.empty((N, H, W, gate_g.Cout // 2), dtype=torch.float32, device=dev)
    d, need = _rb_desc(x, weight, g, False, y, bias, None, None, in_act, out_scale)
    can_fold = True
    if not need:   # not a whole-image shape: the Winograd kernel with the gate behind it (its own transformed weights)
        _conv_ws(d, weight, dev)
        can_fold = bool(_C.load().lvae_conv2d_folds_bn_finalize(C.byref(d)))
    keep = None
    if in_bn is not None:
        coef, keep = _rb_in_bn(d, in_bn, x, can_fold)
    else:
        d.in_scale, d.in_shift = ptr(coef[0]), ptr(coef[1])
    e = RbExt()
    e.prologue, e.epilogue = _C.RB_PRO_AFFINE, _C.RB_EPI_GATE
    _rb_gate_ws(e, gate_w, gate_g, dev, False)
    e.gate_bias, e.act = ptr(gate_bias), ACT[act]
    e.res, e.ab, e.out = ptr(res), ptr(ab), ptr(out)
    parts = None
    if stats_pivot is not None:
        rows = _C.load().lvae_resblock_conv_gate_rows(C.byref(d))
        buf = torch.empty((rows + 1, 2, g.Cout), dtype=torch.float32, device=dev)
        parts = StatParts(buf, rows, True)
        e.out_stats, e.out_stats_pivot = ptr(buf), ptr(stats_pivot)
    _rb_prefetch(e, prefetch)
    call('lvae_resblock_conv_f32', C.byref(d), C.byref(e), stream_ptr())
    del keep
    return y, ab, out, parts, coef


def _rb_bn_bwd(d, bn_bwd, dx):
    """BatchNorm-backward sums of the dgrad result in the epilogue: bn_bwd = (x, coefficient block row 0, act). Returns the partial rows."""
    xb, coef0, act = bn_bwd
    rows = _C.load().lvae_resblock_conv_rows(C.byref(d))
    parts = torch.empty((rows, 2, dx.shape[3]), dtype=torch.float32, device=dx.device)
    d.stats_out, d.stats_pivot, d.stats_x = ptr(parts), ptr(coef0), ptr(xb)
    d.stats_mode, d.stats_act = 1, ACT[act]
    return parts


class PendingApply:
    """A BatchNorm-backward apply that its block did NOT launch: dx = BN'(dh; x) + add from the partial rows `parts`, to be formed by the
    first backward launch of the block that consumes dx (the previous block of the chain) and stored to `out` (ops.ResBlockFn)."""
    __slots__ = ('parts', 'dh', 'x', 'coef0', 'act', 'dgamma', 'dbeta', 'add', 'out')

    def __init__(self, parts, dh, x, coef0, act, dgamma, dbeta, add, out):
        self.parts, self.dh, self.x, self.coef0, self.act, self.dgamma, self.dbeta, self.add, self.out = parts, dh, x, coef0, act, dgamma, dbeta, add, out


def rb_gate_dgrad(dout, ab, gate_w, gate_g, act, drop, weight, g, bn_bwd, prefetch=None, apply=None):
    """Backward of rb_conv_gate up to the block's second BatchNorm in one launch: dab = gate'(dout, ab), dy2 = (dab . Wg^T) * drop,
    dh2 = dgrad3x3(dy2) with the BatchNorm-backward sums of bn_bwd = (y1, coef block row 0, act). Returns (dab, dy2, dh2, parts).
    apply (PendingApply): dout does not exist yet; the launch forms it in its prologue and stores it to apply.out (= dout's memory)."""
    N, H, W, Cn = dout.shape
    dev = dout.device
    dab = torch.empty_like(ab)
    dy2 = torch.empty((N, H, W, Cn), dtype=torch.float32, device=dev)
    dh = torch.empty((N, H, W, g.Cin), dtype=torch.float32, device=dev)
    d, _ = _rb_desc(dout, weight, g, True, dh)
    e = RbExt()
    e.prologue, e.epilogue = _C.RB_PRO_GATE_BWD, _C.RB_EPI_PLAIN
    _rb_gate_ws(e, gate_w, gate_g, dev, True)
    e.act = ACT[act]
    e.dout, e.ab_in, e.dab, e.pro_drop, e.xt_out = ptr(dout), ptr(ab), ptr(dab), ptr(drop), ptr(dy2)
    if apply is not None:
        e.ap_parts, e.ap_rows, e.ap_act, e.ap_M = ptr(apply.parts), apply.parts.shape[0], ACT[apply.act], N * H * W
        e.ap_coef, e.ap_dh, e.ap_x, e.ap_add = ptr(apply.coef0), ptr(apply.dh), ptr(apply.x), ptr(apply.add)
        e.ap_dgamma, e.ap_dbeta, e.ap_out = ptr(apply.dgamma), ptr(apply.dbeta), ptr(apply.out)
    parts = _rb_bn_bwd(d, bn_bwd, dh)
    _rb_prefetch(e, prefetch)
    call('lvae_resblock_conv_f32', C.byref(d), C.byref(e), stream_ptr())
    return dab, dy2, dh, parts


def rb_apply_dgrad(parts_in, dh_in, x_bn, coef0, act, dgamma, dbeta, drop, weight, g, bn_bwd, prefetch=None):
    """BatchNorm backward (training mode; parts_in = the sums the producer of dh_in left) + Dropout2d mask + dgrad of the block's first
    convolution in one launch: dy1 = BN'(dh_in; x_bn) * drop, dh1 = dgrad3x3(dy1) with the sums of bn_bwd = (x, coef block row 0, act).
    coef0: row 0 of the (4, C) coefficient block of the BatchNorm being differentiated. Returns (dy1, dh1, parts)."""
    N, H, W, Cn = dh_in.shape
    dev = dh_in.device
    dy1 = torch.empty((N, H, W, Cn), dtype=torch.float32, device=dev)
    dh = torch.empty((N, H, W, g.Cin), dtype=torch.float32, device=dev)
    d, _ = _rb_desc(dh_in, weight, g, True, dh)
    e = RbExt()
    e.prologue, e.epilogue = _C.RB_PRO_BN_APPLY, _C.RB_EPI_PLAIN
    e.bwd_parts, e.bwd_rows, e.bwd_act, e.bwd_M = ptr(parts_in), parts_in.shape[0], ACT[act], N * H * W
    e.bwd_coef, e.bwd_x, e.dgamma, e.dbeta, e.pro_drop, e.xt_out = ptr(coef0), ptr(x_bn), ptr(dgamma), ptr(dbeta), ptr(drop), ptr(dy1)
    parts = _rb_bn_bwd(d, bn_bwd, dh)
    _rb_prefetch(e, prefetch)
    call('lvae_resblock_conv_f32', C.byref(d), C.byref(e), stream_ptr())
    return dy1, dh, parts


def bn_coef_block(scale, shift, mean, rstd):
    """True when the four coefficient vectors are consecutive rows of one buffer (as bn_stats / bn_finalize_parts return them)."""
    if scale is None or shift is None or mean is None or rstd is None:
        return False
    n = 4 * scale.numel()
    p0 = scale.data_ptr()
    return shift.data_ptr() == p0 + n and mean.data_ptr() == p0 + 2 * n and rstd.data_ptr() == p0 + 3 * n


def affine_act_bwd_parts(parts, dh, x, scale, shift, act, mean, rstd, dgamma, dbeta, drop=None, add=None, out_bf16=False, out=None):
    """affine_act_bwd (training-mode BatchNorm) with the reduction already done by the epilogue of the convolution that produced dh
    (conv2d_dgrad(..., bn_bwd=...)). out: write into this tensor instead of a new one."""
    Cn = x.shape[-1]
    M = x.numel() // Cn
    dx = out if out is not None else torch.empty(x.shape, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    ws = workspace(8 * Cn, x.device)
    rows_per_n = M // x.shape[0]
    dtypes = _dt(dh) | (_dt(x) << 1) | (_dt(dx) << 2)
    call('lvae_affine_act_bwd_parts_f32', ptr(parts), parts.shape[0], ptr_dt(dh), ptr_dt(x), M, Cn, ptr(scale), ptr(shift), ACT[act],
         ptr(mean), ptr(rstd), ptr(dgamma), ptr(dbeta), ptr(drop), rows_per_n, ptr(add), ptr_dt(dx), ws.data_ptr(), ws.numel(),
         dtypes, stream_ptr())
    return dx


def conv2d_dgrad(dy, weight, g, in_hw, out_scale=None, ci_range=None, bn_bwd=None, out_bf16=False):
    """Gradient w.r.t. the conv input (before any fused input transform). dy NHWC (N,OH,OW,Cout) -> (N,H,W,Cin).
    out_scale (N,Cin) multiplies the result per (sample, channel) (Dropout2d mask of the producer).
    ci_range=(a,b) restricts the result to input channels [a,b) (the two halves of a fused channel concat)."""
    _chk_nhwc(dy, 'dy')
    N, OH, OW, Co = dy.shape
    H, W = in_hw
    a, b = ci_range if ci_range is not None else (0, g.Cin)
    dx = torch.empty((N, H, W, b - a), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=dy.device)
    d = _desc(g, weight, dy, None, N, OH, OW, H, W, b - a, g.s_co, g.s_ci,
              GATHER_CONV if g.transposed else GATHER_TRANSPOSED, out_scale=out_scale, y=dx)
    d.w = ptr(weight) + 4 * a * g.s_ci
    _conv_ws(d, weight, dy.device)
    if bn_bwd is None:
        call('lvae_conv2d_f32', C.byref(d), stream_ptr())
        return dx
    # bn_bwd = (x, scale, act): dx is the gradient w.r.t. act(BN(x)); ask the epilogue for the BatchNorm-backward partials
    # (scale must be the first row of the (4, C) coefficient block, see bn_coef_block). Returns (dx, parts | None).
    xb, coef, act = bn_bwd
    parts = None
    rows = _C.load().lvae_conv2d_stats_rows(C.byref(d))
    if rows > 0 and out_scale is None and ci_range is None and tuple(xb.shape) == tuple(dx.shape):
        parts = torch.empty((rows, 2, b - a), dtype=torch.float32, device=dy.device)
        d.stats_out, d.stats_pivot, d.stats_x = ptr(parts), ptr(coef), ptr_dt(xb)
        d.stats_mode, d.stats_act, d.stats_x_dtype = 1, ACT[act], _dt(xb)
    call('lvae_conv2d_f32', C.byref(d), stream_ptr())
    return dx, parts


def conv2d_wgrad_apply_ok(x, weight, g):
    """True when conv2d_wgrad_apply takes the 3x3 convolution (weight, g) on input x (the Winograd-domain fp32 weight gradient of a
    64 -> 64 layer at the >= 16x16 levels)."""
    if precision != PREC_F32 or x.dtype != torch.float32 or x.dim() != 4 or g.transposed or _ddi is not None:
        return False
    N, H, W, _ = x.shape
    d = _desc(g, weight, x, None, N, H, W, H, W, g.Cout, g.s_ci, g.s_co, GATHER_CONV)
    return bool(_C.load().lvae_conv2d_wgrad_apply_ok(C.byref(d)))


def conv2d_wgrad_apply(x, weight, g, dweight, dbias, parts, dh, xbn, coef0, act, dgamma, dbeta, drop=None, in_scale=None, in_shift=None, in_act=None):
    """conv2d_wgrad whose dy operand is the BatchNorm-backward apply that has not run: dy = affine_act_bwd_parts(parts, dh, xbn, ..., drop=drop)
    is formed inside the weight-gradient kernel, stored and returned (the dgrad that follows reads it). coef0: row 0 of the (4, C) coefficient
    block of the BatchNorm being differentiated; dgamma / dbeta are accumulated. Only where conv2d_wgrad_apply_ok()."""
    _chk_nhwc(x, 'x')
    N, H, W, _ = x.shape
    if tuple(dweight.stride()) != tuple(weight.stride()):
        raise _C.LvaeHipError("conv2d_wgrad_apply: gradient strides differ from weight strides")
    d = _desc(g, weight, x, None, N, H, W, H, W, g.Cout, g.s_ci, g.s_co, GATHER_CONV, None, in_scale, in_shift, in_act)
    need = _C.load().lvae_conv2d_wgrad_workspace(C.byref(d))
    ws = workspace(need, x.device)
    dy = torch.empty((N, H, W, g.Cout), dtype=torch.float32, device=x.device)
    ap = _C.BnApply(ptr(parts), parts.shape[0], ACT[act], N * H * W, ptr(coef0), ptr(dh), ptr(xbn), None, ptr(dgamma), ptr(dbeta), ptr(dy), 0, 0, ptr(drop))
    call('lvae_conv2d_wgrad_apply_f32', C.byref(d), C.byref(ap), ptr(dweight), ptr(dbias), ws.data_ptr(), ws.numel(), stream_ptr())
    return dy


def conv1x1_dgrad_cat(dy, weight, g, C1):
    """Both halves of the input gradient of a 1x1 / stride-1 convolution whose input was the channel concat (x [.., C1], x2 [.., Cin - C1]),
    in one launch: returns (dx, dx2), or None when the shape is not the single-shot 1x1 kernel's (the caller launches one dgrad per half)."""
    if not (g.KH == 1 and g.KW == 1 and g.stride == 1 and g.pad == 0 and not g.transposed and dy.dtype == torch.float32):
        return None
    K_, N_ = g.Cout, g.Cin
    kc = g.s_co == 1 and g.s_ci % 4 == 0
    nc = g.s_ci == 1 and g.s_co % 4 == 0
    if K_ > 128 or N_ > 128 or K_ % 4 or N_ % 4 or C1 % 4 or not (0 < C1 < N_) or not (kc or nc):
        return None
    _chk_nhwc(dy, 'dy')
    N, H, W, _ = dy.shape
    dx = torch.empty((N, H, W, C1), dtype=torch.float32, device=dy.device)
    dx2 = torch.empty((N, H, W, N_ - C1), dtype=torch.float32, device=dy.device)
    d = _desc(g, weight, dy, None, N, H, W, H, W, N_, g.s_co, g.s_ci, GATHER_TRANSPOSED, y=dx)
    call('lvae_conv1x1_dgrad_cat_f32', C.byref(d), ptr(dx2), C1, stream_ptr())
    return dx, dx2


def conv2d_wgrad(x, dy, weight, g, dweight, dbias=None, x2=None, in_scale=None, in_shift=None, in_act=None):
    """dweight += d/dw, dbias += d/db for the convolution of `conv2d` with the same fused input transform.
    dweight must have the same strides as weight (a view of the gradient arena)."""
    _chk_nhwc(x, 'x')
    _chk_nhwc(dy, 'dy')
    N, H, W, _ = x.shape
    OH, OW = dy.shape[1], dy.shape[2]
    if tuple(dweight.stride()) != tuple(weight.stride()):
        raise _C.LvaeHipError("conv2d_wgrad: gradient strides %s differ from weight strides %s" %
                              (tuple(dweight.stride()), tuple(weight.stride())))
    d = _desc(g, weight, x, x2, N, H, W, OH, OW, g.Cout, g.s_ci, g.s_co,
              GATHER_TRANSPOSED if g.transposed else GATHER_CONV, None, in_scale, in_shift, in_act)
    d.y_dtype = _dt(dy)
    need = _C.load().lvae_conv2d_wgrad_workspace(C.byref(d))
    ws = workspace(need, x.device)
    call('lvae_conv2d_wgrad_f32', C.byref(d), ptr_dt(dy), ptr(dweight), ptr(dbias), ws.data_ptr(), ws.numel(), stream_ptr())   # dy's type: d.y_dtype


def conv2d_wgrad_grouped(items):
    """items: list of (x, dy, weight, g, dweight, dbias, kw) exactly as for conv2d_wgrad (kw: x2 / in_scale / in_shift /
    in_act). One C call; gradients that share a kernel variant are launched together."""
    n = len(items)
    descs = (ConvDesc * n)()
    dys, dws, dbs = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
    for i, (x, dy, weight, g, dweight, dbias, kw) in enumerate(items):
        _chk_nhwc(x, 'x')
        _chk_nhwc(dy, 'dy')
        if tuple(dweight.stride()) != tuple(weight.stride()):
            raise _C.LvaeHipError("conv2d_wgrad_grouped: gradient strides %s differ from weight strides %s" %
                                  (tuple(dweight.stride()), tuple(weight.stride())))
        N, H, W, _ = x.shape
        d = _desc(g, weight, x, kw.get('x2'), N, H, W, dy.shape[1], dy.shape[2], g.Cout, g.s_ci, g.s_co,
                  GATHER_TRANSPOSED if g.transposed else GATHER_CONV, None, kw.get('in_scale'), kw.get('in_shift'), kw.get('in_act'))
        d.y_dtype = _dt(dy)
        C.memmove(C.byref(descs, i * C.sizeof(ConvDesc)), C.byref(d), C.sizeof(ConvDesc))
        dys[i], dws[i], dbs[i] = ptr_dt(dy), ptr(dweight), ptr(dbias)
    need = _C.load().lvae_conv2d_wgrad_grouped_workspace(descs, n)
    ws = workspace(need, items[0][0].device)
    call('lvae_conv2d_wgrad_grouped_f32', descs, C.cast(dys, C.c_void_p), C.cast(dws, C.c_void_p), C.cast(dbs, C.c_void_p), n,
         ws.data_ptr(), ws.numel(), stream_ptr())


# ----------------------------------------------------------------------------------------------------------------
def bn_stats(x, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.1):
    """Training-mode BatchNorm statistics of NHWC x. Returns (scale, shift, mean, rstd), each (C,)."""
    Cn = x.shape[-1]
    M = x.numel() // Cn
    out = torch.empty((4, Cn), dtype=torch.float32, device=x.device)
    need = _C.load().lvae_bn_stats_workspace(M, Cn)
    ws = workspace(need, x.device)
    call('lvae_bn_stats_f32', ptr(x), M, Cn, ptr(gamma), ptr(beta), eps, momentum, ptr(running_mean), ptr(running_var),
         ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), ws.data_ptr(), ws.numel(), stream_ptr())
    return out[0], out[1], out[2], out[3]


def bn_finalize_parts(parts, M, pivot, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.1):
    """BatchNorm coefficients from the partials a convolution epilogue wrote (conv2d(..., stats_pivot=pivot)); same outputs and
    running-statistics update as bn_stats. pivot may be running_mean itself."""
    rows, _, Cn = parts.shape
    out = torch.empty((4, Cn), dtype=torch.float32, device=parts.device)
    call('lvae_bn_finalize_parts_f32', ptr(parts), rows, M, Cn, ptr(pivot), ptr(gamma), ptr(beta), eps, momentum,
         ptr(running_mean), ptr(running_var), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), stream_ptr())
    return out[0], out[1], out[2], out[3]


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps=1e-5):
    Cn = running_mean.numel()
    out = torch.empty((2, Cn), dtype=torch.float32, device=running_mean.device)
    call('lvae_bn_eval_coeffs_f32', Cn, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), eps, ptr(out[0]),
         ptr(out[1]), stream_ptr())
    return out[0], out[1]


def affine_act(x, scale, shift, act, row_scale=None):
    Cn = x.shape[-1]
    M = x.numel() // Cn
    y = torch.empty_like(x)
    rows_per_n = M // x.shape[0]
    call('lvae_affine_act_f32', ptr(x), M, Cn, ptr(scale), ptr(shift), ACT[act], ptr(row_scale), rows_per_n, ptr(y),
         stream_ptr())
    return y


def affine_act_bwd(dh, x, scale, shift, act, bn_train, mean=None, rstd=None, dgamma=None, dbeta=None, drop=None, add=None):
    Cn = x.shape[-1]
    M = x.numel() // Cn
    dx = torch.empty_like(x)
    need = _C.load().lvae_bn_stats_workspace(M, Cn)
    ws = workspace(need, x.device)
    rows_per_n = M // x.shape[0]
    call('lvae_affine_act_bwd_f32', ptr(dh), ptr(x), M, Cn, ptr(scale), ptr(shift), ACT[act], int(bool(bn_train)),
         ptr(mean), ptr(rstd), ptr(dgamma), ptr(dbeta), ptr(drop), rows_per_n, ptr(add), ptr(dx), ws.data_ptr(),
         ws.numel(), stream_ptr())
    return dx


def gate_fwd(ab, res, act):
    Cn = ab.shape[-1] // 2
    M = ab.numel() // (2 * Cn)
    out = torch.empty(ab.shape[:-1] + (Cn,), dtype=torch.float32, device=ab.device)
    call('lvae_gate_fwd_f32', ptr(ab), ptr(res), M, Cn, ACT[act], ptr(out), stream_ptr())
    return out


def gate_bwd(dout, ab, act):
    Cn = ab.shape[-1] // 2
    M = ab.numel() // (2 * Cn)
    dab = torch.empty_like(ab)
    call('lvae_gate_bwd_f32', ptr(dout), ptr(ab), M, Cn, ACT[act], ptr(dab), stream_ptr())
    return dab


def act_bwd_from_out(dy, y, act):
    dx = torch.empty_like(y)
    call('lvae_act_bwd_from_out_f32', ptr(dy), ptr(y), y.numel(), ACT[act], ptr(dx), stream_ptr())
    return dx


def add(a, b):
    out = torch.empty_like(a)
    call('lvae_add_f32', ptr(a), ptr(b), a.numel(), ptr(out), stream_ptr())
    return out


def fill(t, value=0.0):
    """t[...] = value with our own kernel (t contiguous float32, 16-byte aligned): the gradient arena's zero_grad."""
    call('lvae_fill_f32', ptr(t), t.numel(), float(value), stream_ptr())
    return t


def fill_zero(t):
    """zero a (small) tensor with our own kernel: out = 0 * out is not safe for NaN garbage, so scale_rows_add of a cached zero"""
    z = _zeros_like_cached(t)
    call('lvae_scale_rows_add_f32', ptr(z), None, 1, 1, None, t.numel(), ptr(t), stream_ptr())
    return t


_zero_cache = {}


def _zeros_like_cached(t):
    key = (t.device, t.numel())
    if key not in _zero_cache:
        _zero_cache[key] = torch.zeros(t.numel(), dtype=torch.float32, device=t.device)
    return _zero_cache[key]


def add3(a, b, c):
    out = torch.empty_like(a)
    call('lvae_add3_f32', ptr(a), ptr(b), ptr(c), a.numel(), ptr(out), stream_ptr())
    return out


def sum_of_row_means(x_ln):
    L, N = x_ln.shape
    out = torch.empty((1,), dtype=torch.float32, device=x_ln.device)
    call('lvae_sum_of_row_means_f32', ptr(x_ln), L, N, ptr(out), stream_ptr())
    return out


def scale_rows_add(a, row_scale, b, out=None):
    """out = a * row_scale[n, c] + b  (either optional; with neither it is a device copy into `out`)."""
    Cn = a.shape[-1]
    M = a.numel() // Cn
    if out is None:
        out = torch.empty_like(a)
    call('lvae_scale_rows_add_f32', ptr(a), ptr(row_scale), M // a.shape[0], Cn, ptr(b), M, ptr(out), stream_ptr())
    return out


def colsum(x2d, out, accumulate):
    R, P = x2d.shape
    call('lvae_colsum_f32', ptr(x2d), R, P, ptr(out), int(bool(accumulate)), stream_ptr())


# ----------------------------------------------------------------------------------------------------------------
def normal_stochastic_fwd(p, q, eps_or_z, mode, analytical_kl, Z, N, rows=None):
    """p (N|1,H,W,2Z), q (N,H,W,2Z)|None. Returns z (N,H,W,Z), logprob_p, logprob_q, kl_samplewise (N,), kl_spatial (N,H,W).
    rows = (logprob_p row, kl row): (N,) views of the model's [L][N] matrices the kernel writes into directly (no stacking copies)."""
    p_bcast = int(p.shape[0] == 1 and N > 1)
    _, H, W, _ = p.shape
    dev = p.device
    z = torch.empty((N, H, W, Z), dtype=torch.float32, device=dev)
    small = torch.empty((3, N), dtype=torch.float32, device=dev)
    if rows is not None:
        small = [rows[0], small[1], rows[1]]
    ks = torch.empty((N, H, W), dtype=torch.float32, device=dev) if q is not None else None
    call('lvae_normal_stochastic_fwd_f32', ptr(p), p_bcast, ptr(q), ptr(eps_or_z), N, H * W, Z, mode, int(bool(analytical_kl)),
         ptr(z), ptr(small[0]), ptr(small[1]) if q is not None else None, ptr(small[2]) if q is not None else None, ptr(ks),
         stream_ptr())
    if q is None:
        return z, small[0], None, None, None
    return z, small[0], small[1], small[2], ks


def normal_stochastic_bwd(p, q, eps, z, dz, g_lp, g_lq, g_kl, g_ks, mode, analytical_kl, Z):
    N, H, W, _ = z.shape
    p_bcast = int(p.shape[0] == 1 and N > 1)
    dp = torch.empty((N, H, W, 2 * Z), dtype=torch.float32, device=z.device)
    dq = torch.empty_like(dp) if q is not None else None
    call('lvae_normal_stochastic_bwd_f32', ptr(p), p_bcast, ptr(q), ptr(eps), ptr(z), ptr(dz), ptr(g_lp), ptr(g_lq),
         ptr(g_kl), ptr(g_ks), N, H * W, Z, mode, int(bool(analytical_kl)), ptr(dp), ptr(dq), stream_ptr())
    return dp, dq


def kl_elementwise_fwd(p, q, z, analytical_kl):
    """p, q (N|1,H,W,2Z), z (N,H,W,Z) -> (N,H,W,Z): log q(z) - log p(z), or KL(q||p) when analytical_kl."""
    N, H, W, Z = z.shape
    out = torch.empty_like(z)
    call('lvae_kl_elementwise_fwd_f32', ptr(p), int(p.shape[0] == 1 and N > 1), ptr(q), int(q.shape[0] == 1 and N > 1), ptr(z),
         N, H * W, Z, int(bool(analytical_kl)), ptr(out), stream_ptr())
    return out


def kl_elementwise_bwd(p, q, z, g, analytical_kl, need_dz=True):
    N, H, W, Z = z.shape
    dp = torch.empty((N, H, W, 2 * Z), dtype=torch.float32, device=z.device)
    dq = torch.empty_like(dp)
    dz = torch.empty_like(z) if need_dz else None
    call('lvae_kl_elementwise_bwd_f32', ptr(p), int(p.shape[0] == 1 and N > 1), ptr(q), int(q.shape[0] == 1 and N > 1), ptr(z),
         ptr(g), N, H * W, Z, int(bool(analytical_kl)), ptr(dp), ptr(dq), ptr(dz), stream_ptr())
    return dp, dq, dz


# ----------------------------------------------------------------------------------------------------------------
def bernoulli_fwd(logits, x, u, need_grad):
    """logits, x, u: (N,H,W,C) NHWC. Returns mean, mode, sample, ll (N,) | None, dll_dlogits | None."""
    N = logits.shape[0]
    P = logits.numel() // N
    mean, mode, sample = torch.empty_like(logits), torch.empty_like(logits), torch.empty_like(logits)
    ll = torch.empty((N,), dtype=torch.float32, device=logits.device) if x is not None else None
    dll = torch.empty_like(logits) if (need_grad and x is not None) else None
    call('lvae_bernoulli_fwd_f32', ptr(logits), ptr(x), ptr(u), N, P, ptr(mean), ptr(mode), ptr(sample), ptr(ll), ptr(dll),
         stream_ptr())
    return mean, mode, sample, ll, dll


def dmol_ll_fwd(l, x, need_grad):
    """l (N,H,W,100), x (N,H,W,3) in [0,1]. Returns ll (N,), dll_dl | None."""
    N, H, W, Cp = l.shape
    ll = torch.empty((N,), dtype=torch.float32, device=l.device)
    dl = torch.empty_like(l) if need_grad else None
    need = _C.load().lvae_dmol_workspace(N, H * W)
    ws = workspace(need, l.device)
    call('lvae_dmol_ll_fwd_f32', ptr(l), ptr(x), N, H * W, Cp // 10, ptr(ll), ptr(dl), ws.data_ptr(), ws.numel(), stream_ptr())
    return ll, dl


def dmol_sample(l, u_mix, u_log):
    N, H, W, Cp = l.shape
    s = torch.empty((N, H, W, 3), dtype=torch.float32, device=l.device)
    call('lvae_dmol_sample_f32', ptr(l), ptr(u_mix), ptr(u_log), N, H * W, Cp // 10, ptr(s), stream_ptr())
    return s


def gaussian_fwd(params, x, eps, need_grad):
    """params (N,H,W,2C); x, eps (N,H,W,C). Returns sample, ll | None, dll_dparams | None."""
    N, H, W, C2 = params.shape
    Cn = C2 // 2
    sample = torch.empty((N, H, W, Cn), dtype=torch.float32, device=params.device)
    ll = torch.empty((N,), dtype=torch.float32, device=params.device) if x is not None else None
    dll = torch.empty_like(params) if (need_grad and x is not None) else None
    call('lvae_gaussian_fwd_f32', ptr(params), ptr(x), ptr(eps), N, H * W, Cn, ptr(sample), ptr(ll), ptr(dll), stream_ptr())
    return sample, ll, dll


def discr_logistic_fwd(raw, x, u, need_grad):
    """raw (N,H,W,2C); x, u (N,H,W,C). Returns mean, logscale, sample, ll | None, dll_draw | None."""
    N, H, W, C2 = raw.shape
    Cn = C2 // 2
    mk = lambda: torch.empty((N, H, W, Cn), dtype=torch.float32, device=raw.device)
    mean, ls, sample = mk(), mk(), mk()
    ll = torch.empty((N,), dtype=torch.float32, device=raw.device) if x is not None else None
    dll = torch.empty_like(raw) if (need_grad and x is not None) else None
    call('lvae_discr_logistic_fwd_f32', ptr(raw), ptr(x), ptr(u), N, H * W, Cn, ptr(mean), ptr(ls), ptr(sample), ptr(ll), ptr(dll),
         stream_ptr())
    return mean, ls, sample, ll, dll


def scale_per_sample(a, g):
    N = a.shape[0]
    out = torch.empty_like(a)
    call('lvae_scale_per_sample_f32', ptr(a), ptr(g), N, a.numel() // N, ptr(out), stream_ptr())
    return out


# ----------------------------------------------------------------------------------------------------------------
def upsample2x_fwd(x):
    N, H, W, Cn = x.shape
    y = torch.empty((N, 2 * H, 2 * W, Cn), dtype=torch.float32, device=x.device)
    call('lvae_upsample2x_fwd_f32', ptr(x), N, H, W, Cn, ptr(y), stream_ptr())
    return y


def upsample2x_bwd(dy):
    N, H2, W2, Cn = dy.shape
    dx = torch.empty((N, H2 // 2, W2 // 2, Cn), dtype=torch.float32, device=dy.device)
    call('lvae_upsample2x_bwd_f32', ptr(dy), N, H2 // 2, W2 // 2, Cn, ptr(dx), stream_ptr())
    return dx


def pad_crop(x, src_nchw, out_hw, dst_nchw):
    """Centred zero-pad or centre-crop + layout change. x is (N,C,H,W) contiguous when src_nchw else (N,H,W,C)."""
    if src_nchw:
        N, Cn, H, W = x.shape
    else:
        N, H, W, Cn = x.shape
    OH, OW = int(out_hw[0]), int(out_hw[1])
    shape = (N, Cn, OH, OW) if dst_nchw else (N, OH, OW, Cn)
    y = torch.empty(shape, dtype=torch.float32, device=x.device)
    call('lvae_pad_crop_f32', ptr(x), N, Cn, H, W, int(src_nchw), ptr(y), OH, OW, int(dst_nchw), stream_ptr())
    return y


# ----------------------------------------------------------------------------------------------------------------
def kl_bookkeeping_fwd(kl_ln, free_bits):
    L, N = kl_ln.shape
    dev = kl_ln.device
    kl_sep = torch.empty((N,), dtype=torch.float32, device=dev)
    kl_avg = torch.empty((L,), dtype=torch.float32, device=dev)
    scal = torch.empty((2,), dtype=torch.float32, device=dev)
    call('lvae_kl_bookkeeping_fwd_f32', ptr(kl_ln), L, N, float(free_bits), ptr(kl_sep), ptr(kl_avg), ptr(scal), stream_ptr())
    return kl_sep, kl_avg, scal


def kl_bookkeeping_bwd(kl_ln, free_bits, g_sep, g_avg, g_scal):
    L, N = kl_ln.shape
    dkl = torch.empty_like(kl_ln)
    call('lvae_kl_bookkeeping_bwd_f32', ptr(kl_ln), L, N, float(free_bits), ptr(g_sep), ptr(g_avg), ptr(g_scal), ptr(dkl),
         stream_ptr())
    return dkl


def elbo_loss_fwd(ll, kl_sep, kl_loss, beta):
    N = ll.numel()
    elbo_sep = torch.empty((N,), dtype=torch.float32, device=ll.device)
    scal = torch.empty((3,), dtype=torch.float32, device=ll.device)
    call('lvae_elbo_loss_fwd_f32', ptr(ll), ptr(kl_sep), ptr(kl_loss), float(beta), N, ptr(elbo_sep), ptr(scal), stream_ptr())
    return elbo_sep, scal


def elbo_loss_bwd(g_loss, beta, N):
    d_ll = torch.empty((N,), dtype=torch.float32, device=g_loss.device)
    d_kl = torch.empty((1,), dtype=torch.float32, device=g_loss.device)
    call('lvae_elbo_loss_bwd_f32', ptr(g_loss), float(beta), N, ptr(d_ll), ptr(d_kl), stream_ptr())
    return d_ll, d_kl


# ----------------------------------------------------------------------------------------------------------------
def iw_logmeanexp(elbo_sn):
    S, N = elbo_sn.shape
    out = torch.empty((N,), dtype=torch.float32, device=elbo_sn.device)
    call('lvae_iw_logmeanexp_f32', ptr(elbo_sn), S, N, ptr(out), stream_ptr())
    return out


def iw_online(state, mode, elbo=None, S=0, iw=None, mean=None):
    """state (3, N). mode 0 init / 1 accumulate elbo (N,) / 2 finalize into iw, mean (N,)."""
    call('lvae_iw_online_f32', ptr(elbo), ptr(state), state.shape[1], int(mode), int(S), ptr(iw), ptr(mean), stream_ptr())


def adamax_step(p, g, exp_avg, exp_inf, mask, lr, beta1, beta2, eps, weight_decay, gscale, step_count):
    call('lvae_adamax_step_f32', ptr(p), ptr(g), ptr(exp_avg), ptr(exp_inf), ptr(mask), p.numel(), lr, beta1, beta2, eps,
         weight_decay, ptr(gscale), step_count.data_ptr(), stream_ptr())
    prepared.weights_written()


def l2norm(x, out=None):
    if out is None:
        out = torch.empty((1,), dtype=torch.float32, device=x.device)
    need = _C.load().lvae_sumsq_workspace(x.numel())
    ws = workspace(need, x.device)
    call('lvae_l2norm_f32', ptr(x), x.numel(), ptr(out), ws.data_ptr(), ws.numel(), stream_ptr())
    return out


def rng_fill(out, kind, lo, hi, seed, offset, stream_id):
    """kind: 'normal' | 'uniform' | 'bernoulli' (keep-prob lo, kept value hi). offset: device int64[1] step counter."""
    k = {'normal': 0, 'uniform': 1, 'bernoulli': 2}[kind]
    call('lvae_rng_fill_f32', ptr(out), out.numel(), k, float(lo), float(hi), int(seed) & (2 ** 64 - 1),
         offset.data_ptr() if offset is not None else None, int(stream_id), stream_ptr())
    return out


def counter_advance(counter, by=1):
    call('lvae_counter_advance', counter.data_ptr(), int(by), stream_ptr())
