"""Autograd glue: torch.autograd.Function wrappers whose forward AND backward are hand-written HIP launches.

torch's autograd engine is used only to order the backward calls (plumbing). Parameter gradients are written by
the wgrad / reduction kernels straight into `param.grad` (a view of the flat gradient arena, see arena.py), so the
Functions return None for parameter inputs; activations flow through autograd normally.

All activations are NHWC (N,H,W,C) contiguous float32 CUDA tensors.
"""
import os

import torch
from torch.autograd import Function

from . import kernels as K

_const_cache = {}


def _ones_zeros(C, device):
    key = (C, device)
    if key not in _const_cache:
        _const_cache[key] = (torch.ones(C, device=device), torch.zeros(C, device=device))
    return _const_cache[key]


def grad_buf(p):
    """The accumulation buffer of a parameter (a view of the gradient arena once the model is packed)."""
    if not p.requires_grad:
        return None
    if p.grad is None:
        p.grad = torch.zeros_like(p)  # preserve_format keeps the arena strides
    return p.grad


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------------------------------------------
# Weight-gradient kernels feed nothing but the optimiser, so they do not have to sit on the critical path of the
# backward chain: with a side stream set (engine.TrainStep does it) every wgrad launch goes there, ordered after the
# producer of dy by a stream dependency, and the main stream joins once at the end of backward. Inside a captured
# hipGraph these become parallel branches. The scratch slabs are per stream (kernels.workspace).
_side = {'stream': None}


def set_wgrad_stream(streams):
    """None, one stream, or a list of streams the weight-gradient kernels are spread over (they depend on nothing but
    their inputs, and the small layers' launches fill only a fraction of the CUs each)."""
    if streams is None:
        _side['stream'] = None
    else:
        _side['stream'] = list(streams) if isinstance(streams, (list, tuple)) else [streams]


def join_wgrad_stream():
    sts = _side['stream']
    if sts is not None:
        for st in sts:
            torch.cuda.current_stream().wait_stream(st)


def set_wgrad_grouping(max_rows, flush_at=int(os.environ.get('LVAE_WGRAD_FLUSH', '1024'))):
    """Queue the weight gradients of layers with at most `max_rows` pixels (N*H*W) and issue them `flush_at` at a time (measured
    on the CIFAR-15 step: 12 -> 43.6 ms, 72 -> 42.8, 288 -> 42.4: fuller groups of each kernel variant; the queued (x, dy) pairs
    are at most 4 MB each) through
    lvae_conv2d_wgrad_grouped_f32 (None switches grouping off and flushes). The caller must call flush_wgrad_group() before
    anything reads the gradients."""
    flush_wgrad_group()
    _side['group_rows'] = max_rows
    _side['group_at'] = flush_at


def flush_wgrad_group():
    q, _side['group_q'] = _side.get('group_q') or [], []
    if len(q) == 1:
        x, dy, w, g, dw, db, kw = q[0]
        K.conv2d_wgrad(x, dy, w, g, dw, db, **kw)
    elif q:
        K.conv2d_wgrad_grouped(q)


def drop_wgrad_group():
    """Forget the queued weight gradients WITHOUT launching them: the pass they belong to was abandoned (an exception inside a step or
    inside its capture); their inputs may be tensors of a capture that no longer exists."""
    _side['group_q'] = []


def wgrad(x, dy, w, g, dw, db, **kw):
    rows = _side.get('group_rows')
    if rows is not None and x.shape[0] * x.shape[1] * x.shape[2] <= rows:
        q = _side.setdefault('group_q', [])
        if any(e[4].data_ptr() == dw.data_ptr() for e in q):
            flush_wgrad_group()  # two gradients of one weight must not share a launch
            q = _side['group_q']
        q.append((x, dy, w, g, dw, db, kw))
        if len(q) >= _side.get('group_at', 1024):
            flush_wgrad_group()
        return
    sts = _side['stream']
    if sts is None:
        return K.conv2d_wgrad(x, dy, w, g, dw, db, **kw)
    st = sts[(dw.data_ptr() >> 8) % len(sts)]  # one weight always on the same stream: its accumulations stay ordered
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        K.conv2d_wgrad(x, dy, w, g, dw, db, **kw)
    for t in (x, dy, kw.get('x2'), kw.get('in_scale'), kw.get('in_shift')):
        if t is not None:
            t.record_stream(st)  # the caching allocator must not recycle these blocks before the side kernel ran


# ----------------------------------------------------------------------------------------------------------------
_DGRAD_CAT = os.environ.get('LVAE_DGRAD_CAT', '1') != '0'   # A/B switch, profiling only
_WGRAD_APPLY = os.environ.get('LVAE_WGRAD_APPLY', '1') != '0'   # A/B switch, profiling only
_WGRAD_APPLY_MAXW = int(os.environ.get('LVAE_WGRAD_APPLY_MAXW', '16'))   # 32x32 measured +0.09 ms (profiles/r05_wgrad_apply_ab.txt)


class ConvFn(Function):
    """y = out_act(conv(cat(x, x2)) + bias); call sites: stem, pre_conv (strided / transposed), merge 1x1,
    stochastic convs, likelihood head. `mod` is the parameter holder (lib.nn.Conv2dParams)."""

    @staticmethod
    def forward(ctx, x, x2, mod, out_act, weight, bias):
        g = mod.geom()
        y = K.conv2d(x, weight, g, bias=bias, x2=x2, out_act=out_act)
        ctx.mod, ctx.out_act, ctx.g = mod, out_act, g
        ctx.has_x2 = x2 is not None
        ctx.save_for_backward(x, x2, y if out_act else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, x2, y = ctx.saved_tensors
        mod, g = ctx.mod, ctx.g
        dy = _c(dy)
        if ctx.out_act:
            dy = K.act_bwd_from_out(dy, y, ctx.out_act)
        w = mod.weight
        if w.requires_grad:
            wgrad(x, dy, w, g, grad_buf(w), grad_buf(mod.bias) if mod.bias is not None else None, x2=x2)
        dx = dx2 = None
        hw = (x.shape[1], x.shape[2])
        if x2 is None:
            if ctx.needs_input_grad[0]:
                dx = K.conv2d_dgrad(dy, w, g, hw)
        else:
            C1 = x.shape[3]
            both = K.conv1x1_dgrad_cat(dy, w, g, C1) if (_DGRAD_CAT and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]) else None
            if both is not None:
                dx, dx2 = both   # merge / skip 1x1: both halves of the channel concat from one launch
            else:
                if ctx.needs_input_grad[0]:
                    dx = K.conv2d_dgrad(dy, w, g, hw, ci_range=(0, C1))
                if ctx.needs_input_grad[1]:
                    dx2 = K.conv2d_dgrad(dy, w, g, hw, ci_range=(C1, g.Cin))
        return dx, dx2, None, None, None, None


def conv(x, mod, x2=None, out_act=None):
    return ConvFn.apply(x, x2, mod, out_act, mod.weight, mod.bias)


# ----------------------------------------------------------------------------------------------------------------
_GATE_STATS = os.environ.get('LVAE_NO_GATE_STATS') is None  # A/B switch, profiling only


# The fused blocks of the low-resolution levels run as one dependent chain, and every launch streams weights of its own that are cold in
# the L2s. Inside a block the first launch warms the L2s for the second (kernels.rb_weight_ranges); ACROSS blocks the order is only known
# from the previous step: each block remembers which fused block ran right after it (per direction) and what that block's first launch
# streams. Speed only: a stale link makes a launch touch bytes nobody needs; the link holds the scratch tensors themselves (third element of a
# range, kernels.rb_weight_ranges), so the addresses a launch — or a captured graph — touches stay allocated whatever the cache does.
import weakref as _weakref

_rb_chain = {'fwd': None, 'bwd': None}


def _rb_link(direction, blk, first_ranges):
    blk.__dict__['_rb_first_' + direction] = first_ranges
    prev = _rb_chain[direction]
    prev = prev() if prev is not None else None
    if prev is not None and prev is not blk:
        prev.__dict__['_rb_next_' + direction] = _weakref.ref(blk)
    _rb_chain[direction] = _weakref.ref(blk)


def _rb_next_ranges(direction, blk):
    nb = blk.__dict__.get('_rb_next_' + direction)
    nb = nb() if nb is not None else None
    return nb.__dict__.get('_rb_first_' + direction) if nb is not None else None


# The last launch of a residual block's backward is the BatchNorm-1 apply, dx = BN1'(dh1; x) + dout. When the tensor x is exactly the
# output of the previous residual block of the chain (lib/nn.py hands the producer over with the tensor object, like the BatchNorm
# partials) and that block's backward starts with a kernel that can form its `dout` itself (the fused gate-backward launches), the apply
# is not launched: the block returns an unwritten dx and leaves a kernels.PendingApply with the consumer, whose first launch computes dx
# in its prologue and stores it there. This assumes what loss.backward() does: the backward pass continues through the producing block
# (torch.autograd.grad with respect to a tensor BETWEEN two such blocks would be handed the unwritten dx). LVAE_DEFER_APPLY=0 keeps the launch.
_DEFER_APPLY = os.environ.get('LVAE_DEFER_APPLY', '1') != '0'
_DEFER_LARGE = os.environ.get('LVAE_DEFER_APPLY_LARGE', '1') != '0'   # ... also into the persistent gate-backward kernel of the >= 16x16 levels


def _take_pending(blk, dout):
    pend = blk.__dict__.pop('_pending_apply', None)
    if pend is not None and pend.out.data_ptr() != dout.data_ptr():
        raise K._C.LvaeHipError("deferred BatchNorm-backward apply: the gradient that reached the consuming block is not the tensor the "
                                "producer left unwritten (the block output has another consumer?)")
    return pend


class ResBlockFn(Function):
    """Whole pre-activation residual block ('bacdbacd' / 'bacdbac' recipes of lib/nn.py:64-89, with or without
    BatchNorm, Dropout2d and the gate) as ONE autograd node:

        y1 = drop1(conv1(act(bn1(x))))   -> one conv launch (BN-apply+act fused in the A-operand load, bias + dropout
        y2 = drop2(conv2(act(bn2(y1))))     scale in the epilogue) + one statistics pass per BN
        out = gate(conv1x1(y2)) + x      or  y2 + x

    Saved for backward: x, y1, y2, ab and the BN coefficients; act(bn(.)) is recomputed inside the wgrad loader.
    """

    @staticmethod
    def forward(ctx, x, blk, m1, m2, training, *params):
        act = blk.act
        dev = x.device
        C = x.shape[3]
        st = []
        h = x
        # BatchNorm partials of h written by the epilogue of the kernel that produced it: for x by the previous block's gate
        # kernel (handed over through blk._in_parts by lib/nn.py), for conv1's output by conv1 itself
        parts, pivot_in = blk.__dict__.pop('_in_parts', None) or (None, None)
        # compute_dtype bf16: conv outputs and gate pre-activations of the block (and, in backward, their gradients) live in bf16 where
        # every kernel involved has that form; the block's input / output (the residual stream) stay fp32
        # low-resolution levels (whole images per workgroup): two fused launches per direction (csrc/resblock_img.hip)
        full = (training and blk.gate is not None and blk.bn1 is not None and blk.bn2 is not None and blk.gate.bias is not None and
                blk.conv1.bias is not None and blk.conv2.bias is not None and blk.bn1.running_mean is not None and
                blk.bn2.running_mean is not None and x.dtype == torch.float32)
        rb_fwd, rb_bwd = K.rb_policy(x, blk.conv1.weight, blk.conv1.geom()) if full else (False, False)
        if rb_fwd:
            return ResBlockFn._forward_fused(ctx, x, blk, m1, m2, parts, pivot_in, rb_bwd)
        s16 = (not rb_bwd and blk.gate is not None and blk.bn1 is not None and blk.bn2 is not None and blk.gate.bias is not None and
               (training or not torch.is_grad_enabled()) and K.resblock_bf16_storage(x, blk.conv1.weight, blk.conv1.geom()))
        # larger levels (fp32): conv2 and the gate in ONE launch (the gate behind the Winograd kernel's epilogue, conv3x3_wino.hip)
        fuse_gate = bool(full and not s16 and K._RB_GATE_LARGE and K.rb_rows(x, blk.conv2.weight, blk.conv2.geom()) == 0 and
                         K.rb_gate_rows(x, blk.conv2.weight, blk.conv2.geom()) > 0)   # (whole-image shapes follow rb_policy above)
        fused_out = None
        for i, (bn, cv, m) in enumerate(((blk.bn1, blk.conv1, m1), (blk.bn2, blk.conv2, m2))):
            nxt = blk.bn2 if i == 0 else None  # conv1's output is BatchNorm 2's input: statistics in conv1's epilogue
            want_stats = training and nxt is not None and nxt.running_mean is not None
            if i == 1 and fuse_gate:
                in_bn = coef = None
                if parts is not None:
                    in_bn = (parts, bn.running_mean, bn)
                else:
                    coef = K.bn_stats(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum)
                pivot = st[0][3].detach() if (st[0][3] is not None and _GATE_STATS) else None
                gate = blk.gate
                y, ab_f, out_f, oparts_f, (sc, sh, mean, rstd) = K.rb_conv_gate(
                    h, cv.weight, cv.geom(), cv.bias, act, m, gate.weight, gate.geom(), gate.bias, x, act, in_bn=in_bn, coef=coef, stats_pivot=pivot)
                fused_out = (ab_f, out_f, oparts_f, pivot)
                st.append((h, sc, sh, mean, rstd))
                h = y
                continue
            if bn is not None and training and parts is not None:
                # statistics of h exist as partial sums: finalized inside the convolution where the kernel can (<= 4x4 levels)
                y, parts_out, (sc, sh, mean, rstd) = K.conv2d(
                    h, cv.weight, cv.geom(), bias=cv.bias, in_act=act, out_scale=m, in_bn=(parts, pivot_in if i == 0 else bn.running_mean, bn),
                    stats_pivot=nxt.running_mean if want_stats else None, out_bf16=s16)
            else:
                if bn is not None:
                    if training:
                        sc, sh, mean, rstd = K.bn_stats(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum)
                    else:
                        sc, sh = K.bn_eval_coeffs(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
                        mean = rstd = None
                else:
                    sc, sh = _ones_zeros(C, dev)
                    mean = rstd = None
                if want_stats:
                    y, parts_out = K.conv2d(h, cv.weight, cv.geom(), bias=cv.bias, in_scale=sc, in_shift=sh, in_act=act, out_scale=m,
                                            stats_pivot=nxt.running_mean, out_bf16=s16)
                else:
                    y, parts_out = K.conv2d(h, cv.weight, cv.geom(), bias=cv.bias, in_scale=sc, in_shift=sh, in_act=act, out_scale=m,
                                            out_bf16=s16), None
            parts = parts_out
            st.append((h, sc, sh, mean, rstd))
            h = y
        y2 = h
        ab = None
        blk.__dict__['_out_parts'] = None
        if fused_out is not None:
            ab, out, oparts, pivot = fused_out
            if oparts is not None:
                blk.__dict__['_out_parts'] = (oparts, pivot)
        elif blk.gate is not None:
            if training and blk.bn1 is not None and blk.bn1.running_mean is not None and _GATE_STATS:
                # the block output is (usually) the next block's BatchNorm input: statistics in the gate kernel's epilogue, around
                # this block's own running mean (same residual stream: a pivot inside the data range); the copy keeps the pivot
                # fixed when this block's statistics are updated before the consumer's finalize reads it
                pivot = st[0][3].detach() if st[0][3] is not None else blk.bn1.running_mean.clone()
                ab, out, oparts = K.conv1x1_gate(y2, blk.gate.weight, blk.gate.geom(), blk.gate.bias, x, act, stats_pivot=pivot)
                if oparts is not None:
                    blk.__dict__['_out_parts'] = (oparts, pivot)
            else:
                ab, out = K.conv1x1_gate(y2, blk.gate.weight, blk.gate.geom(), blk.gate.bias, x, act)
        else:
            out = K.add(y2, x)
        ctx.blk, ctx.training, ctx.s16 = blk, training, s16
        (x0, sc1, sh1, mean1, rstd1), (y1, sc2, sh2, mean2, rstd2) = st
        ctx.rb_bwd = bool(rb_bwd and ab is not None and K.bn_coef_block(sc1, sh1, mean1, rstd1) and K.bn_coef_block(sc2, sh2, mean2, rstd2))
        ctx.defer_to = blk.__dict__.pop('_in_src', None)
        acc = 'f32-dh' if ctx.rb_bwd else None   # what the first backward launch of this block can absorb: a deferred apply with an fp32 dh, or any
        if (not ctx.rb_bwd and _DEFER_LARGE and training and blk.gate is not None and blk.gate.bias is not None and blk.gate.weight.requires_grad and
                ab is not None and K.gate_bwd_fused_ok(x, blk.gate.weight, blk.gate.geom())):
            acc = 'any-dh'
        blk.__dict__['_accepts_deferred'] = acc
        ctx.save_for_backward(x0, y1, y2, ab, sc1, sh1, mean1, rstd1, sc2, sh2, mean2, rstd2, m1, m2)
        return out

    @staticmethod
    def _forward_fused(ctx, x, blk, m1, m2, parts, pivot_in, rb_bwd):
        """conv1 (+ BatchNorm-2 partial sums) and conv2 + gate + residual (+ the next block's BatchNorm partial sums): two launches."""
        act = blk.act
        bn1, bn2, cv1, cv2, gate = blk.bn1, blk.bn2, blk.conv1, blk.conv2, blk.gate
        in_bn = coef1 = None
        if parts is not None and parts.has_pivot:
            in_bn = (parts, pivot_in, bn1)
        elif parts is not None:
            N, H, W, _ = x.shape
            coef1 = K.bn_finalize_parts(parts.rows_view(), N * H * W, pivot_in, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var,
                                        bn1.eps, bn1.momentum)
        else:
            coef1 = K.bn_stats(x, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, bn1.eps, bn1.momentum)
        _rb_link('fwd', blk, K.rb_weight_ranges(x, cv1.weight, cv1.geom(), False))
        nxt = K.rb_weight_ranges(x, cv2.weight, cv2.geom(), False, gate=(gate.weight, gate.geom()))   # what the second launch will stream
        y1, parts2, coef1 = K.rb_conv(x, cv1.weight, cv1.geom(), cv1.bias, act, m1, in_bn=in_bn, coef=coef1, stats_pivot=bn2.running_mean,
                                      prefetch=nxt)
        blk.__dict__['_out_parts'] = None
        pivot = coef1[2].detach() if _GATE_STATS else None   # this block's own batch mean: inside the data range of the residual stream
        y2, ab, out, oparts, coef2 = K.rb_conv_gate(y1, cv2.weight, cv2.geom(), cv2.bias, act, m2, gate.weight, gate.geom(), gate.bias, x, act,
                                                    in_bn=(parts2, bn2.running_mean, bn2), stats_pivot=pivot,
                                                    prefetch=_rb_next_ranges('fwd', blk))
        if oparts is not None:
            blk.__dict__['_out_parts'] = (oparts, pivot)
        ctx.blk, ctx.training, ctx.s16 = blk, True, False
        ctx.rb_bwd = bool(rb_bwd and K.bn_coef_block(*coef1) and K.bn_coef_block(*coef2))
        ctx.defer_to = blk.__dict__.pop('_in_src', None)
        blk.__dict__['_accepts_deferred'] = 'f32-dh' if ctx.rb_bwd else None
        ctx.save_for_backward(x, y1, y2, ab, coef1[0], coef1[1], coef1[2], coef1[3], coef2[0], coef2[1], coef2[2], coef2[3], m1, m2)
        return out

    @staticmethod
    def backward(ctx, dout):
        blk = ctx.blk
        act = blk.act
        x, y1, y2, ab, sc1, sh1, mean1, rstd1, sc2, sh2, mean2, rstd2, m1, m2 = ctx.saved_tensors
        dout = _c(dout)
        hw = (x.shape[1], x.shape[2])
        s16 = ctx.s16   # bf16-stored block internals: every launch below then has to take the bf16-storage kernel (it raises otherwise)
        if ctx.rb_bwd:
            # low-resolution levels: gate backward + dgrad conv2, then BatchNorm-2 backward + dgrad conv1, then the BatchNorm-1 apply
            gate, w2, w1, bn2, bn1 = blk.gate, blk.conv2.weight, blk.conv1.weight, blk.bn2, blk.bn1
            gw = gate.weight
            _rb_link('bwd', blk, K.rb_weight_ranges(dout, w2, blk.conv2.geom(), True, gate=(gw, gate.geom()), gate_bwd=True))
            dab, dy2, dh2, parts2 = K.rb_gate_dgrad(dout, ab, gw, gate.geom(), act, m2, w2, blk.conv2.geom(), bn_bwd=(y1, sc2, act),
                                                    prefetch=K.rb_weight_ranges(dout, w1, blk.conv1.geom(), True), apply=_take_pending(blk, dout))
            if gw.requires_grad:
                wgrad(y2, dab, gw, gate.geom(), grad_buf(gw), grad_buf(gate.bias))
            wgrad(y1, dy2, w2, blk.conv2.geom(), grad_buf(w2), grad_buf(blk.conv2.bias), in_scale=sc2, in_shift=sh2, in_act=act)
            dy1, dh1, parts1 = K.rb_apply_dgrad(parts2, dh2, y1, sc2, act, grad_buf(bn2.weight), grad_buf(bn2.bias), m1, w1, blk.conv1.geom(),
                                                bn_bwd=(x, sc1, act), prefetch=_rb_next_ranges('bwd', blk))
            wgrad(x, dy1, w1, blk.conv1.geom(), grad_buf(w1), grad_buf(blk.conv1.bias), in_scale=sc1, in_shift=sh1, in_act=act)
            dx = ResBlockFn._final_apply(ctx, parts1, dh1, x, sc1, sh1, mean1, rstd1, act, bn1, dout)
            return (dx, None, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 5)
        pend = _take_pending(blk, dout)
        if blk.gate is not None:
            gw = blk.gate.weight
            dy2 = None
            if gw.requires_grad and blk.gate.bias is not None:
                # large levels: gate derivative, dgrad and the gate convolution's weight gradient in one persistent kernel
                dy2 = K.conv1x1_gate_bwd_wgrad(dout, ab, y2, gw, blk.gate.geom(), act, grad_buf(gw), grad_buf(blk.gate.bias), out_scale=m2,
                                               out_bf16=s16, apply=pend)
                if dy2 is not None:
                    pend = None
            if pend is not None:   # (the fused kernel did not take the shape after all: the apply gets its launch, writing the tensor dout is)
                pend = ResBlockFn._run_pending(pend)
            if dy2 is None:
                if s16:
                    raise K._C.LvaeHipError("bf16-stored residual block: the fused gate backward did not take this shape")
                dab, dy2 = K.conv1x1_gate_bwd(dout, ab, gw, blk.gate.geom(), act, out_scale=m2)
                wgrad(y2, dab, gw, blk.gate.geom(), grad_buf(gw), grad_buf(blk.gate.bias))
        else:
            if pend is not None:
                pend = ResBlockFn._run_pending(pend)
            dy2 = K.scale_rows_add(dout, m2, None) if m2 is not None else dout
        # second half
        w2 = blk.conv2.weight
        wgrad(y1, dy2, w2, blk.conv2.geom(), grad_buf(w2), grad_buf(blk.conv2.bias), in_scale=sc2, in_shift=sh2, in_act=act)
        bn2 = blk.bn2
        train2 = bn2 is not None and ctx.training
        parts2 = None
        if train2 and K.bn_coef_block(sc2, sh2, mean2, rstd2):  # BatchNorm-backward sums in the dgrad kernel's epilogue
            dh2, parts2 = K.conv2d_dgrad(dy2, w2, blk.conv2.geom(), hw, bn_bwd=(y1, sc2, act), out_bf16=s16)
        else:
            dh2 = K.conv2d_dgrad(dy2, w2, blk.conv2.geom(), hw)
        w1 = blk.conv1.weight
        wg1_done = False
        if (parts2 is not None and _WGRAD_APPLY and not s16 and _side['stream'] is None and w1.requires_grad and dh2.dtype == torch.float32 and
                x.shape[2] <= _WGRAD_APPLY_MAXW and K.conv2d_wgrad_apply_ok(x, w1, blk.conv1.geom())):
            # >= 16x16 levels (fp32): the BatchNorm-2 apply runs inside conv1's weight-gradient kernel, which needs its result as an operand
            # anyway (and stores it for the dgrad below): one launch, its finalize and one tensor pass less
            dy1 = K.conv2d_wgrad_apply(x, w1, blk.conv1.geom(), grad_buf(w1), grad_buf(blk.conv1.bias), parts2, dh2, y1, sc2, act,
                                       grad_buf(bn2.weight), grad_buf(bn2.bias), drop=m1, in_scale=sc1, in_shift=sh1, in_act=act)
            wg1_done = True
        elif parts2 is not None:
            dy1 = K.affine_act_bwd_parts(parts2, dh2, y1, sc2, sh2, act, mean2, rstd2, grad_buf(bn2.weight), grad_buf(bn2.bias),
                                         drop=m1, out_bf16=s16)
        else:
            dy1 = K.affine_act_bwd(dh2, y1, sc2, sh2, act, train2, mean2, rstd2,
                                   grad_buf(bn2.weight) if train2 else None, grad_buf(bn2.bias) if train2 else None, drop=m1)
        # first half
        if not wg1_done:
            wgrad(x, dy1, w1, blk.conv1.geom(), grad_buf(w1), grad_buf(blk.conv1.bias), in_scale=sc1, in_shift=sh1, in_act=act)
        bn1 = blk.bn1
        train1 = bn1 is not None and ctx.training
        parts1 = None
        if train1 and K.bn_coef_block(sc1, sh1, mean1, rstd1):
            dh1, parts1 = K.conv2d_dgrad(dy1, w1, blk.conv1.geom(), hw, bn_bwd=(x, sc1, act), out_bf16=s16)
        else:
            dh1 = K.conv2d_dgrad(dy1, w1, blk.conv1.geom(), hw)
        if parts1 is not None:
            dx = ResBlockFn._final_apply(ctx, parts1, dh1, x, sc1, sh1, mean1, rstd1, act, bn1, dout)
        else:
            dx = K.affine_act_bwd(dh1, x, sc1, sh1, act, train1, mean1, rstd1,
                                  grad_buf(bn1.weight) if train1 else None, grad_buf(bn1.bias) if train1 else None, add=dout)
        return (dx, None, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 5)


    @staticmethod
    def _run_pending(p):
        """The launch a deferred apply would have been (its consumer could not absorb it after all)."""
        n = p.coef0.numel()
        coef = p.coef0.new_empty(0).set_(p.coef0.untyped_storage(), p.coef0.storage_offset(), (4, n), (n, 1))
        K.affine_act_bwd_parts(p.parts, p.dh, p.x, coef[0], coef[1], p.act, coef[2], coef[3], p.dgamma, p.dbeta, add=p.add, out=p.out)
        return None

    @staticmethod
    def _final_apply(ctx, parts1, dh1, x, sc1, sh1, mean1, rstd1, act, bn1, dout):
        """dx = BN1'(dh1; x) + dout: launched here, or left to the first backward launch of the block that produced x (see _DEFER_APPLY)."""
        tgt = ctx.defer_to
        acc = tgt.__dict__.get('_accepts_deferred') if tgt is not None else None
        if (_DEFER_APPLY and acc is not None and x.dtype == torch.float32 and (dh1.dtype == torch.float32 or acc == 'any-dh') and
                K.bn_coef_block(sc1, sh1, mean1, rstd1)):
            dx = torch.empty_like(x)
            tgt.__dict__['_pending_apply'] = K.PendingApply(parts1, dh1, x, sc1, act, grad_buf(bn1.weight), grad_buf(bn1.bias), dout, dx)
            return dx
        return K.affine_act_bwd_parts(parts1, dh1, x, sc1, sh1, act, mean1, rstd1, grad_buf(bn1.weight), grad_buf(bn1.bias), add=dout)


# ----------------------------------------------------------------------------------------------------------------
# small nodes used by the non-fused 'cabdcabd' recipe (lib/nn.py:50-62)
class BnDropFn(Function):
    """y = BN(x) * mask[n, c] materialised (BatchNorm after the activation, then Dropout2d)."""

    @staticmethod
    def forward(ctx, x, bn, mask, training, *params):
        if bn is not None:
            if training:
                sc, sh, mean, rstd = K.bn_stats(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum)
            else:
                sc, sh = K.bn_eval_coeffs(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
                mean = rstd = None
        else:
            sc, sh = _ones_zeros(x.shape[3], x.device)
            mean = rstd = None
        y = K.affine_act(x, sc, sh, None, row_scale=mask)
        ctx.bn, ctx.training = bn, training
        ctx.save_for_backward(x, sc, sh, mean, rstd, mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, sc, sh, mean, rstd, mask = ctx.saved_tensors
        dy = _c(dy)
        if mask is not None:
            dy = K.scale_rows_add(dy, mask, None)
        bn = ctx.bn
        train = bn is not None and ctx.training
        dx = K.affine_act_bwd(dy, x, sc, sh, None, train, mean, rstd, grad_buf(bn.weight) if train else None,
                              grad_buf(bn.bias) if train else None)
        return (dx, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)


class GateFn(Function):
    """out = act(a) * sigmoid(b) [+ res] from ab = conv1x1 output (lib/nn.py:121-126)."""

    @staticmethod
    def forward(ctx, ab, res, act):
        ctx.act = act
        ctx.has_res = res is not None
        ctx.save_for_backward(ab)
        return K.gate_fwd(ab, res, act)

    @staticmethod
    def backward(ctx, dout):
        (ab,) = ctx.saved_tensors
        dout = _c(dout)
        return K.gate_bwd(dout, ab, ctx.act), (dout if ctx.has_res else None), None


class AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        return K.add(a, b)

    @staticmethod
    def backward(ctx, d):
        return d, d


# ----------------------------------------------------------------------------------------------------------------
class NormalStochFn(Function):
    """lib/stochastic.py:45-99 elementwise core. Returns z, logprob_p, logprob_q, kl_samplewise, kl_spatial."""

    @staticmethod
    def forward(ctx, p, q, noise, mode, analytical, Z, N, rows=None):
        z, lp, lq, kl, ks = K.normal_stochastic_fwd(p, q, noise, mode, analytical, Z, N, rows=rows)
        ctx.set_materialize_grads(False)  # unused outputs arrive as None (the kernel takes NULL) instead of zero-filled tensors
        ctx.mode, ctx.analytical, ctx.Z = mode, analytical, Z
        ctx.has_q = q is not None
        ctx.p_bcast = p.shape[0] == 1 and N > 1
        ctx.save_for_backward(p, q, noise if mode == 0 else None, z)
        if q is None:
            return z, lp
        return z, lp, lq, kl, ks

    @staticmethod
    def backward(ctx, dz, g_lp, g_lq=None, g_kl=None, g_ks=None):
        p, q, eps, z = ctx.saved_tensors
        cc = lambda t: None if t is None else _c(t)
        if dz is None and g_lp is None and g_lq is None and g_kl is None and g_ks is None:
            return None, None, None, None, None, None, None, None
        dp, dq = K.normal_stochastic_bwd(p, q, eps, z, cc(dz), cc(g_lp), cc(g_lq), cc(g_kl), cc(g_ks), ctx.mode,
                                         ctx.analytical, ctx.Z)
        if ctx.p_bcast:
            red = torch.empty_like(p)
            K.colsum(dp.view(dp.shape[0], -1), red.view(-1), False)
            dp = red
        return dp, dq, None, None, None, None, None, None


class KlElementwiseFn(Function):
    """`kl_elementwise` of lib/stochastic.py:88-91 / kl_normal_mc :209-226 as a differentiable node (off the training path)."""

    @staticmethod
    def forward(ctx, z, p, q, analytical):
        ctx.analytical = analytical
        ctx.save_for_backward(z, p, q)
        return K.kl_elementwise_fwd(p, q, z, analytical)

    @staticmethod
    def backward(ctx, g):
        z, p, q = ctx.saved_tensors
        dp, dq, dz = K.kl_elementwise_bwd(p, q, z, _c(g), ctx.analytical, need_dz=ctx.needs_input_grad[0])
        out = []
        for t, d in ((p, dp), (q, dq)):
            if t.shape[0] == 1 and d.shape[0] > 1:
                red = torch.empty_like(t)
                K.colsum(d.view(d.shape[0], -1), red.view(-1), False)
                d = red
            out.append(d)
        return dz, out[0], out[1], None


# ----------------------------------------------------------------------------------------------------------------
class BernoulliFn(Function):
    """lib/likelihoods.py:60-78: returns ll (N,) [differentiable], mean, mode, sample (NHWC, no grad)."""

    @staticmethod
    def forward(ctx, logits, x, u):
        mean, mode, sample, ll, dll = K.bernoulli_fwd(logits, x, u, True)
        ctx.save_for_backward(dll)
        ctx.mark_non_differentiable(mode, sample)
        ctx.has_ll = ll is not None
        if ll is None:
            ll = torch.zeros((logits.shape[0],), device=logits.device)
        return ll, mean, mode, sample

    @staticmethod
    def backward(ctx, g_ll, g_mean, g_mode, g_sample):
        (dll,) = ctx.saved_tensors
        if dll is None:
            return None, None, None
        return K.scale_per_sample(dll, _c(g_ll)), None, None


class GaussianFn(Function):
    """lib/likelihoods.py:81-114: returns ll (N,) [differentiable w.r.t. params] and the reparameterised sample."""

    @staticmethod
    def forward(ctx, params, x, eps):
        sample, ll, dll = K.gaussian_fwd(params, x, eps, True)
        ctx.save_for_backward(dll)
        ctx.mark_non_differentiable(sample)
        if ll is None:
            ll = torch.zeros((params.shape[0],), device=params.device)
        return ll, sample

    @staticmethod
    def backward(ctx, g_ll, g_sample):
        (dll,) = ctx.saved_tensors
        return (K.scale_per_sample(dll, _c(g_ll)) if dll is not None else None), None, None


class DiscrLogisticFn(Function):
    """lib/likelihoods.py:117-180: returns ll (N,), mean, logscale, sample (the last three carry no gradient)."""

    @staticmethod
    def forward(ctx, raw, x, u):
        mean, ls, sample, ll, dll = K.discr_logistic_fwd(raw, x, u, True)
        ctx.save_for_backward(dll)
        ctx.mark_non_differentiable(mean, ls, sample)
        if ll is None:
            ll = torch.zeros((raw.shape[0],), device=raw.device)
        return ll, mean, ls, sample

    @staticmethod
    def backward(ctx, g_ll, g_mean, g_ls, g_sample):
        (dll,) = ctx.saved_tensors
        return (K.scale_per_sample(dll, _c(g_ll)) if dll is not None else None), None, None


class DmolFn(Function):
    """lib/likelihoods.py:227-230 + 291-382: ll (N,) from params (N,H,W,100) and x (N,H,W,3) in [0,1]."""

    @staticmethod
    def forward(ctx, l, x):
        ll, dl = K.dmol_ll_fwd(l, x, True)
        ctx.save_for_backward(dl)
        return ll

    @staticmethod
    def backward(ctx, g_ll):
        (dl,) = ctx.saved_tensors
        return K.scale_per_sample(dl, _c(g_ll)), None


class SegmentMarkFn(Function):
    """Identity in forward. Its backward runs after every backward node of the model segment that starts at this tensor has been
    issued: the deferred (grouped) weight gradients of that segment are flushed and the gradient exchange is told the segment's slice
    of the gradient arena is complete (dist.GradAllReduce.segment_done)."""

    @staticmethod
    def forward(ctx, x, tracker, seg):
        ctx.tracker, ctx.seg = tracker, seg
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        flush_wgrad_group()
        ctx.tracker.segment_done(ctx.seg)
        return g, None, None


def segment_mark(x, tracker, seg):
    y = SegmentMarkFn.apply(x, tracker, seg)
    parts = getattr(x, '_lvae_bn_parts', None)   # BatchNorm partials travel with the tensor object (lib/nn.py)
    if parts is not None:
        y._lvae_bn_parts = parts
    return y


class FanoutFn(Function):
    """n aliases of an activation that has n consumers (a top-down layer's input feeds conv_in_p, the merge layer and the skip
    merger; a bottom-up value feeds the next bottom-up layer and its top-down layer): their gradients are summed by ONE launch of
    our own kernel instead of autograd's chain of n - 1 tensor adds."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [_c(g) for g in grads if g is not None]
        if not gs:
            return None, None
        while len(gs) > 1:
            gs = ([K.add3(gs[0], gs[1], gs[2])] + gs[3:]) if len(gs) >= 3 else [K.add(gs[0], gs[1])]
        return gs[0], None


def fanout(x, n):
    if n <= 1 or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    outs = FanoutFn.apply(x, n)
    parts = getattr(x, '_lvae_bn_parts', None)   # BatchNorm partials travel with the tensor object (lib/nn.py)
    if parts is not None:
        for o in outs:
            o._lvae_bn_parts = parts
    return outs


# ----------------------------------------------------------------------------------------------------------------
class UpsampleFn(Function):
    @staticmethod
    def forward(ctx, x):
        return K.upsample2x_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        return K.upsample2x_bwd(_c(dy))


class CropFn(Function):
    """centre crop (or pad) of an NHWC tensor; backward is the inverse pad (or crop)."""

    @staticmethod
    def forward(ctx, x, out_hw):
        ctx.in_hw = (x.shape[1], x.shape[2])
        return K.pad_crop(x, False, out_hw, False)

    @staticmethod
    def backward(ctx, dy):
        return K.pad_crop(_c(dy), False, ctx.in_hw, False), None


# ----------------------------------------------------------------------------------------------------------------
class KLBookFn(Function):
    """models/lvae.py:192-198: kl (L,N) -> kl_sep (N,), kl_avg_layerwise (L,), scalars [kl_loss, kl]."""

    @staticmethod
    def forward(ctx, kl_ln, free_bits):
        ctx.fb = free_bits
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(kl_ln)
        kl_sep, kl_avg, scal = K.kl_bookkeeping_fwd(kl_ln, free_bits)
        kl_loss, kl = scal[0], scal[1]
        ctx.mark_non_differentiable(kl)   # `kl` (the batch mean of kl_sep) is a metric
        return kl_sep, kl_avg, kl_loss, kl

    @staticmethod
    def backward(ctx, g_sep, g_avg, g_kl_loss, g_kl):
        (kl_ln,) = ctx.saved_tensors
        cc = lambda t: None if t is None else _c(t)
        if g_sep is None and g_avg is None and g_kl_loss is None:
            return None, None
        g_scal = None
        if g_kl_loss is not None:   # [d/d kl_loss, d/d kl = 0] as the kernel expects them, written by our own copy kernel
            g_scal = torch.empty((2,), dtype=torch.float32, device=kl_ln.device)
            K.scale_rows_add(cc(g_kl_loss).view(1, 1), None, None, out=g_scal[0:1])
            K.fill_zero(g_scal[1:2])
        return K.kl_bookkeeping_bwd(kl_ln, ctx.fb, cc(g_sep), cc(g_avg), g_scal), None


class StackFn(Function):
    """stack L per-layer (N,) vectors into one (L,N) matrix; backward hands the rows back as views."""

    @staticmethod
    def forward(ctx, *rows):
        L, N = len(rows), rows[0].numel()
        base = rows[0].data_ptr()
        if all(r.is_contiguous() and r.data_ptr() == base + 4 * N * i for i, r in enumerate(rows)):
            # the stochastic kernels wrote their per-sample sums straight into the rows of one [L][N] matrix (models/lvae.py)
            return rows[0].new_empty(0).set_(rows[0].untyped_storage(), rows[0].storage_offset(), (L, N), (N, 1))
        out = torch.empty((L, N), dtype=torch.float32, device=rows[0].device)
        for i, r in enumerate(rows):
            K.scale_rows_add(_c(r).view(1, -1), None, None, out=out[i])
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        return tuple(g[i] for i in range(g.shape[0]))


class ElboLossFn(Function):
    """experiment/experiment_manager.py:329-344: returns elbo_sep (N,), scalars [loss, elbo, recons]."""

    @staticmethod
    def forward(ctx, ll, kl_sep, kl_loss, beta):
        ctx.beta, ctx.N = beta, ll.numel()
        ctx.kl_dim0 = kl_loss.dim() == 0
        elbo_sep, scal = K.elbo_loss_fwd(ll, kl_sep, kl_loss, beta)
        loss, elbo, recons = scal[0], scal[1], scal[2]
        ctx.mark_non_differentiable(elbo_sep, elbo, recons)   # only d(loss) is propagated; elbo / recons are metrics
        return elbo_sep, loss, elbo, recons

    @staticmethod
    def backward(ctx, g_sep, g_loss, g_elbo, g_recons):
        d_ll, d_kl = K.elbo_loss_bwd(_c(g_loss).view(1), ctx.beta, ctx.N)
        return d_ll, None, d_kl.view(()) if ctx.kl_dim0 else d_kl, None
