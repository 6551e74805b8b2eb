"""ctypes binding of liblvae_hip.so (the C ABI declared in include/lvae_hip.h).

The product path has NO fallback: if the shared library is missing or a call is made without a GPU tensor the
import / call raises. PyTorch is used for device memory and streams only.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'liblvae_hip.so')

ACT = {None: 0, 'none': 0, 'elu': 1, 'relu': 2, 'leakyrelu': 3, 'selu': 4}
GATHER_CONV, GATHER_TRANSPOSED = 0, 1
PREC_F32, PREC_BF16 = 0, 1
DT_F32, DT_BF16 = 0, 1
VARIANT_DIRECT, VARIANT_POS, VARIANT_WINO_F32, VARIANT_WINO_SIX, VARIANT_BF16_DIRECT, VARIANT_SIX_DIRECT = 0, 2, 3, 4, 5, 6
WGRAD_VARIANT_GENERIC, WGRAD_VARIANT_IMG, WGRAD_VARIANT_BF16, WGRAD_VARIANT_WINO, WGRAD_VARIANT_DIRECT_1X1, WGRAD_VARIANT_TILE, WGRAD_VARIANT_THIN = range(7)
FORM_AUTO, FORM_F32_MFMA, FORM_SIX_PRODUCT, FORM_SIX_PRODUCT_DIRECT = 0, 1, 2, 3


class LvaeHipError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """mirror of struct lvae_conv_desc"""
    _fields_ = [
        ('x', C.c_void_p), ('x2', C.c_void_p), ('C1', C.c_int32), ('C2', C.c_int32),
        ('w', C.c_void_p), ('w_stap', C.c_int64), ('w_sk', C.c_int64), ('w_sn', C.c_int64),
        ('bias', C.c_void_p), ('in_scale', C.c_void_p), ('in_shift', C.c_void_p), ('in_act', C.c_int32),
        ('out_scale', C.c_void_p), ('out_act', C.c_int32), ('y', C.c_void_p),
        ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('OH', C.c_int32), ('OW', C.c_int32),
        ('Cout', C.c_int32), ('KH', C.c_int32), ('KW', C.c_int32), ('stride', C.c_int32), ('pad', C.c_int32),
        ('gather', C.c_int32), ('precision', C.c_int32), ('workspace', C.c_void_p), ('workspace_bytes', C.c_int64), ('workspace_ready', C.c_int32), ('stats_out', C.c_void_p), ('stats_pivot', C.c_void_p), ('stats_mode', C.c_int32), ('stats_act', C.c_int32),
        ('stats_x', C.c_void_p),
        ('in_fold', C.c_void_p),
        ('form', C.c_int32), ('x_dtype', C.c_uint8), ('y_dtype', C.c_uint8), ('stats_x_dtype', C.c_uint8), ('reserved_', C.c_uint8),
    ]


class BnFold(C.Structure):
    """mirror of struct lvae_bn_fold"""
    _fields_ = [('parts', C.c_void_p), ('rows', C.c_int32), ('M', C.c_int64), ('gamma', C.c_void_p), ('beta', C.c_void_p),
                ('eps', C.c_float), ('momentum', C.c_float), ('running_mean', C.c_void_p), ('running_var', C.c_void_p),
                ('coef_out', C.c_void_p)]


class BnApply(C.Structure):
    """mirror of struct lvae_bn_apply"""
    _fields_ = [('parts', C.c_void_p), ('rows', C.c_int32), ('act', C.c_int32), ('M', C.c_int64), ('coef', C.c_void_p), ('dh', C.c_void_p),
                ('x', C.c_void_p), ('add', C.c_void_p), ('dgamma', C.c_void_p), ('dbeta', C.c_void_p), ('out', C.c_void_p),
                ('dh_bf16', C.c_int32), ('reserved_', C.c_int32), ('drop', C.c_void_p)]


class RbExt(C.Structure):
    """mirror of struct lvae_rb_ext"""
    _fields_ = [('prologue', C.c_int32), ('epilogue', C.c_int32), ('gate_w', C.c_void_p), ('gate_w_sk', C.c_int64), ('gate_w_sn', C.c_int64),
                ('gate_ws', C.c_void_p), ('gate_ws_bytes', C.c_int64), ('gate_ws_ready', C.c_int32), ('gate_bias', C.c_void_p), ('act', C.c_int32), ('res', C.c_void_p), ('ab', C.c_void_p), ('out', C.c_void_p),
                ('out_stats', C.c_void_p), ('out_stats_pivot', C.c_void_p), ('dout', C.c_void_p), ('ab_in', C.c_void_p), ('dab', C.c_void_p),
                ('bwd_parts', C.c_void_p), ('bwd_rows', C.c_int32), ('bwd_act', C.c_int32), ('bwd_M', C.c_int64), ('bwd_coef', C.c_void_p),
                ('bwd_x', C.c_void_p), ('dgamma', C.c_void_p), ('dbeta', C.c_void_p), ('pro_drop', C.c_void_p), ('xt_out', C.c_void_p),
                ('pf_ptr', C.c_void_p * 2), ('pf_bytes', C.c_int64 * 2),
                ('ap_parts', C.c_void_p), ('ap_rows', C.c_int32), ('ap_act', C.c_int32), ('ap_M', C.c_int64), ('ap_coef', C.c_void_p),
                ('ap_dh', C.c_void_p), ('ap_x', C.c_void_p), ('ap_add', C.c_void_p), ('ap_dgamma', C.c_void_p), ('ap_dbeta', C.c_void_p),
                ('ap_out', C.c_void_p)]


RB_PRO_AFFINE, RB_PRO_BN_APPLY, RB_PRO_GATE_BWD = 0, 1, 2
RB_EPI_PLAIN, RB_EPI_GATE = 0, 1

ABI_VERSION = 16  # LVAE_ABI_VERSION of include/lvae_hip.h

_P, _I, _L, _F, _Z, _U = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t, C.c_uint64

# name -> (restype, argtypes); every symbol of include/lvae_hip.h
SIGNATURES = {
    'lvae_abi_version': (C.c_int, []),
    'lvae_last_error': (C.c_char_p, []),
    'lvae_conv2d_workspace': (_Z, [C.POINTER(ConvDesc)]),
    'lvae_conv2d_f32': (C.c_int, [C.POINTER(ConvDesc), _P]),
    'lvae_conv2d_bf16': (C.c_int, [C.POINTER(ConvDesc), _P]),
    'lvae_conv1x1_dgrad_cat_f32': (C.c_int, [C.POINTER(ConvDesc), _P, _I, _P]),
    'lvae_conv2d_stats_rows': (_I, [C.POINTER(ConvDesc)]),
    'lvae_conv2d_variant': (_I, [C.POINTER(ConvDesc)]),
    'lvae_resblock_bf16_storage': (_I, [C.POINTER(ConvDesc)]),
    'lvae_conv2d_folds_bn_finalize': (_I, [C.POINTER(ConvDesc)]),
    'lvae_conv2d_stats_buffer_rows': (_I, [C.POINTER(ConvDesc)]),
    'lvae_conv2d_prepare_entry_bytes': (_Z, []),
    'lvae_conv2d_prepare_entry': (C.c_int, [C.POINTER(ConvDesc), _P]),
    'lvae_conv2d_prepare_weights': (C.c_int, [_P, _I, _I, _P]),
    'lvae_conv1x1_gate_f32': (C.c_int, [C.POINTER(ConvDesc), _P, _I, _P, _P]),
    'lvae_conv1x1_gate_stats_rows': (_I, [C.POINTER(ConvDesc)]),
    'lvae_conv1x1_gate_bwd_f32': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _I, _P, _P]),
    'lvae_conv1x1_gate_bwd_wgrad_workspace': (_Z, [C.POINTER(ConvDesc)]),
    'lvae_conv1x1_gate_bwd_wgrad_f32': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _I, _P, _L, _L, _P, _P, _Z, C.POINTER(BnApply), _P]),
    'lvae_resblock_gate_workspace': (_Z, [C.POINTER(ConvDesc)]),
    'lvae_resblock_gate_prepare_entry': (C.c_int, [C.POINTER(ConvDesc), _P]),
    'lvae_resblock_conv_rows': (_I, [C.POINTER(ConvDesc)]),
    'lvae_resblock_conv_gate_rows': (_I, [C.POINTER(ConvDesc)]),
    'lvae_resblock_conv_workspace': (_Z, [C.POINTER(ConvDesc)]),
    'lvae_resblock_conv_prepare_entry': (C.c_int, [C.POINTER(ConvDesc), _P]),
    'lvae_resblock_conv_f32': (C.c_int, [C.POINTER(ConvDesc), C.POINTER(RbExt), _P]),
    'lvae_conv2d_wgrad_workspace': (_Z, [C.POINTER(ConvDesc)]),
    'lvae_conv2d_wgrad_f32': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _Z, _P]),
    'lvae_conv2d_wgrad_bf16': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _Z, _P]),
    'lvae_conv2d_wgrad_variant': (_I, [C.POINTER(ConvDesc)]),
    'lvae_conv2d_wgrad_apply_ok': (_I, [C.POINTER(ConvDesc)]),
    'lvae_conv2d_wgrad_apply_f32': (C.c_int, [C.POINTER(ConvDesc), C.POINTER(BnApply), _P, _P, _P, _Z, _P]),
    'lvae_conv2d_wgrad_grouped_workspace': (_Z, [C.POINTER(ConvDesc), _I]),
    'lvae_conv2d_wgrad_grouped_f32': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _I, _P, _Z, _P]),
    'lvae_bn_stats_workspace': (_Z, [_L, _I]),
    'lvae_bn_stats_f32': (C.c_int, [_P, _L, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    'lvae_bn_finalize_parts_f32': (C.c_int, [_P, _I, _L, _I, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P]),
    'lvae_bn_eval_coeffs_f32': (C.c_int, [_I, _P, _P, _P, _P, _F, _P, _P, _P]),
    'lvae_affine_act_f32': (C.c_int, [_P, _L, _I, _P, _P, _I, _P, _L, _P, _P]),
    'lvae_affine_act_bwd_f32': (C.c_int, [_P, _P, _L, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _L, _P, _P, _P, _Z, _P]),
    'lvae_affine_act_bwd_parts_f32': (C.c_int, [_P, _I, _P, _P, _L, _I, _P, _P, _I, _P, _P, _P, _P, _P, _L, _P, _P, _P, _Z, _I, _P]),
    'lvae_gate_fwd_f32': (C.c_int, [_P, _P, _L, _I, _I, _P, _P]),
    'lvae_gate_bwd_f32': (C.c_int, [_P, _P, _L, _I, _I, _P, _P]),
    'lvae_act_bwd_from_out_f32': (C.c_int, [_P, _P, _L, _I, _P, _P]),
    'lvae_add_f32': (C.c_int, [_P, _P, _L, _P, _P]),
    'lvae_add3_f32': (C.c_int, [_P, _P, _P, _L, _P, _P]),
    'lvae_sum_of_row_means_f32': (C.c_int, [_P, _I, _I, _P, _P]),
    'lvae_scale_rows_add_f32': (C.c_int, [_P, _P, _L, _I, _P, _L, _P, _P]),
    'lvae_colsum_f32': (C.c_int, [_P, _L, _L, _P, _I, _P]),
    'lvae_normal_stochastic_fwd_f32': (C.c_int, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    'lvae_normal_stochastic_bwd_f32': (C.c_int, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    'lvae_kl_elementwise_fwd_f32': (C.c_int, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P]),
    'lvae_kl_elementwise_bwd_f32': (C.c_int, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    'lvae_bernoulli_fwd_f32': (C.c_int, [_P, _P, _P, _I, _L, _P, _P, _P, _P, _P, _P]),
    'lvae_dmol_workspace': (_Z, [_I, _I]),
    'lvae_dmol_ll_fwd_f32': (C.c_int, [_P, _P, _I, _I, _I, _P, _P, _P, _Z, _P]),
    'lvae_dmol_sample_f32': (C.c_int, [_P, _P, _P, _I, _I, _I, _P, _P]),
    'lvae_gaussian_fwd_f32': (C.c_int, [_P, _P, _P, _I, _L, _I, _P, _P, _P, _P]),
    'lvae_discr_logistic_fwd_f32': (C.c_int, [_P, _P, _P, _I, _L, _I, _P, _P, _P, _P, _P, _P]),
    'lvae_scale_per_sample_f32': (C.c_int, [_P, _P, _I, _L, _P, _P]),
    'lvae_upsample2x_fwd_f32': (C.c_int, [_P, _I, _I, _I, _I, _P, _P]),
    'lvae_upsample2x_bwd_f32': (C.c_int, [_P, _I, _I, _I, _I, _P, _P]),
    'lvae_pad_crop_f32': (C.c_int, [_P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _P]),
    'lvae_kl_bookkeeping_fwd_f32': (C.c_int, [_P, _I, _I, _F, _P, _P, _P, _P]),
    'lvae_kl_bookkeeping_bwd_f32': (C.c_int, [_P, _I, _I, _F, _P, _P, _P, _P, _P]),
    'lvae_elbo_loss_fwd_f32': (C.c_int, [_P, _P, _P, _F, _I, _P, _P, _P]),
    'lvae_elbo_loss_bwd_f32': (C.c_int, [_P, _F, _I, _P, _P, _P]),
    'lvae_iw_online_f32': (C.c_int, [_P, _P, _I, _I, _I, _P, _P, _P]),
    'lvae_iw_logmeanexp_f32': (C.c_int, [_P, _I, _I, _P, _P]),
    'lvae_adamax_step_f32': (C.c_int, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _P, _P, _P]),
    'lvae_sumsq_workspace': (_Z, [_L]),
    'lvae_l2norm_f32': (C.c_int, [_P, _L, _P, _P, _Z, _P]),
    'lvae_rng_fill_f32': (C.c_int, [_P, _L, _I, _F, _F, _U, _P, _U, _P]),
    'lvae_counter_advance': (C.c_int, [_P, _U, _P]),
    'lvae_fill_f32': (C.c_int, [_P, _L, _F, _P]),
    'lvae_allreduce_unique_id': (C.c_int, [C.c_char_p, _P]),
    'lvae_allreduce_init': (C.c_int, [C.c_char_p, _P, _I, _I, C.POINTER(C.c_void_p)]),
    'lvae_allreduce_enqueue': (C.c_int, [_P, _P, _L, _P, _P, _P]),
    'lvae_allreduce_wait': (C.c_int, [_P, _P, _P]),
    'lvae_allreduce_destroy': (C.c_int, [_P]),
}

_lib = None


def load():
    """dlopen liblvae_hip.so and type every entry point. Raises LvaeHipError when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LvaeHipError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
                           "`make -C ladder-vae-pytorch_amd/csrc`. There is no CPU / PyTorch fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype, fn.argtypes = res, args
    if lib.lvae_abi_version() != ABI_VERSION:
        raise LvaeHipError("liblvae_hip.so ABI version %d, expected %d" % (lib.lvae_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t, dtypes=(torch.float32, torch.int64, torch.uint8)):
    """device pointer of a tensor (None -> NULL). Refuses CPU tensors: the HIP path never runs on host memory. fp32 (and the integer
    types of counters / scratch) only: an entry point that has a storage-type argument for the tensor takes it through ptr_dt()."""
    if t is None:
        return None
    if not t.is_cuda:
        raise LvaeHipError("lvae_hip kernels need CUDA/HIP tensors; got a %s tensor (no CPU fallback exists)" % t.device)
    if t.dtype not in dtypes:
        raise LvaeHipError("unexpected dtype %s for this entry point (it has no storage-type argument for this tensor)" % t.dtype)
    return t.data_ptr()


def ptr_dt(t):
    """device pointer of an activation tensor whose element type (fp32 | bf16) the callee is TOLD through a dtype field / mask of the
    same call (lvae_conv_desc.x_dtype / y_dtype / stats_x_dtype, the `dtypes` mask of lvae_affine_act_bwd_parts_f32)."""
    return ptr(t, (torch.float32, torch.bfloat16))


def check(rc, name):
    if rc != 0:
        raise LvaeHipError("%s failed (%d): %s" % (name, rc, load().lvae_last_error().decode()))


def call(name, *args):
    check(getattr(load(), name)(*args), name)
