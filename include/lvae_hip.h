/*
 * liblvae_hip.so — C ABI of the MI355X (gfx950) Ladder-VAE hot path.
 *
 * The reference (addtt/ladder-vae-pytorch) has no FFI/plugin boundary of its own: its hot path is stock torch
 * ops called from Python. Each entry point below therefore names the reference call site(s)
 * whose arithmetic it replaces (paths relative to the reference root), and INTEGRATION.md shows the ctypes
 * stub a maintainer of the reference would add.
 *
 * Conventions
 *  - activations are NHWC, contiguous, float32, device memory; `N` = batch.
 *  - the caller owns every buffer including workspaces; nothing here allocates, synchronises or touches the
 *    default stream. `stream` is a hipStream_t passed as void*.
 *  - every function returns 0 on success, a negative LVAE_E* code for a rejected argument, or the positive
 *    hipError_t of a failed launch. `lvae_last_error()` returns a static description of the last failure.
 *  - re-entrant. Mutable process state: the thread-local last-error string, and one atomic "dynamic-LDS attribute set" flag per
 *    kernel (an idempotent hipFuncSetAttribute on first use). The library reads NO environment variable: which kernel variant runs
 *    depends only on the descriptor — shape, alignment, `precision`, `workspace` and the `form` request below. Thresholds between
 *    variants are compile-time constants (csrc/lvae_common.h tune(); a -DLVAE_TUNING_ENV build, used by tools/ only, can sweep them
 *    through LVAE_* variables). Variants that were measured slower in round 2 (256-pixel Winograd workgroups, the persistent bf16 3x3
 *    kernel, the six-product weight gradient) are no longer compiled; phase-skip switches exist only in -DLVAE_PHASE_DEBUG builds.
 *  - the data-parallel gradient exchange is part of the library too (lvae_allreduce_*, at the end of this file): a private RCCL communicator
 *    whose ncclAllReduce runs on a side stream forked off / joined to the launch stream by the library; librccl is resolved at run time
 *    from the path the caller passes (this library has no link-time dependency on it).
 */
#ifndef LVAE_HIP_H
#define LVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LVAE_ABI_VERSION 16

#define LVAE_EINVAL (-1)   /* bad argument (null pointer, non-positive size, unsupported combination) */
#define LVAE_EALIGN (-2)   /* pointer / channel count not aligned as the vector path requires */
#define LVAE_EWORKSPACE (-3) /* workspace too small */

/* activation ids shared by every kernel (models/lvae.py:64-69 nonlin table) */
enum { LVAE_STATS_BN_FWD = 0, LVAE_STATS_BN_BWD = 1 };
enum { LVAE_PREC_F32 = 0, LVAE_PREC_BF16 = 1 };
/* Storage type of an activation tensor (lvae_conv_desc.x_dtype / y_dtype / stats_x_dtype and the `dtypes` masks below). bf16 storage exists
 * for the tensors INSIDE a residual block under precision = LVAE_PREC_BF16 (conv outputs y1 / y2, the gate pre-activations ab and their
 * gradients): what torch.autocast(bfloat16) stores for the reference's nn.Conv2d outputs (BASELINE configs[1], [3], [4]). The residual
 * stream, statistics, KL / likelihood terms, parameters and their gradients stay fp32. lvae_resblock_bf16_storage() says whether every kernel
 * of a block's forward and backward has the bf16-storage form for a shape; kernels without it reject bf16 tensors (LVAE_EINVAL). */
enum { LVAE_DT_F32 = 0, LVAE_DT_BF16 = 1 };
/* lvae_conv_desc.form: which arithmetic form an fp32 (LVAE_PREC_F32) descriptor asks for. All forms pass the same parity tests. */
enum {
  LVAE_FORM_AUTO = 0,               /* the library's choice for the shape (what the training step uses) */
  LVAE_FORM_F32_MFMA = 1,           /* every matrix product on v_mfma_f32_32x32x2_f32: Winograd / gate backward without the six-product form */
  LVAE_FORM_SIX_PRODUCT = 2,        /* six exact bf16-piece products per fp32 product wherever the selected kernel has that form (adds the
                                       GateLayer2d forward, which is HBM-bound and defaults to the fp32 MFMA) */
  LVAE_FORM_SIX_PRODUCT_DIRECT = 3  /* large 3x3 layers as a DIRECT six-product convolution instead of Winograd (measured: ties at 16x16, loses at 32x32) */
};
enum { LVAE_ACT_NONE = 0, LVAE_ACT_ELU = 1, LVAE_ACT_RELU = 2, LVAE_ACT_LEAKYRELU = 3, LVAE_ACT_SELU = 4 };

/* spatial gather of an implicit-GEMM convolution */
enum {
  LVAE_GATHER_CONV = 0,      /* ih = oh*stride - pad + kh           (nn.Conv2d forward; dgrad of ConvTranspose2d) */
  LVAE_GATHER_TRANSPOSED = 1 /* ih = (oh + pad - kh)/stride if exact (nn.ConvTranspose2d forward; dgrad of Conv2d) */
};

int lvae_abi_version(void);
const char* lvae_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32).
 *
 *   y[n,oh,ow,co] = out_act( (bias[co] + sum_{kh,kw,ci} T(x)[n,ih,iw,ci] * w[kh,kw,ci,co]) * out_scale[n,co] )
 *   T(x)[..ci] = in_act(x[..ci]*in_scale[ci] + in_shift[ci])   (identity when in_scale == NULL), zero outside the image
 *
 * replaces: nn.Conv2d / nn.ConvTranspose2d call sites lib/nn.py:83-87,118 ; lib/stochastic.py:25-27 ;
 *           models/lvae.py:73-76 ; models/lvae_layers.py:263-276,347-356 ; lib/likelihoods.py:55-58,199-202
 *           with the BatchNorm-apply + activation of lib/nn.py:80-82 fused into T, the Dropout2d scaling of
 *           lib/nn.py:89 fused into out_scale, torch.cat of models/lvae_layers.py:359 fused as (x, x2), and the
 *           autograd dgrad of all of them (same kernel, transposed weight strides, other gather).
 * ---------------------------------------------------------------------------------------------------------- */
/* Folded BatchNorm finalize of a convolution INPUT (training mode, nn.BatchNorm2d of lib/nn.py:80-81): instead of in_scale /
 * in_shift the kernel gets the partial sums the producer's statistics epilogue wrote — parts [rows + 1][2][C1], whose LAST row
 * holds that producer's pivot (kernels for which lvae_conv2d_folds_bn_finalize != 0, and the gate kernel, store it) — computes
 * scale / shift in its prologue (M = N*H*W elements per channel; gamma / beta may be NULL = 1 / 0), publishes (scale, shift,
 * mean, rstd) to coef_out [4][C1] for the backward and applies the momentum update to running_mean / running_var (may be NULL).
 * Only for descriptors with lvae_conv2d_folds_bn_finalize(d) != 0 (asked on the descriptor WITHOUT in_fold): the position-major
 * kernel of the <= 4x4 levels and the Winograd kernels with at most 64 input channels and at most 512 workgroups. parts must be
 * 16-byte aligned, C1 % 4 == 0. A kernel with lvae_conv2d_folds_bn_finalize(d) != 0 also WRITES rows + 1 rows of statistics
 * (stats_out, LVAE_STATS_BN_FWD): size that buffer accordingly. */
typedef struct lvae_bn_fold {
  const float* parts;
  int32_t rows;
  int64_t M;
  const float* gamma;
  const float* beta;
  float eps, momentum;
  float* running_mean;
  float* running_var;
  float* coef_out;
} lvae_bn_fold;

typedef struct lvae_conv_desc {
  const float* x;        /* [N,H,W,C1] */
  const float* x2;       /* [N,H,W,C2] or NULL: channels C1..C1+C2 of the logical input */
  int32_t C1, C2;
  const float* w;        /* element (tap, k, n) at w[tap*w_stap + k*w_sk + n*w_sn]; k = GEMM reduction channel */
  int64_t w_stap, w_sk, w_sn;
  const float* bias;     /* [Cout] or NULL */
  const float* in_scale; /* [C1+C2] or NULL */
  const float* in_shift; /* [C1+C2] (required when in_scale != NULL) */
  int32_t in_act;
  const float* out_scale; /* [N,Cout] or NULL */
  int32_t out_act;
  float* y;              /* [N,OH,OW,Cout] */
  int32_t N, H, W, OH, OW, Cout;
  int32_t KH, KW, stride, pad;
  int32_t gather;        /* LVAE_GATHER_* */
  int32_t precision;     /* LVAE_PREC_F32 (0): results as an fp32 multiply-add chain (fp32 MFMA, Winograd, or six exact bf16-piece
                            products per fp32 product: finite inputs — of any magnitude up to FLT_MAX, denormals included — only differ from the
                            fp32 MFMA by terms below 2^-24 of a product; an infinite operand gives NaN there, where an fp32 multiply would give
                            +-inf. Winograd (either form) spreads a non-finite input to every output of the tiles whose 4x4 block contains
                            it and to no other; tests/test_forms_gpu.py pins all of this); LVAE_PREC_BF16: operands rounded to bf16 at the matrix-core input, fp32
                            accumulate (kernel variants that have no bf16 form run in fp32) */
  void* workspace;       /* scratch for lvae_conv2d_f32 (transformed weights of the Winograd path) or NULL */
  int64_t workspace_bytes; /* lvae_conv2d_workspace(d) bytes enable every kernel variant; fewer select a variant needing none */
  int32_t workspace_ready; /* non-zero: `workspace` already holds this descriptor's transformed weights (written by
                              lvae_conv2d_prepare_weights after the last change of w); the launch skips its own transform */
  float* stats_out;       /* NULL, or [lvae_conv2d_stats_buffer_rows(d)][2][Cout]: per-workgroup partial BatchNorm statistics of the
                              OUTPUT y: (sum(y - pivot), sum((y - pivot)^2)) per channel, for lvae_bn_finalize_parts_f32 */
  const float* stats_pivot; /* [Cout] pivot of those sums (e.g. the running mean of the BatchNorm that consumes y) */
  /* stats_mode = LVAE_STATS_BN_BWD: y is the gradient dh w.r.t. h = act(x*scale + shift) of a training-mode BatchNorm; the
   * epilogue writes the partials of that BatchNorm's backward, (sum g, sum g*xhat) with g = y * act'(x*scale+shift) and
   * xhat = (x - mean)*rstd, to stats_out for lvae_affine_act_bwd_parts_f32. stats_x = x [N,OH,OW,Cout]; stats_pivot then points to
   * the [4][Cout] coefficient block (scale, shift, mean, rstd) as written by lvae_bn_stats_f32 into consecutive arrays. */
  int32_t stats_mode;     /* LVAE_STATS_BN_FWD (0, the default) or LVAE_STATS_BN_BWD */
  int32_t stats_act;      /* activation of that BatchNorm block (mode LVAE_STATS_BN_BWD) */
  const float* stats_x;
  /* Folded BatchNorm finalize of the INPUT: NULL, or a HOST pointer to the block below (read by the launcher, not by the device) */
  const struct lvae_bn_fold* in_fold;
  int32_t form;           /* LVAE_FORM_* (0 = LVAE_FORM_AUTO) */
  /* one byte each (they share the 8 bytes `form` occupies: grouped launches pass twelve descriptors in one 4 KB kernel-argument block) */
  uint8_t x_dtype;        /* LVAE_DT_*: element type of x (and x2); the pointer fields keep their float* spelling */
  uint8_t y_dtype;        /* LVAE_DT_*: element type of y (lvae_conv2d_wgrad_*: of dy) */
  uint8_t stats_x_dtype;  /* LVAE_DT_*: element type of stats_x */
  uint8_t reserved_;      /* 0 */
} lvae_conv_desc;

/* Scratch bytes lvae_conv2d_f32 can use for `d` (0 when no variant needs any). Large 3x3 / stride-1 / 64-channel layers run
 * as Winograd F(2x2,3x3) on the fp32 MFMA (2.25x fewer multiplies; coefficients 0, +-1, +-1/2, result within a few ulp of
 * the direct sum) when the scratch is supplied; without it the direct halo-tile kernel runs. */
size_t lvae_conv2d_workspace(const lvae_conv_desc* d);
int lvae_conv2d_f32(const lvae_conv_desc* d, void* stream);
/* dgrad of a 1x1 / stride-1 convolution whose INPUT was a channel concat (x, x2) — MergeLayer / SkipConnectionMerger,
 * models/lvae_layers.py:347-359 — in ONE launch (round 5): `d` describes the dgrad over all C1 + C2 input channels exactly as for lvae_conv2d_f32
 * (d->x = dy, d->Cout = C1 + C2, transposed weight strides); columns [0, split) are written to d->y [N,H,W,split], columns [split, Cout) to
 * dx2 [N,H,W,Cout - split]. At most 128 reduction and 128 output channels, multiples of 4; otherwise LVAE_EINVAL (use two launches with
 * a weight offset, as lvae_conv2d_f32 allows). */
int lvae_conv1x1_dgrad_cat_f32(const lvae_conv_desc* d, float* dx2, int32_t split, void* stream);
/* Which kernel family lvae_conv2d_f32 runs for `d` as given (workspace and form included): diagnostics for the parity tests and for
 * bench.py's roofline record (which matrix unit issues the FLOPs). */
enum {
  LVAE_VARIANT_DIRECT = 0,       /* fp32 MFMA, direct: halo-tile 3x3, single-shot 1x1 or the generic implicit GEMM */
  LVAE_VARIANT_POS = 2,          /* position-major 3x3 of the <= 4x4 levels (fp32 MFMA) */
  LVAE_VARIANT_WINO_F32 = 3,     /* Winograd F(2x2,3x3), position GEMMs on the fp32 MFMA */
  LVAE_VARIANT_WINO_SIX = 4,     /* Winograd F(2x2,3x3), position GEMMs as six bf16-piece products on the bf16 MFMA */
  LVAE_VARIANT_BF16_DIRECT = 5,  /* direct 3x3, bf16 operands on the bf16 MFMA (precision LVAE_PREC_BF16) */
  LVAE_VARIANT_SIX_DIRECT = 6    /* direct 3x3, six-product form (form LVAE_FORM_SIX_PRODUCT_DIRECT) */
};
int32_t lvae_conv2d_variant(const lvae_conv_desc* d);
/* 1 when a residual block whose 3x3 convolutions look like `d` (a FORWARD descriptor: 64 -> 64 channels, precision LVAE_PREC_BF16, its
 * workspace attached) can keep its internal tensors in bf16: forward, dgrad and weight gradient of the convolution, the GateLayer2d
 * forward and its fused backward, and the BatchNorm-backward apply all have a bf16-storage form for N x H x W. 0 otherwise. */
int32_t lvae_resblock_bf16_storage(const lvae_conv_desc* d);
/* The same convolution with bf16 matrix-core operands (v_mfma_f32_32x32x16_bf16): activations (after the fused input transform)
 * and weights are rounded to bf16, products are exact, accumulation, bias, statistics and the stored result are fp32 — the
 * arithmetic of the reference's nn.Conv2d call sites under torch.autocast(bfloat16) (BASELINE configs[1], [3], [4]). Equivalent to
 * lvae_conv2d_f32 on a descriptor with precision = LVAE_PREC_BF16. Set precision before asking lvae_conv2d_stats_rows. */
int lvae_conv2d_bf16(const lvae_conv_desc* d, void* stream);
/* BatchNorm statistics of the convolution OUTPUT fused into the producing kernel's epilogue (saves the separate pass over y):
 * rows of d->stats_out the launch will write, or 0 when the kernel variant this descriptor selects cannot produce them (then
 * leave stats_out NULL and use lvae_bn_stats_f32 on y). Set workspace / workspace_bytes before asking: the answer depends on
 * the variant. */
int32_t lvae_conv2d_stats_rows(const lvae_conv_desc* d);
/* 1 when the variant selected for `d` also stores its pivot behind the partial rows (stats_out then needs rows + 1 rows) and can
 * itself consume such a buffer through d->in_parts (the position-major kernel of the <= 4x4 levels); 0 otherwise. */
int32_t lvae_conv2d_folds_bn_finalize(const lvae_conv_desc* d);
/* Rows to ALLOCATE for d->stats_out: lvae_conv2d_stats_rows(d), plus one when the selected variant stores its pivot row behind them
 * (stats_mode LVAE_STATS_BN_FWD and lvae_conv2d_folds_bn_finalize, asked on the descriptor without its in_fold); 0 when there is no
 * statistics epilogue. Size the buffer
 * from this, read the partial rows [0, lvae_conv2d_stats_rows(d)). */
int32_t lvae_conv2d_stats_buffer_rows(const lvae_conv_desc* d);

/* Batched weight pre-transform: one launch for every convolution of a training step instead of one per convolution call.
 * For each descriptor with lvae_conv2d_workspace(d) > 0 give it a PRIVATE scratch buffer in d->workspace (kept until the
 * weights change), let lvae_conv2d_prepare_entry write its lvae_conv2d_prepare_entry_bytes()-byte table entry, copy the
 * table to device memory once, and call lvae_conv2d_prepare_weights(table, n, largest Cout) after every optimizer step;
 * convolutions then run with d->workspace_ready = 1. */
size_t lvae_conv2d_prepare_entry_bytes(void);
int lvae_conv2d_prepare_entry(const lvae_conv_desc* d, void* entry);
int lvae_conv2d_prepare_weights(const void* entries, int32_t n, int32_t max_cout, void* stream);

/* GateLayer2d forward fused with its 1x1 convolution and the residual add — lib/nn.py:118-126 and lib/nn.py:99.
 * `d` describes the 1x1 conv C -> 2C (d->y receives the pre-activations ab [N,H,W,2C] when non-NULL: the backward
 * reads them); out[..., c] = act(ab[..., c]) * sigmoid(ab[..., C + c]) + res[..., c]  (res may be NULL).
 * Supported: Cin <= 128, 2C <= 128, channel counts multiples of 4; otherwise LVAE_EINVAL (compose lvae_conv2d_f32 +
 * lvae_gate_fwd_f32 instead). */
int lvae_conv1x1_gate_f32(const lvae_conv_desc* d, const float* res, int32_t act, float* out, void* stream);
/* With d->stats_out / d->stats_pivot ([C] pivot) set, lvae_conv1x1_gate_f32 also writes BatchNorm partials of `out` — the next
 * residual block's BatchNorm input — as [lvae_conv1x1_gate_stats_rows(d) + 1][2][C]: the partial rows for
 * lvae_bn_finalize_parts_f32 / lvae_bn_fold, then one row whose first C floats are the pivot (0 rows: not supported for this
 * shape, leave stats_out NULL). */
int32_t lvae_conv1x1_gate_stats_rows(const lvae_conv_desc* d);
/* GateLayer2d backward fused with the dgrad of its 1x1 convolution (autograd of lib/nn.py:118-126): forms
 *   dab[m,c] = dout*sigmoid(b)*act'(a) ; dab[m,C+c] = dout*act(a)*sigmoid(b)*(1-sigmoid(b))      (a, b = the halves of ab)
 * in the kernel's operand staging (also written to `dab` [M,2C] when non-NULL: the weight gradient of the gate convolution
 * reads it) and computes dx = dab . W^T with the descriptor's epilogue. `d` describes that dgrad exactly as for
 * lvae_conv2d_f32 (C1 = 2C, transposed weight strides, y = dx, out_scale = dropout mask); d->x is ignored.
 * Same shape limits as lvae_conv1x1_gate_f32, otherwise LVAE_EINVAL (compose lvae_gate_bwd_f32 + lvae_conv2d_f32). */
int lvae_conv1x1_gate_bwd_f32(const lvae_conv_desc* d, const float* dout, const float* ab, int32_t act, float* dab,
                              void* stream);

/* GateLayer2d backward of a 64-channel block as ONE persistent kernel: gate derivative + input gradient (exactly
 * lvae_conv1x1_gate_bwd_f32, same descriptor) AND the weight / bias gradient of the gate convolution, dw[ci*dw_sk + co*dw_sn] +=
 * sum_m y[m,ci] * dab[m,co], db[co] += sum_m dab[m,co] (autograd of lib/nn.py:118-126): dab is formed once per 64-pixel tile in LDS and
 * never written to memory; y [M,64] is the convolution input the forward saved. Deterministic (per-workgroup partial slabs in
 * `workspace`, summed in a fixed order). lvae_conv1x1_gate_bwd_wgrad_workspace(d) == 0: shape not supported (needs C = 64 and at
 * least 16384 pixels) — use lvae_conv1x1_gate_bwd_f32 + lvae_conv2d_wgrad_f32. */
/* Deferred BatchNorm-backward apply (round 5; `ap` NULL or ap->parts NULL: none). In a chain of residual blocks the block that ran just
 * before this one in the backward ends with dx = BN1'(dh; x) + add (lvae_affine_act_bwd_parts_f32), and that dx IS this call's dout. With
 * `ap` the kernel reduces the partial rows parts [rows][2][64] (written by the producer of dh, stats_mode LVAE_STATS_BN_BWD) itself, forms
 * dout on the fly from (dh — element type dh_bf16 —, x, add or NULL; coefficient block coef [4][64] = scale, shift, mean, rstd; M = N*H*W),
 * stores it to out [M][64] (fp32) and accumulates dgamma / dbeta; `dout` is not read. One launch, its finalize launch and one tensor pass
 * per gated block less. Not with form LVAE_FORM_F32_MFMA. */
typedef struct lvae_bn_apply {
  const float* parts;
  int32_t rows;
  int32_t act;
  int64_t M;
  const float* coef;
  const float* dh;
  const float* x;
  const float* add;
  float* dgamma;
  float* dbeta;
  float* out;
  int32_t dh_bf16;
  int32_t reserved_;
  const float* drop;   /* lvae_conv2d_wgrad_apply_f32 only: Dropout2d mask [N][C] multiplied into the result, or NULL */
} lvae_bn_apply;
size_t lvae_conv1x1_gate_bwd_wgrad_workspace(const lvae_conv_desc* d);
int lvae_conv1x1_gate_bwd_wgrad_f32(const lvae_conv_desc* d, const float* dout, const float* ab, const float* y, int32_t act,
                                    float* dw, int64_t dw_sk, int64_t dw_sn, float* db, void* workspace, size_t workspace_bytes,
                                    const lvae_bn_apply* ap, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused residual-block kernels of the low-resolution levels (H*W a divisor of 64: 8x8, 4x4, 2x2 ...), "whole-image tiles"
 * (csrc/resblock_img.hip). A workgroup's 64 / 128 GEMM rows are whole images, so the 3x3 convolution of `d` (64 -> 64 channels,
 * stride 1, pad 1, fp32 tensors) can take its input from — and hand its output to — the neighbouring ops of a residual block
 * (lib/nn.py:78-99,118-126) through LDS instead of through memory and separate launches:
 *
 *   prologue  LVAE_RB_PRO_AFFINE    T(x) = in_act(x*in_scale + in_shift), exactly lvae_conv2d_f32's input transform; d->in_fold works too
 *             LVAE_RB_PRO_BN_APPLY  d->x = dh, the gradient w.r.t. h = act(BN(bwd_x)); the input of the convolution (a dgrad descriptor) is
 *                                   the training-mode BatchNorm backward of lvae_affine_act_bwd_parts_f32 — partial rows bwd_parts
 *                                   [bwd_rows][2][64] written by the producer of dh (stats_mode LVAE_STATS_BN_BWD), coefficient block
 *                                   bwd_coef [4][64], bwd_M = N*H*W, dgamma / dbeta accumulated by workgroup 0 — times the Dropout2d mask
 *                                   pro_drop [N][64]; also stored to xt_out [N,H,W,64] (the weight gradient of the producer reads it)
 *             LVAE_RB_PRO_GATE_BWD  the input of the convolution is the GateLayer2d backward of lvae_conv1x1_gate_bwd_f32: dab from
 *                                   (dout [M][64], ab_in [M][128]) — stored to dab when non-NULL — then (dab . gate_w^T) * pro_drop, stored to xt_out;
 *                                   with ap_parts != NULL dout itself is formed in the prologue (deferred BatchNorm-backward apply, see the struct)
 *   epilogue  LVAE_RB_EPI_PLAIN     y = (conv + bias) * out_scale and the statistics epilogue of lvae_conv2d_f32 (stats_out: lvae_resblock_conv_rows(d)
 *                                   rows, plus one pivot row behind them for LVAE_STATS_BN_FWD)
 *             LVAE_RB_EPI_GATE      (forward prologue only) y as above, then lvae_conv1x1_gate_f32 on it inside the same launch: ab [M][128]
 *                                   (optional), out = act(a) * sigmoid(b) + res, BatchNorm partials of out in out_stats [rows + 1][2][64]
 *
 * gate_w: element (reduction index k, output column n) at gate_w[k * gate_w_sk + n * gate_w_sn] — forward: k = input channel (64),
 * n = gate channel (128); backward: k = gate channel (128), n = input channel (64).
 * d->workspace: lvae_resblock_conv_workspace(d) bytes of pre-split weights, private to (w, orientation) like lvae_conv2d_f32's; filled by the
 * launch itself unless d->workspace_ready, or by lvae_conv2d_prepare_weights from a table entry written by lvae_resblock_conv_prepare_entry.
 * Arithmetic by d->precision: LVAE_PREC_F32 = six exact bf16-piece products per fp32 product (as LVAE_FORM_SIX_PRODUCT), LVAE_PREC_BF16 =
 * bf16 operands. lvae_resblock_conv_rows(d) == 0: shape not supported (use the one-kernel-per-op entry points).
 * replaces: lib/nn.py:80-89 (BatchNorm + activation + conv + Dropout2d), :118-126 + :99 (GateLayer2d + residual) and their autograd.
 * ---------------------------------------------------------------------------------------------------------- */
enum { LVAE_RB_PRO_AFFINE = 0, LVAE_RB_PRO_BN_APPLY = 1, LVAE_RB_PRO_GATE_BWD = 2 };
enum { LVAE_RB_EPI_PLAIN = 0, LVAE_RB_EPI_GATE = 1 };
typedef struct lvae_rb_ext {
  int32_t prologue, epilogue; /* LVAE_RB_PRO_*, LVAE_RB_EPI_* */
  const float* gate_w;    /* the 1x1 gate weight: only read when the pre-split copy below is not ready */
  int64_t gate_w_sk, gate_w_sn;
  void* gate_ws;          /* lvae_resblock_gate_workspace() bytes, private to (gate_w, direction): the pre-split gate weights */
  int64_t gate_ws_bytes;
  int32_t gate_ws_ready;  /* non-zero: gate_ws was written by lvae_conv2d_prepare_weights (table entry: lvae_resblock_gate_prepare_entry) after the last change of gate_w */
  const float* gate_bias; /* [128] or NULL (forward) */
  int32_t act;            /* activation of the gate (LVAE_ACT_*) */
  const float* res;       /* [M][64] or NULL */
  float* ab;              /* [M][128] or NULL */
  float* out;             /* [M][64] */
  float* out_stats;       /* NULL or [rows + 1][2][64] */
  const float* out_stats_pivot;
  const float* dout;      /* gate backward: [M][64] */
  const float* ab_in;     /* [M][128] */
  float* dab;             /* [M][128] or NULL */
  const float* bwd_parts; /* BatchNorm-apply prologue */
  int32_t bwd_rows;
  int32_t bwd_act;
  int64_t bwd_M;
  const float* bwd_coef;  /* [4][64]: scale, shift, mean, rstd */
  const float* bwd_x;     /* [M][64] */
  float* dgamma;          /* [64] accumulated, or NULL */
  float* dbeta;
  const float* pro_drop;  /* [N][64] or NULL */
  float* xt_out;          /* [M][64] or NULL */
  /* Optional L2 warm-up for the launch that FOLLOWS this one on the stream: up to two byte ranges (its pre-split weights — they are different
   * for every convolution of a step, i.e. cold in every XCD's L2 when their kernel starts) that this launch's workgroups touch once per
   * XCD while they wait for their own operands. Speed only: the ranges are read, never interpreted. */
  const void* pf_ptr[2];
  int64_t pf_bytes[2];
  /* Deferred BatchNorm-backward apply in front of LVAE_RB_PRO_GATE_BWD (round 5; ap_parts == NULL: none). In a chain of residual blocks the
   * block that ran just before this one in the backward ends with dx = BN1'(dh1; x) + dout (lvae_affine_act_bwd_parts_f32 with `add`), and
   * that dx IS this launch's dout. Instead of a launch of its own, this prologue forms it from (ap_parts [ap_rows][2][64] written by the
   * producer of ap_dh, coefficient block ap_coef [4][64], ap_M = N*H*W, ap_dh, ap_x, ap_add — all required), accumulates ap_dgamma / ap_dbeta
   * (workgroup 0) and stores it to ap_out [M][64]; `dout` is not read. */
  const float* ap_parts;
  int32_t ap_rows;
  int32_t ap_act;
  int64_t ap_M;
  const float* ap_coef;
  const float* ap_dh;
  const float* ap_x;
  const float* ap_add;
  float* ap_dgamma;
  float* ap_dbeta;
  float* ap_out;
} lvae_rb_ext;
/* Pre-split copy of a gate weight for lvae_resblock_conv_f32: `g` describes the 1x1 convolution in the direction it is used (forward:
 * C1 = 64, Cout = 128, w_sk / w_sn = strides of the input / gate channel; backward: C1 = 128, Cout = 64, strides swapped); only C1, Cout, w,
 * w_sk, w_sn, precision and workspace are read. */
size_t lvae_resblock_gate_workspace(const lvae_conv_desc* g);
int lvae_resblock_gate_prepare_entry(const lvae_conv_desc* g, void* entry);
int32_t lvae_resblock_conv_rows(const lvae_conv_desc* d);
/* Rows of out_stats (= workgroups) of an LVAE_RB_EPI_GATE launch. Besides the whole-image shapes (= lvae_resblock_conv_rows) the forward
 * conv + gate fusion exists for the shapes of the 256-pixel six-product Winograd kernel (fp32, 64 -> 64 channels: the 16x16 and 32x32 levels at
 * batch 256; lvae_conv2d_variant(d) == LVAE_VARIANT_WINO_SIX with at least 256 workgroups): attach THAT kernel's workspace
 * (lvae_conv2d_workspace / lvae_conv2d_prepare_entry) to d before asking and before the launch. 0: use lvae_conv2d_f32 + lvae_conv1x1_gate_f32. */
int32_t lvae_resblock_conv_gate_rows(const lvae_conv_desc* d);
size_t lvae_resblock_conv_workspace(const lvae_conv_desc* d);
int lvae_resblock_conv_prepare_entry(const lvae_conv_desc* d, void* entry);
int lvae_resblock_conv_f32(const lvae_conv_desc* d, const lvae_rb_ext* ext, void* stream);

/* Weight / bias gradient of the convolution described by `d` (d->y is unused, d->w gives only the strides):
 *   dw[tap,k,n] += sum_{n,oh,ow} T(x)[n,ih,iw,k] * dy[n,oh,ow,n]     db[n] += sum dy[..,n]
 * written with the strides d->w_stap/w_sk/w_sn into `dw` (accumulating). Deterministic: split-K partial slabs
 * in `workspace` are summed in a fixed order by a second kernel. `dy` is [N,OH,OW,Cout].
 * workspace bytes needed: lvae_conv2d_wgrad_workspace(d).
 * replaces: autograd's convolution_backward (weight, bias) for every call site listed above. */
size_t lvae_conv2d_wgrad_workspace(const lvae_conv_desc* d);
int lvae_conv2d_wgrad_f32(const lvae_conv_desc* d, const float* dy, float* dw, float* db, void* workspace,
                          size_t workspace_bytes, void* stream);
/* The same gradient with bf16 matrix-core operands (transposed LDS reads feed v_mfma_f32_32x32x16_bf16; fp32 accumulation, fp32 bias
 * gradient, fp32 partial slabs): lvae_conv2d_wgrad_f32 on a descriptor with precision = LVAE_PREC_BF16. Shapes without a bf16
 * kernel (anything but 3x3 / stride 1 / <= 64 input channels / >= 16384 pixels) run in fp32. */
int lvae_conv2d_wgrad_bf16(const lvae_conv_desc* d, const float* dy, float* dw, float* db, void* workspace, size_t workspace_bytes,
                           void* stream);
/* Weight gradient whose dY operand is the result of a BatchNorm-backward apply that has not run (round 5): dy = BN'(ap->dh; ap->x) * ap->drop
 * — exactly lvae_affine_act_bwd_parts_f32(parts, dh, x, ..., drop) — is formed while the kernel stages its operand, stored to ap->out
 * [N,H,W,64] for the dgrad that follows, and used as dY; ap->dgamma / dbeta are accumulated. One launch, its finalize launch and one
 * tensor pass per BatchNorm less. lvae_conv2d_wgrad_apply_ok(d) != 0: the Winograd-domain fp32 weight gradient of a 64 -> 64 layer with
 * W = 16 or 32 (the >= 16x16 levels at batch 256). ap->add must be NULL, dh fp32. */
int32_t lvae_conv2d_wgrad_apply_ok(const lvae_conv_desc* d);
int lvae_conv2d_wgrad_apply_f32(const lvae_conv_desc* d, const lvae_bn_apply* ap, float* dw, float* db, void* workspace,
                                size_t workspace_bytes, void* stream);
/* Which kernel family lvae_conv2d_wgrad_f32 (and the grouped call) runs for `d` (d->x_dtype / y_dtype = the storage types of x / dy):
 * diagnostics for the parity tests and the profiles. */
enum {
  LVAE_WGRAD_VARIANT_GENERIC = 0,   /* fp32 MFMA, generic implicit GEMM (strided / transposed / odd channel counts) */
  LVAE_WGRAD_VARIANT_IMG = 1,       /* whole-image tiles of the H*W <= 64 levels on the bf16 matrix pipe: six exact bf16-piece products per fp32
                                       product (LVAE_PREC_F32) or bf16 operands (LVAE_PREC_BF16); 3x3 (<= 64 -> <= 64 channels) and 1x1
                                       (<= 64 -> <= 128), up to 32 gradients per launch (csrc/conv_wgrad_img.hip, round 5) */
  LVAE_WGRAD_VARIANT_BF16 = 2,      /* 3x3, bf16 operands, >= 16384 pixels (whole-slab or half-slab form) */
  LVAE_WGRAD_VARIANT_WINO = 3,      /* Winograd-domain fp32 MFMA (large 3x3 layers) */
  LVAE_WGRAD_VARIANT_DIRECT_1X1 = 4,/* 1x1 at >= 32 k pixels straight from memory */
  LVAE_WGRAD_VARIANT_TILE = 5,      /* stride-1 "same" 3x3 / 1x1 on LDS-resident tiles, fp32 MFMA, wave-specialised */
  LVAE_WGRAD_VARIANT_THIN = 6       /* stems: one workgroup per image, vector ALU */
};
int32_t lvae_conv2d_wgrad_variant(const lvae_conv_desc* d);
/* n independent weight gradients (descs[i], dy[i], dw[i], db[i]; db[i] may be NULL): same results as n calls of
 * lvae_conv2d_wgrad_f32 in index order. The low-resolution levels of the ladder fill 16-64 CUs per gradient and depend on
 * nothing but their own inputs, so launches that share a kernel variant go out together (up to 12 per launch, one grouped
 * slab reduction); everything else is issued one by one. No two entries may accumulate into overlapping dw/db.
 * workspace: lvae_conv2d_wgrad_grouped_workspace(descs, n) bytes. */
size_t lvae_conv2d_wgrad_grouped_workspace(const lvae_conv_desc* descs, int32_t n);
int lvae_conv2d_wgrad_grouped_f32(const lvae_conv_desc* descs, const float* const* dy, float* const* dw, float* const* db,
                                  int32_t n, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * BatchNorm2d (training statistics) — lib/nn.py:80-81 (nn.BatchNorm2d, momentum 0.1, eps 1e-5)
 * ---------------------------------------------------------------------------------------------------------- */
/* Per-channel batch statistics of x [M,C] (M = N*H*W): writes
 *   scale[c] = gamma[c]*rstd[c], shift[c] = beta[c] - mean[c]*scale[c], mean[c], rstd[c]
 * and, when running_mean != NULL, updates running_mean/var with `momentum` (unbiased variance).
 * Two launches (per-chunk partial sums, fixed-order finalize: bitwise reproducible, no float atomics). A single launch with a
 * last-workgroup-done hand-off was measured and rejected: on the 8-XCD part every workgroup's agent-scope release/acquire
 * writes back and invalidates its XCD's L2 (+20 us per launch).
 * workspace: lvae_bn_stats_workspace(M, C) bytes. */
size_t lvae_bn_stats_workspace(int64_t M, int32_t C);
int lvae_bn_stats_f32(const float* x, int64_t M, int32_t C, const float* gamma, const float* beta, float eps,
                      float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                      float* mean, float* rstd, void* workspace, size_t workspace_bytes, void* stream);
/* Finalize from per-workgroup partials written by a convolution epilogue (lvae_conv_desc.stats_out): parts [rows][2][C] with
 * the pivot the producer used; same outputs and running-statistics update as lvae_bn_stats_f32 over M rows. `pivot` may alias
 * running_mean (each channel's pivot is read before its running mean is updated). */
int lvae_bn_finalize_parts_f32(const float* parts, int32_t rows, int64_t M, int32_t C, const float* pivot, const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                               float* scale, float* shift, float* mean, float* rstd, void* stream);
/* Inference form: scale/shift from the running statistics. */
int lvae_bn_eval_coeffs_f32(int32_t C, const float* gamma, const float* beta, const float* running_mean,
                            const float* running_var, float eps, float* scale, float* shift, void* stream);

/* y = act(x*scale[c] + shift[c]) materialised (used where no convolution consumes it: 'cabdcabd' blocks). */
int lvae_affine_act_f32(const float* x, int64_t M, int32_t C, const float* scale, const float* shift, int32_t act,
                        const float* row_scale, int64_t rows_per_n, float* y, void* stream);

/* Backward of h = act(x*scale + shift) with batch statistics (BatchNorm training) or fixed coefficients.
 *   g = dh * act'(u), u = x*scale+shift
 *   bn_train: dgamma += sum g*xhat, dbeta += sum g, dx = scale*(g - mean(g) - xhat*mean(g*xhat))
 *   else    : dx = scale*g
 * then dx *= drop[n,c] (optional Dropout2d mask of the producer) and dx += add (optional residual gradient).
 * Three launches (partial sums, finalize, apply); workspace lvae_bn_stats_workspace(M,C) bytes. */
int lvae_affine_act_bwd_f32(const float* dh, const float* x, int64_t M, int32_t C, const float* scale,
                            const float* shift, int32_t act, int32_t bn_train, const float* mean, const float* rstd,
                            float* dgamma, float* dbeta, const float* drop, int64_t rows_per_n, const float* add,
                            float* dx, void* workspace, size_t workspace_bytes, void* stream);
/* The training-mode case with the reduction already done by the epilogue of the convolution that produced dh
 * (lvae_conv_desc.stats_mode = LVAE_STATS_BN_BWD, parts [rows][2][C]): finalize + apply, two launches.
 * workspace: 2*C floats. */
int lvae_affine_act_bwd_parts_f32(const float* parts, int32_t rows, const float* dh, const float* x, int64_t M, int32_t C,
                                  const float* scale, const float* shift, int32_t act, const float* mean, const float* rstd,
                                  float* dgamma, float* dbeta, const float* drop, int64_t rows_per_n, const float* add,
                                  float* dx, void* workspace, size_t workspace_bytes, int32_t dtypes, void* stream);
/* dtypes: bit 0 dh, bit 1 x, bit 2 dx stored as bf16 (LVAE_DT_BF16), else fp32; `add` and everything else fp32. */

/* ------------------------------------------------------------------------------------------------------------
 * GateLayer2d epilogue + residual add — lib/nn.py:121-126 and lib/nn.py:99
 *   ab [M,2C] -> out[m,c] = act(ab[m,c]) * sigmoid(ab[m,C+c]) + res[m,c]      (res may be NULL)
 *   bwd: dab[m,c] = dout*sigmoid(b)*act'(a) ; dab[m,C+c] = dout*act(a)*sigmoid(b)*(1-sigmoid(b))
 * ---------------------------------------------------------------------------------------------------------- */
int lvae_gate_fwd_f32(const float* ab, const float* res, int64_t M, int32_t C, int32_t act, float* out, void* stream);
int lvae_gate_bwd_f32(const float* dout, const float* ab, int64_t M, int32_t C, int32_t act, float* dab, void* stream);

/* generic small elementwise helpers used by the host glue */
/* y = act(x) in place-capable; dact: dx = dy * act'(x) expressed from the OUTPUT y (elu/relu/leaky/selu allow it) */
int lvae_act_bwd_from_out_f32(const float* dy, const float* y, int64_t n, int32_t act, float* dx, void* stream);
int lvae_add_f32(const float* a, const float* b, int64_t n, float* out, void* stream);
/* out = (a + b) + c — the gradient of an activation with three consumers (TopDownLayer input: models/lvae_layers.py:134-160). */
int lvae_add3_f32(const float* a, const float* b, const float* c, int64_t n, float* out, void* stream);
/* out[0] = sum_l mean_n x[l][n] — `logp` of models/lvae.py:301 from the [L][N] matrix of per-layer, per-sample log p(z). */
int lvae_sum_of_row_means_f32(const float* x, int32_t L, int32_t N, float* out, void* stream);
/* residual without gate (lib/nn.py:99 when gated is falsy): out = (a*rowscale) + b */
int lvae_scale_rows_add_f32(const float* a, const float* row_scale, int64_t rows_per_n, int32_t C, const float* b,
                            int64_t M, float* out, void* stream);
/* out[p] (+)= sum_r x[r, p]  (batch reduction of a broadcast parameter's gradient, e.g. top_prior_params) */
int lvae_colsum_f32(const float* x, int64_t R, int64_t P, float* out, int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * NormalStochasticBlock2d elementwise core — lib/stochastic.py:45-99 and kl_normal_mc lib/stochastic.py:209-226
 * p, q: [N,HW,2Z] (mu = channels [0,Z), logvar = [Z,2Z)); p may be batch-broadcast (p_bcast=1: [1,HW,2Z]); q may be
 * NULL (sample from p). mode: 0 = z = mu + exp(lv/2)*eps, 1 = z = mu (use_mode), 2 = z given (forced_latent).
 * outputs: z [N,HW,Z]; logprob_p, logprob_q, kl_samplewise [N]; kl_spatial [N,HW] (analytical, sum over Z).
 * ---------------------------------------------------------------------------------------------------------- */
int lvae_normal_stochastic_fwd_f32(const float* p, int32_t p_bcast, const float* q, const float* eps, int32_t N,
                                   int32_t HW, int32_t Z, int32_t mode, int32_t analytical_kl, float* z,
                                   float* logprob_p, float* logprob_q, float* kl_samplewise, float* kl_spatial,
                                   void* stream);
/* Backward. Upstream: dz [N,HW,Z] (may be NULL), g_lp, g_lq, g_kl [N] (may be NULL), g_ks [N,HW] (may be NULL).
 * Writes dp [N,HW,2Z] (caller sums over N when p was broadcast) and dq [N,HW,2Z] (NULL when q is NULL). */
int lvae_normal_stochastic_bwd_f32(const float* p, int32_t p_bcast, const float* q, const float* eps, const float* z,
                                   const float* dz, const float* g_lp, const float* g_lq, const float* g_kl,
                                   const float* g_ks, int32_t N, int32_t HW, int32_t Z, int32_t mode,
                                   int32_t analytical_kl, float* dp, float* dq, void* stream);

/* `kl_elementwise` of NormalStochasticBlock2d.forward (lib/stochastic.py:88-91,108) and kl_normal_mc (lib/stochastic.py:209-226):
 * out[n,pix,c] = log q(z) - log p(z)  (analytical_kl = 0)  or  KL(q || p)  (analytical_kl = 1).  p, q [N|1,HW,2Z] (mu | logvar on
 * the channel axis; *_bcast = 1: one row set shared by the batch), z, out [N,HW,Z]. The backward writes dp, dq [N,HW,2Z] (the
 * caller sums a broadcast operand over N) and, when dz != NULL, dz [N,HW,Z] (zero for the analytical form). */
int lvae_kl_elementwise_fwd_f32(const float* p, int32_t p_bcast, const float* q, int32_t q_bcast, const float* z, int32_t N,
                                int32_t HW, int32_t Z, int32_t analytical_kl, float* out, void* stream);
int lvae_kl_elementwise_bwd_f32(const float* p, int32_t p_bcast, const float* q, int32_t q_bcast, const float* z, const float* g,
                                int32_t N, int32_t HW, int32_t Z, int32_t analytical_kl, float* dp, float* dq, float* dz,
                                void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Likelihood heads (elementwise part; the 3x3 parameter conv is lvae_conv2d_f32)
 * ---------------------------------------------------------------------------------------------------------- */
/* Bernoulli — lib/likelihoods.py:60-78,385-388. logits [N,P] (P = H*W*C), x [N,P] or NULL, u [N,P] uniforms.
 * mean = sigmoid(logits), mode = round(mean), sample = u < mean, ll[n] = sum x*max(log m,-100)+(1-x)*max(log(1-m),-100)
 * dll_dlogits [N,P] (optional) = d ll[n] / d logits. */
int lvae_bernoulli_fwd_f32(const float* logits, const float* x, const float* u, int32_t N, int64_t P, float* mean,
                           float* mode, float* sample, float* ll, float* dll_dlogits, void* stream);
/* Discretized mixture of logistics — lib/likelihoods.py:291-382 (x given in [0,1]; the 2x-1 of :228 is applied here).
 * l [N,HW,10*nmix], x [N,HW,3]; ll [N]; dll_dl [N,HW,10*nmix] optional. */
size_t lvae_dmol_workspace(int32_t N, int32_t HW);
int lvae_dmol_ll_fwd_f32(const float* l, const float* x, int32_t N, int32_t HW, int32_t nmix, float* ll, float* dll_dl,
                         void* workspace, size_t workspace_bytes, void* stream);
/* lib/stochastic.py:141-206 + the (s+1)/2 clamp of lib/likelihoods.py:221-225.
 * u_mix [N,HW,nmix], u_log [N,HW,3] uniforms in (1e-5, 1-1e-5); sample [N,HW,3] in [0,1]. */
int lvae_dmol_sample_f32(const float* l, const float* u_mix, const float* u_log, int32_t N, int32_t HW, int32_t nmix,
                         float* sample, void* stream);
/* Gaussian head — lib/likelihoods.py:81-114 + log_normal :391-411. params [N,P,2C] (P = H*W; mean = channels [0,C),
 * logvar = [C,2C)), x / eps / sample [N,P,C]. sample = mean + exp(logvar/2)*eps; ll[n] = sum -0.5((x-mean)^2/var + logvar + log 2pi).
 * dll_dparams [N,P,2C] optional. */
int lvae_gaussian_fwd_f32(const float* params, const float* x, const float* eps, int32_t N, int64_t P, int32_t C,
                          float* sample, float* ll, float* dll_dparams, void* stream);
/* Discretized logistic head (256 bins) — lib/likelihoods.py:117-180 + log_discretized_logistic :233-288 and the sampler
 * logistic_rsample lib/stochastic.py:115-138. raw [N,P,2C] conv output; mean = raw_mean + 0.5, logscale = max(raw_ls - 1, -7)
 * (both [N,P,C], optional outputs); u uniforms in (1e-7, 1-1e-7); x is rescaled by 255/256 + 1/512 inside (:172).
 * dll_draw [N,P,2C] optional: gradient w.r.t. the RAW conv output. */
int lvae_discr_logistic_fwd_f32(const float* raw, const float* x, const float* u, int32_t N, int64_t P, int32_t C, float* mean,
                                float* logscale, float* sample, float* ll, float* dll_draw, void* stream);
/* out[n, i] = g[n] * a[n, i]  (chain rule through a per-sample scalar; backward of the two heads above) */
int lvae_scale_per_sample_f32(const float* a, const float* g, int32_t N, int64_t P, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Resampling / geometry
 * ---------------------------------------------------------------------------------------------------------- */
/* bilinear x2, align_corners=False — boilr.nn.Interpolate(scale=2) at models/lvae.py:143-144 (restated, unpinned) */
int lvae_upsample2x_fwd_f32(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, float* y, void* stream);
int lvae_upsample2x_bwd_f32(const float* dy, int32_t N, int32_t H, int32_t W, int32_t C, float* dx, void* stream);
/* centred zero pad (OH>=H) or centre crop (OH<=H) — boilr pad_img_tensor / crop_img_tensor, models/lvae.py:185,324.
 * Also converts layout: src_nchw / dst_nchw select NCHW vs NHWC on either side. */
int lvae_pad_crop_f32(const float* x, int32_t N, int32_t C, int32_t H, int32_t W, int32_t src_nchw, float* y, int32_t OH,
                      int32_t OW, int32_t dst_nchw, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * KL bookkeeping + free bits — models/lvae.py:192-198 with boilr.nn.free_bits_kl (restated, parity unpinned):
 *   kl [L,N] (layer-major) -> kl_sep[n] = sum_l, kl_avg_layerwise[l] = mean_n,
 *   scalars[0] = kl_loss = sum_l mean_n max(kl, free_bits)   (plain mean when free_bits < 1e-6)
 *   scalars[1] = kl      = mean_n kl_sep
 * bwd: dkl from upstream g_sep [N], g_avg [L], g_scalars [2] (each may be NULL).
 * ---------------------------------------------------------------------------------------------------------- */
int lvae_kl_bookkeeping_fwd_f32(const float* kl, int32_t L, int32_t N, float free_bits, float* kl_sep,
                                float* kl_avg_layerwise, float* scalars, void* stream);
int lvae_kl_bookkeeping_bwd_f32(const float* kl, int32_t L, int32_t N, float free_bits, const float* g_sep,
                                const float* g_avg, const float* g_scalars, float* dkl, void* stream);
/* ELBO / loss assembly — experiment/experiment_manager.py:329-344:
 *   elbo_sep[n] = ll[n] - kl_sep[n]; scalars[0] = loss = mean(-ll) + beta*kl_loss; [1] = elbo; [2] = recons
 * bwd (of `loss` only; elbo/recons are metrics): d_ll[n] = -g/N, d_kl_loss = g*beta, g = g_loss[0] on device. */
int lvae_elbo_loss_fwd_f32(const float* ll, const float* kl_sep, const float* kl_loss, float beta, int32_t N,
                           float* elbo_sep, float* scalars, void* stream);
int lvae_elbo_loss_bwd_f32(const float* g_loss, float beta, int32_t N, float* d_ll, float* d_kl_loss, void* stream);

/* Importance-weighted bound — evaluate.py:30,86-87 (the loop is boilr's test_procedure: S forward passes, then
 * logsumexp - log S). elbo [S,N] (sample-major) -> out[n] = log mean_s exp(elbo[s][n]). */
int lvae_iw_logmeanexp_f32(const float* elbo, int32_t S, int32_t N, float* out, void* stream);
/* The same bound accumulated one sample at a time (so that a captured top-down + likelihood graph can be replayed S times):
 * state [3][N] = running max, sum of exp(elbo - max), plain sum. mode 0: initialise; mode 1: fold in elbo [N]; mode 2: iw[n] = max +
 * log(sumexp) - log S and mean[n] = sum / S. */
int lvae_iw_online_f32(const float* elbo, float* state, int32_t N, int32_t mode, int32_t S, float* iw, float* mean, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Optimiser and norms over the flat parameter arena — torch.optim.Adamax at experiment_manager.py:76-81 and the
 * L2 loop at experiment_manager.py:346-350.
 * ---------------------------------------------------------------------------------------------------------- */
/* step_count: device uint64[1] holding the number of COMPLETED steps (advance it with lvae_counter_advance after the
 * update; both are plain launches, so a captured graph replays them). mask [n] (optional, 0/1): elements with 0 are frozen (parameters that do not require grad). gscale: device float
 * pointer (optional) multiplied into the gradient (1/world_size after a sum all-reduce). */
int lvae_adamax_step_f32(float* p, const float* g, float* exp_avg, float* exp_inf, const float* mask, int64_t n,
                         float lr, float beta1, float beta2, float eps, float weight_decay, const float* gscale,
                         const uint64_t* step_count, void* stream);
/* out[0] = sqrt(sum x^2) ; workspace >= lvae_sumsq_workspace(n) bytes; deterministic two-pass */
size_t lvae_sumsq_workspace(int64_t n);
int lvae_l2norm_f32(const float* x, int64_t n, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * On-device noise (Philox4x32-10). `seed` is a host value; `offset` a device uint64[1] step counter advanced
 * with lvae_counter_advance (a plain launch, so graph replays draw fresh numbers); `stream_id` names the call site.
 * kind: 0 = standard normal, 1 = uniform(lo,hi), 2 = Bernoulli(p=lo) keep-mask scaled by `hi` (Dropout2d: hi=1/(1-p))
 * ---------------------------------------------------------------------------------------------------------- */
int lvae_rng_fill_f32(float* out, int64_t n, int32_t kind, float lo, float hi, uint64_t seed, const uint64_t* offset,
                      uint64_t stream_id, void* stream);
int lvae_counter_advance(uint64_t* counter, uint64_t by, void* stream);
/* out[0:n] = value (16-byte aligned out). replaces: optimizer.zero_grad() of boilr's training loop on the flat gradient arena. */
int lvae_fill_f32(float* out, int64_t n, float value, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Gradient exchange of the data-parallel step (the reference itself is single-process): SUM all-reduce of slices of
 * the flat fp32 gradient arena over the ranks of one node through a communicator of our own on librccl (RCCL over xGMI), each bucket on a
 * SIDE stream so that it overlaps the backward kernels still being issued. All calls enqueue work and return; nothing synchronises.
 * Inside a hipGraph capture the fork / join become graph edges and RCCL's kernels graph nodes.
 *   lvae_allreduce_unique_id  rank 0 only: a 128-byte communicator id, to be handed to every rank by the caller (e.g. an eager broadcast)
 *   lvae_allreduce_init       every rank (collective: ncclCommInitRank), with this rank's GPU current; *handle owns the communicator and
 *                             two events
 *   lvae_allreduce_enqueue    buf[0:n] = sum over ranks, enqueued on side_stream behind everything issued on launch_stream so far.
 *                             scratch (n floats) non-NULL: out of place into scratch + copy back — for one-rank rehearsals, where the
 *                             in-place form enqueues nothing; NULL: in place
 *   lvae_allreduce_wait       launch_stream waits for everything enqueued on side_stream so far (call before the optimizer step)
 *   lvae_allreduce_destroy    releases the communicator and the events (NULL is a no-op)
 * librccl_path: the librccl.so of the process (torch ships one: <torch>/lib/librccl.so). Errors: LVAE_E*, a hipError_t, or 1000 + the
 * ncclResult_t; lvae_last_error() has the text.
 * ---------------------------------------------------------------------------------------------------------- */
int lvae_allreduce_unique_id(const char* librccl_path, void* id128);
int lvae_allreduce_init(const char* librccl_path, const void* id128, int32_t world, int32_t rank, void** handle);
int lvae_allreduce_enqueue(void* handle, float* buf, int64_t n, float* scratch, void* launch_stream, void* side_stream);
int lvae_allreduce_wait(void* handle, void* launch_stream, void* side_stream);
int lvae_allreduce_destroy(void* handle);

#ifdef __cplusplus
}
#endif
#endif /* LVAE_HIP_H */
