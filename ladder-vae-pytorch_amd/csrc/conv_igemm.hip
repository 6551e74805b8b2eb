// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32, exact fp32).
//
//   GEMM view:  C[m][n] = sum_{tap,k} A[m][(tap,k)] * B[(tap,k)][n]
//     m = (image, oh, ow) output pixel, n = output channel, k = input channel
//     A is gathered from the NHWC input (zero outside the image) with an optional fused per-channel
//     affine + activation (BatchNorm-apply + ELU of the consumer's input) and an optional second source
//     tensor (channel concat of MergeLayer without materialising the cat).
//     B is the weight tensor addressed through (tap, k, n) strides, so the same kernel serves forward
//     (n contiguous), dgrad (k contiguous: same memory, roles of ci/co swapped) and odd shapes (scalar path).
//   Epilogue: + bias, * per-(image, channel) scale (Dropout2d), activation.
//
// One workgroup = 256 threads = 4 waves as 2(M) x 2(N); each wave owns (BM/2) x (BN/2) of the tile as 32x32
// MFMA accumulators. K is consumed in stages of KC = 32 channels of one tap; A and B stages are double
// buffered in LDS with k-contiguous rows padded to 36 floats so that the ds_read_b128 fragment reads of 32
// consecutive rows are bank-conflict free; global loads of stage s+1 are issued before the MFMAs of stage s.
#include <stdlib.h>

#include "lvae_common.h"

namespace lvae {

constexpr int KC = 32;
constexpr int LDK = KC + 4;

enum { B_NCONTIG = 0, B_KCONTIG = 1, B_SCALAR = 2 };

struct ConvArgs {
  lvae_conv_desc d;
  int M;        // N*OH*OW
  int ohw;      // OH*OW
  int Cin;      // C1 + C2
  int nchunks;  // K stages
  int cpt;      // stages per tap (vector-A path)
  int ktot;     // KH*KW*Cin (flattened-K path)
  int ntn;      // tiles along N
};

__device__ __forceinline__ bool tap_coord(int o, int k, int stride, int pad, int limit, int gather, int& i) {
  if (gather == LVAE_GATHER_CONV) {
    i = o * stride - pad + k;
    return i >= 0 && i < limit;
  }
  int t = o + pad - k;
  if (t < 0) return false;
  i = t / stride;
  return (t - i * stride == 0) && i < limit;
}

template <int BM, int BN, bool A_VEC, int B_MODE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  kernarg_warmup<(sizeof(ConvArgs) < 1024 ? sizeof(ConvArgs) : 1024)>();
  constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 32, NI = WN / 32;
  constexpr int AP = BM / 32;          // float4 A loads per thread per stage (vector path)
  constexpr int BP = BN / 32;          // float4 B loads per thread per stage (vector paths)
  constexpr int AS = BM * KC / 256;    // scalar A loads per thread
  constexpr int BS = BN * KC / 256;    // scalar B loads per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [2][BM][LDK]
  float* Bs = smem + 2 * BM * LDK;        // [2][BN][LDK]

  const lvae_conv_desc& d = a.d;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  // XCD-aware tile order: consecutive tiles (which share halo rows and all weights) stay on one XCD's L2
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_n = bid % a.ntn, tile_m = bid / a.ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- per-thread A row bookkeeping (vector path): rows arow + 32*p
  const int arow = t >> 3, ac4 = (t & 7) * 4;
  int a_n[AP], a_oh[AP], a_ow[AP];
  if (A_VEC) {
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      int m = m0 + arow + 32 * p;
      if (m < a.M) {
        int n = m / a.ohw, rem = m - n * a.ohw;
        a_n[p] = n;
        a_oh[p] = rem / d.OW;
        a_ow[p] = rem - a_oh[p] * d.OW;
      } else {
        a_n[p] = -1;
        a_oh[p] = a_ow[p] = 0;
      }
    }
  }

  f32x4 a_reg[A_VEC ? AP : 1];
  float a_sreg[A_VEC ? 1 : AS];
  f32x4 b_reg[B_MODE != B_SCALAR ? BP : 1];
  float b_sreg[B_MODE == B_SCALAR ? BS : 1];

  auto load_stage = [&](int chunk) {
    // ---------------- A ----------------
    if (A_VEC) {
      const int tap = chunk / a.cpt, ci = (chunk - tap * a.cpt) * KC + ac4;
      const int kh = tap / d.KW, kw = tap - kh * d.KW;
      const bool c_ok = ci < a.Cin;
      const float* src = d.x;
      int cs = ci, cstride = d.C1;
      if (c_ok && ci >= d.C1) {
        src = d.x2;
        cs = ci - d.C1;
        cstride = d.C2;
      }
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (d.in_scale && c_ok) {
        sc = *reinterpret_cast<const f32x4*>(d.in_scale + ci);
        sh = *reinterpret_cast<const f32x4*>(d.in_shift + ci);
      }
#pragma unroll
      for (int p = 0; p < AP; ++p) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        int ih, iw;
        if (c_ok && a_n[p] >= 0 && tap_coord(a_oh[p], kh, d.stride, d.pad, d.H, d.gather, ih) &&
            tap_coord(a_ow[p], kw, d.stride, d.pad, d.W, d.gather, iw)) {
          size_t off = ((size_t)(a_n[p] * d.H + ih) * d.W + iw) * cstride + cs;
          v = *reinterpret_cast<const f32x4*>(src + off);
          if (d.in_scale) {
            v = v * sc + sh;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = act_fwd(v[j], d.in_act);
          }
        }
        a_reg[p] = v;
      }
    } else {
      // flattened K: this thread always holds column kk = t & 31 of the stage
      const int kf = chunk * KC + (t & 31);
      const bool k_ok = kf < a.ktot;
      int tap = 0, ci = 0;
      if (k_ok) {
        tap = kf / a.Cin;
        ci = kf - tap * a.Cin;
      }
      const int kh = tap / d.KW, kw = tap - kh * d.KW;
      const float* src = d.x;
      int cs = ci, cstride = d.C1;
      if (ci >= d.C1) {
        src = d.x2;
        cs = ci - d.C1;
        cstride = d.C2;
      }
      float sc = 1.f, sh = 0.f;
      if (d.in_scale && k_ok) {
        sc = d.in_scale[ci];
        sh = d.in_shift[ci];
      }
#pragma unroll
      for (int e = 0; e < AS; ++e) {
        const int m = m0 + (t >> 5) + 8 * e;
        float v = 0.f;
        if (k_ok && m < a.M) {
          int n = m / a.ohw, rem = m - n * a.ohw;
          int oh = rem / d.OW, ow = rem - oh * d.OW, ih, iw;
          if (tap_coord(oh, kh, d.stride, d.pad, d.H, d.gather, ih) &&
              tap_coord(ow, kw, d.stride, d.pad, d.W, d.gather, iw)) {
            v = src[((size_t)(n * d.H + ih) * d.W + iw) * cstride + cs];
            if (d.in_scale) v = act_fwd(v * sc + sh, d.in_act);
          }
        }
        a_sreg[e] = v;
      }
    }
    // ---------------- B ----------------
    if (B_MODE == B_NCONTIG) {
      // float4 along n; BN/4 float4 per k row
      constexpr int F4 = BN / 4;
      const int tap = chunk / a.cpt, k0 = (chunk - tap * a.cpt) * KC;
#pragma unroll
      for (int p = 0; p < BP; ++p) {
        const int idx = t + 256 * p, kk = idx / F4, n = n0 + (idx - kk * F4) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k0 + kk < a.Cin && n < d.Cout)
          v = *reinterpret_cast<const f32x4*>(d.w + tap * d.w_stap + (int64_t)(k0 + kk) * d.w_sk + n);
        b_reg[p] = v;
      }
    } else if (B_MODE == B_KCONTIG) {
      const int tap = chunk / a.cpt, k0 = (chunk - tap * a.cpt) * KC;
#pragma unroll
      for (int p = 0; p < BP; ++p) {
        const int n = n0 + (t >> 3) + 32 * p, k = k0 + (t & 7) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < a.Cin && n < d.Cout)
          v = *reinterpret_cast<const f32x4*>(d.w + tap * d.w_stap + (int64_t)n * d.w_sn + k);
        b_reg[p] = v;
      }
    } else {
      // scalar: element idx -> (n = idx / KC, kk = idx % KC); kk = t & 31 is fixed per thread
      int tap, k;
      bool k_ok;
      if (A_VEC) {
        tap = chunk / a.cpt;
        k = (chunk - tap * a.cpt) * KC + (t & 31);
        k_ok = k < a.Cin;
      } else {
        const int kf = chunk * KC + (t & 31);
        k_ok = kf < a.ktot;
        tap = k_ok ? kf / a.Cin : 0;
        k = kf - tap * a.Cin;
      }
#pragma unroll
      for (int e = 0; e < BS; ++e) {
        const int n = n0 + (t >> 5) + 8 * e;
        b_sreg[e] = (k_ok && n < d.Cout) ? d.w[tap * d.w_stap + (int64_t)k * d.w_sk + (int64_t)n * d.w_sn] : 0.f;
      }
    }
  };

  auto store_stage = [&](int buf) {
    float* Ab = As + buf * BM * LDK;
    float* Bb = Bs + buf * BN * LDK;
    if (A_VEC) {
#pragma unroll
      for (int p = 0; p < AP; ++p) *reinterpret_cast<f32x4*>(Ab + (arow + 32 * p) * LDK + ac4) = a_reg[p];
    } else {
#pragma unroll
      for (int e = 0; e < AS; ++e) Ab[((t >> 5) + 8 * e) * LDK + (t & 31)] = a_sreg[e];
    }
    if (B_MODE == B_NCONTIG) {
      constexpr int F4 = BN / 4;
#pragma unroll
      for (int p = 0; p < BP; ++p) {
        const int idx = t + 256 * p, kk = idx / F4, nl = (idx - kk * F4) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) Bb[(nl + j) * LDK + kk] = b_reg[p][j];
      }
    } else if (B_MODE == B_KCONTIG) {
#pragma unroll
      for (int p = 0; p < BP; ++p)
        *reinterpret_cast<f32x4*>(Bb + ((t >> 3) + 32 * p) * LDK + (t & 7) * 4) = b_reg[p];
    } else {
#pragma unroll
      for (int e = 0; e < BS; ++e) Bb[((t >> 5) + 8 * e) * LDK + (t & 31)] = b_sreg[e];
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  load_stage(0);
  store_stage(0);
  __syncthreads();

  for (int c = 0; c < a.nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < a.nchunks) load_stage(c + 1);
    const float* Ab = As + buf * BM * LDK + (wm * WM + li) * LDK + 4 * lh;
    const float* Bb = Bs + buf * BN * LDK + (wn * WN + li) * LDK + 4 * lh;
#pragma unroll
    for (int kk = 0; kk < KC; kk += 8) {
      f32x4 af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const f32x4*>(Ab + mi * 32 * LDK + kk);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bf[ni] = *reinterpret_cast<const f32x4*>(Bb + ni * 32 * LDK + kk);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
    }
    if (c + 1 < a.nchunks) store_stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = n0 + wn * WN + ni * 32 + li;
    if (col >= d.Cout) continue;
    const float bias = d.bias ? d.bias[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= a.M) continue;
        float v = acc[mi][ni][r] + bias;
        if (d.out_scale) v *= d.out_scale[(size_t)(row / a.ohw) * d.Cout + col];
        v = act_fwd(v, d.out_act);
        d.y[(size_t)row * d.Cout + col] = v;
      }
    }
  }
}

template <int BM, int BN, bool A_VEC, int B_MODE>
static int launch_conv(const ConvArgs& a, hipStream_t s) {
  constexpr size_t smem = (size_t)(2 * BM * LDK + 2 * BN * LDK) * sizeof(float);
  static std::atomic<bool> attr_set{false};  // idempotent attribute write; the flag itself is race-free
  auto kern = conv_igemm_kernel<BM, BN, A_VEC, B_MODE>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)smem);
    if (e != hipSuccess) {
      set_error("conv2d: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  const int ntm = (a.M + BM - 1) / BM;
  ConvArgs b = a;
  b.ntn = (a.d.Cout + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3(ntm * b.ntn), dim3(256), smem, s, b);
  LVAE_LAUNCH_CHECK("conv2d");
  return 0;
}

template <bool A_VEC, int B_MODE>
static int dispatch_tile(const ConvArgs& a, hipStream_t s) {
  const bool wide = a.d.Cout > 64;
  const int ntn128 = wide ? (a.d.Cout + 127) / 128 : 1;
  const bool big = (int64_t)((a.M + 127) / 128) * ntn128 >= 256;
  if (wide) return big ? launch_conv<128, 128, A_VEC, B_MODE>(a, s) : launch_conv<64, 128, A_VEC, B_MODE>(a, s);
  return big ? launch_conv<128, 64, A_VEC, B_MODE>(a, s) : launch_conv<64, 64, A_VEC, B_MODE>(a, s);
}

int conv_desc_check(const lvae_conv_desc* d, const char* who) {
  LVAE_REQUIRE(d != nullptr, LVAE_EINVAL, "%s: null descriptor", who);
  LVAE_REQUIRE(d->x && d->w, LVAE_EINVAL, "%s: null x or w", who);
  LVAE_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->OH > 0 && d->OW > 0 && d->Cout > 0 && d->C1 > 0 && d->C2 >= 0,
               LVAE_EINVAL, "%s: non-positive dimension", who);
  LVAE_REQUIRE((d->C2 == 0) == (d->x2 == nullptr), LVAE_EINVAL, "%s: x2/C2 mismatch", who);
  LVAE_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0, LVAE_EINVAL, "%s: bad kernel geometry", who);
  LVAE_REQUIRE(d->gather == LVAE_GATHER_CONV || d->gather == LVAE_GATHER_TRANSPOSED, LVAE_EINVAL, "%s: bad gather", who);
  LVAE_REQUIRE((d->in_scale == nullptr) || (d->in_shift != nullptr), LVAE_EINVAL, "%s: in_scale without in_shift", who);
  LVAE_REQUIRE((int64_t)d->N * d->OH * d->OW < (int64_t)1 << 31 && (int64_t)d->N * d->H * d->W < (int64_t)1 << 31,
               LVAE_EINVAL, "%s: pixel count exceeds int32", who);
  if (d->gather == LVAE_GATHER_CONV) {
    LVAE_REQUIRE((d->H + 2 * d->pad - d->KH) / d->stride + 1 == d->OH && (d->W + 2 * d->pad - d->KW) / d->stride + 1 == d->OW,
                 LVAE_EINVAL, "%s: output size %dx%d inconsistent with input %dx%d k%d s%d p%d", who, d->OH, d->OW, d->H,
                 d->W, d->KH, d->stride, d->pad);
  } else {
    // every output pixel must map inside the input: (OH-1 + pad - 0)/stride <= H-1 is guaranteed by bounds checks;
    // require the forward-conv relation of the transposed view to hold so that no input pixel is dropped
    LVAE_REQUIRE((d->OH + 2 * d->pad - d->KH) / d->stride + 1 == d->H && (d->OW + 2 * d->pad - d->KW) / d->stride + 1 == d->W,
                 LVAE_EINVAL, "%s: transposed geometry inconsistent (in %dx%d out %dx%d k%d s%d p%d)", who, d->H, d->W,
                 d->OH, d->OW, d->KH, d->stride, d->pad);
  }
  return 0;
}

int conv3x3_bf16_try(const lvae_conv_desc* d, int split, hipStream_t s);
int conv3x3_bf16_stats_rows(const lvae_conv_desc* d, int split);
int conv3x3_bf16_form(const lvae_conv_desc* d);
size_t conv3x3_bf16_workspace(const lvae_conv_desc* d, int split);
int conv3x3_pos_try(const lvae_conv_desc* d, hipStream_t s);
int conv3x3_pos_stats_rows(const lvae_conv_desc* d);
bool conv3x3_pos_eligible(const lvae_conv_desc* d);
int conv3x3_halo_try(const lvae_conv_desc* d, hipStream_t s);
int conv3x3_halo_stats_rows(const lvae_conv_desc* d);
int conv3x3_wino_stats_rows(const lvae_conv_desc* d);
int conv3x3_wino_try(const lvae_conv_desc* d, void* workspace, size_t workspace_bytes, hipStream_t s);
size_t conv3x3_wino_workspace(const lvae_conv_desc* d);
bool conv3x3_wino_eligible(const lvae_conv_desc* d);
int conv1x1_try(const lvae_conv_desc* d, const float* gate_res, float* gate_out, int gate_act, hipStream_t s);
int conv1x1_gate_fwd_wgs(const lvae_conv_desc* d);
int conv1x1_gate_fwd_try(const lvae_conv_desc* d, const float* res, float* out, int act, hipStream_t s);
int conv1x1_try_ex(const lvae_conv_desc* d, const float* gate_res, float* gate_out, int gate_act, const float* gb_dout,
                   const float* gb_ab, float* gb_dab, int gb_act, hipStream_t s);

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace lvae

using namespace lvae;

extern "C" size_t lvae_conv2d_workspace(const lvae_conv_desc* d) {
  if (d == nullptr) return 0;
  if (conv3x3_pos_eligible(d)) return 0;
  const int form = conv3x3_bf16_form(d);
  if (form != 0) return conv3x3_bf16_workspace(d, form);  // pre-split bf16 weight planes
  return conv3x3_wino_eligible(d) ? conv3x3_wino_workspace(d) : 0;
}

namespace lvae {
int conv3x3_wino_variant(const lvae_conv_desc* d);
bool conv3x3_wino_folds(const lvae_conv_desc* d);
}

extern "C" int32_t lvae_conv2d_variant(const lvae_conv_desc* d) {
  if (d == nullptr || tune("LVAE_DISABLE_HALO", 0) != 0) return LVAE_VARIANT_DIRECT;
  if (conv3x3_pos_eligible(d)) return LVAE_VARIANT_POS;
  const int form = conv3x3_bf16_form(d);
  if (form != 0 && d->workspace != nullptr && (size_t)d->workspace_bytes >= conv3x3_bf16_workspace(d, form))
    return form == 1 ? LVAE_VARIANT_BF16_DIRECT : LVAE_VARIANT_SIX_DIRECT;
  const int w = conv3x3_wino_variant(d);
  return w ? w : LVAE_VARIANT_DIRECT;
}

namespace lvae {
size_t conv3x3_wgrad_bf16_workspace(const lvae_conv_desc* d);
size_t conv1x1_gate_bwd_fused_workspace(const lvae_conv_desc* d);
}

extern "C" int32_t lvae_resblock_bf16_storage(const lvae_conv_desc* d) {
  if (d == nullptr || d->precision != LVAE_PREC_BF16 || d->C1 != 64 || d->C2 != 0 || d->Cout != 64 || d->gather != LVAE_GATHER_CONV) return 0;
  if (lvae_conv2d_variant(d) != LVAE_VARIANT_BF16_DIRECT) return 0;
  // the block's BatchNorm statistics / BatchNorm-backward sums must come out of the convolutions' own epilogues: the stand-alone
  // statistics kernels have no bf16-storage form
  if (lvae_conv2d_stats_rows(d) <= 0) return 0;
  lvae_conv_desc t = *d;  // dgrad view of the same layer (same shape; only the gather differs for the kernel choice)
  t.gather = LVAE_GATHER_TRANSPOSED;
  if (lvae_conv2d_variant(&t) != LVAE_VARIANT_BF16_DIRECT) return 0;
  if (lvae_conv2d_stats_rows(&t) <= 0) return 0;
  if (conv3x3_wgrad_bf16_workspace(d) == 0) return 0;
  lvae_conv_desc g = *d;  // the block's GateLayer2d: 1x1, 64 -> 128 forward; its fused backward is described by the 128 -> 64 dgrad view
  g.KH = g.KW = 1; g.pad = 0; g.Cout = 128; g.in_scale = g.in_shift = nullptr; g.out_scale = nullptr; g.in_act = g.out_act = 0; g.in_fold = nullptr;
  if (conv1x1_gate_fwd_wgs(&g) == 0) return 0;
  g.C1 = 128; g.Cout = 64; g.w_sk = 1; g.w_sn = 128;
  if (conv1x1_gate_bwd_fused_workspace(&g) == 0) return 0;
  return 1;
}

extern "C" int32_t lvae_conv2d_stats_rows(const lvae_conv_desc* d) {
  if (d == nullptr || tune("LVAE_DISABLE_HALO", 0) != 0) return 0;
  const int p = conv3x3_pos_stats_rows(d);
  if (p > 0) return p;
  const int form = conv3x3_bf16_form(d);
  if (form != 0) return conv3x3_bf16_stats_rows(d, form);
  const int w = conv3x3_wino_stats_rows(d);
  if (w > 0) return w;
  if (d->workspace != nullptr && conv3x3_wino_eligible(d) && (size_t)d->workspace_bytes >= conv3x3_wino_workspace(d)) return 0;
  return conv3x3_halo_stats_rows(d);
}

extern "C" int32_t lvae_conv2d_folds_bn_finalize(const lvae_conv_desc* d) {
  if (d == nullptr || tune("LVAE_DISABLE_HALO", 0) != 0) return 0;
  if (conv3x3_pos_eligible(d)) return 1;
  const int v = lvae_conv2d_variant(d);
  return (v == LVAE_VARIANT_WINO_F32 || v == LVAE_VARIANT_WINO_SIX) && conv3x3_wino_folds(d) ? 1 : 0;
}

extern "C" int32_t lvae_conv2d_stats_buffer_rows(const lvae_conv_desc* d) {
  if (d == nullptr) return 0;
  const int32_t rows = lvae_conv2d_stats_rows(d);
  if (rows <= 0) return 0;
  if (d->stats_mode != LVAE_STATS_BN_FWD) return rows;  // BatchNorm-backward sums have no pivot
  lvae_conv_desc t = *d;  // the pivot row depends on the kernel choice, which is made on the descriptor without its fold
  t.in_fold = nullptr;
  return rows + (lvae_conv2d_folds_bn_finalize(&t) != 0 ? 1 : 0);
}

extern "C" int lvae_conv2d_f32(const lvae_conv_desc* d, void* stream) {
  int rc = conv_desc_check(d, "lvae_conv2d_f32");
  if (rc) return rc;
  LVAE_REQUIRE(d->y != nullptr, LVAE_EINVAL, "lvae_conv2d_f32: null y");
  LVAE_REQUIRE(d->stats_out == nullptr || (d->stats_pivot != nullptr && lvae_conv2d_stats_rows(d) > 0), LVAE_EINVAL,
               "lvae_conv2d_f32: stats_out set but lvae_conv2d_stats_rows(d) == 0 (this kernel variant has no statistics epilogue)");
  LVAE_REQUIRE(d->stats_out == nullptr || d->stats_mode == LVAE_STATS_BN_FWD ||
                   (d->stats_mode == LVAE_STATS_BN_BWD && d->stats_x != nullptr && (reinterpret_cast<uintptr_t>(d->stats_x) & 15) == 0),
               LVAE_EINVAL, "lvae_conv2d_f32: bad stats_mode / stats_x");
  static const bool halo_off = tune("LVAE_DISABLE_HALO", 0) != 0;  // A/B switch (tuning builds only)
  LVAE_REQUIRE((d->x_dtype == LVAE_DT_F32 && d->y_dtype == LVAE_DT_F32 && d->stats_x_dtype == LVAE_DT_F32) ||
                   lvae_conv2d_variant(d) == LVAE_VARIANT_BF16_DIRECT,
               LVAE_EINVAL, "lvae_conv2d_f32: bf16-stored tensors need the bf16 3x3 kernel (precision LVAE_PREC_BF16, lvae_conv2d_variant(d) == "
                            "LVAE_VARIANT_BF16_DIRECT)");
  if (d->in_fold != nullptr) {
    lvae_conv_desc t = *d;  // the kernel choice is made on the descriptor without its fold (as the caller asked lvae_conv2d_folds_bn_finalize)
    t.in_fold = nullptr;
    LVAE_REQUIRE(lvae_conv2d_folds_bn_finalize(&t) != 0 && d->in_fold->parts != nullptr &&
                     (reinterpret_cast<uintptr_t>(d->in_fold->parts) & 15) == 0 && d->C1 % 4 == 0 && d->in_fold->rows > 0 &&
                     d->in_fold->M > 0 && d->in_scale == nullptr,
                 LVAE_EINVAL,
                 "lvae_conv2d_f32: in_fold set but lvae_conv2d_folds_bn_finalize(d) == 0 (or bad parts / rows / M, or in_scale given too)");
  }
  if (!halo_off) {
    int hr = conv3x3_pos_try(d, (hipStream_t)stream);
    if (hr != -1000) return hr;
    const int form = conv3x3_bf16_form(d);
    if (form != 0) {
      hr = conv3x3_bf16_try(d, form, (hipStream_t)stream);
      if (hr != -1000) return hr;
    }
    hr = conv3x3_wino_try(d, d->workspace, (size_t)d->workspace_bytes, (hipStream_t)stream);
    if (hr != -1000) return hr;
    LVAE_REQUIRE(d->in_fold == nullptr, LVAE_EINVAL, "lvae_conv2d_f32: no kernel took the folded BatchNorm finalize (in_fold)");
    hr = conv3x3_halo_try(d, (hipStream_t)stream);
    if (hr != -1000) return hr;
    hr = conv1x1_try(d, nullptr, nullptr, 0, (hipStream_t)stream);
    if (hr != -1000) return hr;
  }
  ConvArgs a;
  a.d = *d;
  a.M = d->N * d->OH * d->OW;
  a.ohw = d->OH * d->OW;
  a.Cin = d->C1 + d->C2;
  a.ktot = d->KH * d->KW * a.Cin;
  a.ntn = 1;
  const bool a_vec = (d->C1 % 4 == 0) && (d->C2 % 4 == 0) && aligned16(d->x) && (!d->x2 || aligned16(d->x2)) &&
                     (!d->in_scale || (aligned16(d->in_scale) && aligned16(d->in_shift)));
  if (a_vec) {
    a.cpt = (a.Cin + KC - 1) / KC;
    a.nchunks = d->KH * d->KW * a.cpt;
  } else {
    a.cpt = 1;
    a.nchunks = (a.ktot + KC - 1) / KC;
  }
  hipStream_t s = (hipStream_t)stream;
  const bool w16 = aligned16(d->w) && (d->w_stap % 4 == 0);
  if (!a_vec) return dispatch_tile<false, B_SCALAR>(a, s);
  if (d->w_sn == 1 && w16 && d->Cout % 4 == 0 && d->w_sk % 4 == 0) return dispatch_tile<true, B_NCONTIG>(a, s);
  if (d->w_sk == 1 && w16 && a.Cin % 4 == 0 && d->w_sn % 4 == 0) return dispatch_tile<true, B_KCONTIG>(a, s);
  return dispatch_tile<true, B_SCALAR>(a, s);
}

extern "C" int lvae_conv2d_bf16(const lvae_conv_desc* d, void* stream) {
  LVAE_REQUIRE(d != nullptr, LVAE_EINVAL, "lvae_conv2d_bf16: null descriptor");
  lvae_conv_desc dd = *d;
  dd.precision = LVAE_PREC_BF16;
  return lvae_conv2d_f32(&dd, stream);
}

// GateLayer2d forward fused with its 1x1 convolution and the residual add (lib/nn.py:118-126, 99):
//   ab = conv1x1(T(x)) + bias  (written to d->y when non-null; needed by the backward)
//   out[m, c] = act(ab[m, c]) * sigmoid(ab[m, C + c]) + res[m, c]
// rows of BatchNorm partials ([rows][2][C], C = Cout/2) lvae_conv1x1_gate_f32 writes for its `out` when d->stats_out is set
extern "C" int32_t lvae_conv1x1_gate_stats_rows(const lvae_conv_desc* d) {
  if (d == nullptr || d->Cout % 8 != 0) return 0;
  const int persistent = conv1x1_gate_fwd_wgs(d);  // one row per workgroup of the persistent kernel (conv1x1_gate_fwd.hip)
  if (persistent) return persistent;
  const int c4n = d->Cout / 8;
  if (c4n <= 0 || 256 % c4n != 0 || d->Cout > 128) return 0;
  return (int32_t)(((int64_t)d->N * d->H * d->W + 63) / 64);  // 64-pixel tiles
}

extern "C" int lvae_conv1x1_gate_f32(const lvae_conv_desc* d, const float* res, int32_t act, float* out, void* stream) {
  int rc = conv_desc_check(d, "lvae_conv1x1_gate_f32");
  if (rc) return rc;
  LVAE_REQUIRE(out != nullptr, LVAE_EINVAL, "lvae_conv1x1_gate_f32: null out");
  LVAE_REQUIRE(d->Cout % 2 == 0, LVAE_EINVAL, "lvae_conv1x1_gate_f32: Cout must be 2*C");
  LVAE_REQUIRE(d->stats_out == nullptr || (d->stats_pivot != nullptr && lvae_conv1x1_gate_stats_rows(d) > 0 &&
                                           (reinterpret_cast<uintptr_t>(d->stats_pivot) & 15) == 0),
               LVAE_EINVAL, "lvae_conv1x1_gate_f32: stats_out set but lvae_conv1x1_gate_stats_rows(d) == 0");
  rc = conv1x1_gate_fwd_try(d, res, out, act, (hipStream_t)stream);
  LVAE_REQUIRE(rc != -1000 || (d->x_dtype == LVAE_DT_F32 && d->y_dtype == LVAE_DT_F32), LVAE_EINVAL,
               "lvae_conv1x1_gate_f32: bf16-stored x / ab need the persistent 64-channel kernel with precision LVAE_PREC_BF16");
  if (rc == -1000) rc = conv1x1_try(d, res, out, act, (hipStream_t)stream);
  LVAE_REQUIRE(rc != -1000, LVAE_EINVAL,
               "lvae_conv1x1_gate_f32: unsupported shape (needs a 1x1 stride-1 conv, Cin <= 128, Cout <= 128, channels %% 4 == 0, "
               "16-byte aligned buffers); use lvae_conv2d_f32 + lvae_gate_fwd_f32");
  return rc;
}

// GateLayer2d backward fused with the dgrad of its 1x1 convolution: dab (the gradient w.r.t. the pre-activations ab, also
// written to `dab` for the weight-gradient call) is formed in the kernel's operand staging from dout and ab, then
// dx = dab . W^T with the descriptor's epilogue (out_scale = Dropout2d mask of the producer). `d` describes that dgrad:
// C1 = 2C, Cout = channels of the gate convolution's input, d->x is ignored.
extern "C" int lvae_conv1x1_gate_bwd_f32(const lvae_conv_desc* d, const float* dout, const float* ab, int32_t act, float* dab,
                                         void* stream) {
  LVAE_REQUIRE(d && dout && ab && d->y, LVAE_EINVAL, "lvae_conv1x1_gate_bwd_f32: null pointer");
  lvae_conv_desc dd = *d;
  dd.x = ab;  // any valid, aligned device pointer: the A operand is computed, not loaded
  int rc = conv_desc_check(&dd, "lvae_conv1x1_gate_bwd_f32");
  if (rc) return rc;
  LVAE_REQUIRE(dd.C1 % 8 == 0 && dd.C2 == 0, LVAE_EINVAL, "lvae_conv1x1_gate_bwd_f32: C1 must be 2C");
  rc = conv1x1_try_ex(&dd, nullptr, nullptr, 0, dout, ab, dab, act, (hipStream_t)stream);
  LVAE_REQUIRE(rc != -1000, LVAE_EINVAL,
               "lvae_conv1x1_gate_bwd_f32: unsupported shape (needs a 1x1 stride-1 conv, 2C <= 128, Cout <= 128, 16-byte aligned "
               "buffers); use lvae_gate_bwd_f32 + lvae_conv2d_f32");
  return rc;
}
