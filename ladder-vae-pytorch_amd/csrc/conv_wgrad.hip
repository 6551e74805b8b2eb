// Weight / bias gradient of the implicit-GEMM convolution on fp32 MFMA.
//
//   dW[tap][ci][co] = sum_{m=(n,oh,ow)} T(x)[m @ tap][ci] * dy[m][co]        db[co] = sum_m dy[m][co]
//
// GEMM view per tap: M' = ci (A = T(x)^T), N' = co (B = dy), K' = output pixels. Both operands are pixel-major
// in memory (NHWC), i.e. already "k-major": the LDS stages are [pixel][channel] and the MFMA fragments are
// ds_read_b32 of 32 consecutive channels (conflict free). The pixel range is split over `ksplit` workgroups
// per (tap, ci-tile, co-tile); each writes its 64x64 partial to a slab, and a second kernel sums the slabs in a
// fixed order (bitwise reproducible, no float atomics) and accumulates into the gradient arena.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "lvae_common.h"

namespace lvae {

int conv_desc_check(const lvae_conv_desc* d, const char* who);

constexpr int KP = 32;   // pixels per stage
constexpr int CT = 64;   // channel tile (both ci and co)

struct WgradArgs {
  lvae_conv_desc d;
  const float* dy;
  float* slab_w;   // [ksplit][taps][Cin][Cout]
  float* slab_b;   // [ksplit][Cout]
  int M, ohw, Cin, ntaps, ncit, ncot, ksplit, px_per_split;
};

__device__ __forceinline__ bool tap_coord_w(int o, int k, int stride, int pad, int limit, int gather, int& i) {
  if (gather == LVAE_GATHER_CONV) {
    i = o * stride - pad + k;
    return i >= 0 && i < limit;
  }
  int t = o + pad - k;
  if (t < 0) return false;
  i = t / stride;
  return (t - i * stride == 0) && i < limit;
}

template <bool X_VEC, bool Y_VEC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  kernarg_warmup<(sizeof(WgradArgs) < 1024 ? sizeof(WgradArgs) : 1024)>();
  __shared__ __attribute__((aligned(16))) float Xs[2][KP][CT];
  __shared__ __attribute__((aligned(16))) float Ys[2][KP][CT];
  const lvae_conv_desc& d = a.d;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wci = wave >> 1, wco = wave & 1;

  int bid = blockIdx.x;
  const int ks = bid % a.ksplit;
  bid /= a.ksplit;
  const int cot = bid % a.ncot;
  bid /= a.ncot;
  const int cit = bid % a.ncit;
  const int tap = bid / a.ncit;
  const int kh = tap / d.KW, kw = tap - kh * d.KW;
  const int ci0 = cit * CT, co0 = cot * CT;
  const int p_begin = ks * a.px_per_split;
  const int p_end = min(a.M, p_begin + a.px_per_split);
  const bool do_bias = (a.slab_b != nullptr) && tap == 0 && cit == 0;

  // vector staging: thread -> pixel row (t>>4) + 16*p, channels (t&15)*4
  const int prow = t >> 4, c4 = (t & 15) * 4;
  f32x4 xr[2], yr[2];
  float xs_[8], ys_[8];
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  float bsum_s = 0.f;

  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  float sc_s = 1.f, sh_s = 0.f;
  if (X_VEC) {
    if (d.in_scale && ci0 + c4 < a.Cin) {
      sc = *reinterpret_cast<const f32x4*>(d.in_scale + ci0 + c4);
      sh = *reinterpret_cast<const f32x4*>(d.in_shift + ci0 + c4);
    }
  } else if (d.in_scale && ci0 + (t & 63) < a.Cin) {
    sc_s = d.in_scale[ci0 + (t & 63)];
    sh_s = d.in_shift[ci0 + (t & 63)];
  }

  auto load_x_pixel = [&](int m, int& n, int& ih, int& iw) -> bool {
    if (m >= p_end) return false;
    n = m / a.ohw;
    int rem = m - n * a.ohw;
    int oh = rem / d.OW, ow = rem - oh * d.OW;
    return tap_coord_w(oh, kh, d.stride, d.pad, d.H, d.gather, ih) &&
           tap_coord_w(ow, kw, d.stride, d.pad, d.W, d.gather, iw);
  };

  auto load_stage = [&](int p0) {
    if (X_VEC) {
      const int ci = ci0 + c4;
      const float* src = d.x;
      int cs = ci, cstride = d.C1;
      if (ci >= d.C1) {
        src = d.x2;
        cs = ci - d.C1;
        cstride = d.C2;
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        int n, ih, iw;
        if (ci < a.Cin && load_x_pixel(p0 + prow + 16 * p, n, ih, iw)) {
          v = *reinterpret_cast<const f32x4*>(src + ((size_t)(n * d.H + ih) * d.W + iw) * cstride + cs);
          if (d.in_scale) {
            v = v * sc + sh;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = act_fwd(v[j], d.in_act);
          }
        }
        xr[p] = v;
      }
    } else {
      const int ci = ci0 + (t & 63);
      const float* src = d.x;
      int cs = ci, cstride = d.C1;
      if (ci >= d.C1) {
        src = d.x2;
        cs = ci - d.C1;
        cstride = d.C2;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = 0.f;
        int n, ih, iw;
        if (ci < a.Cin && load_x_pixel(p0 + (t >> 6) + 4 * e, n, ih, iw)) {
          v = src[((size_t)(n * d.H + ih) * d.W + iw) * cstride + cs];
          if (d.in_scale) v = act_fwd(v * sc_s + sh_s, d.in_act);
        }
        xs_[e] = v;
      }
    }
    if (Y_VEC) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int m = p0 + prow + 16 * p;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < p_end && co0 + c4 < d.Cout) v = *reinterpret_cast<const f32x4*>(a.dy + (size_t)m * d.Cout + co0 + c4);
        yr[p] = v;
        if (do_bias) bsum += v;
      }
    } else {
      const int co = co0 + (t & 63);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int m = p0 + (t >> 6) + 4 * e;
        float v = (m < p_end && co < d.Cout) ? a.dy[(size_t)m * d.Cout + co] : 0.f;
        ys_[e] = v;
        if (do_bias) bsum_s += v;
      }
    }
  };

  auto store_stage = [&](int buf) {
    if (X_VEC) {
#pragma unroll
      for (int p = 0; p < 2; ++p) *reinterpret_cast<f32x4*>(&Xs[buf][prow + 16 * p][c4]) = xr[p];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) Xs[buf][(t >> 6) + 4 * e][t & 63] = xs_[e];
    }
    if (Y_VEC) {
#pragma unroll
      for (int p = 0; p < 2; ++p) *reinterpret_cast<f32x4*>(&Ys[buf][prow + 16 * p][c4]) = yr[p];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) Ys[buf][(t >> 6) + 4 * e][t & 63] = ys_[e];
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int nstages = (p_end - p_begin + KP - 1) / KP;
  if (nstages > 0) {
    load_stage(p_begin);
    store_stage(0);
  }
  __syncthreads();
  for (int s = 0; s < nstages; ++s) {
    const int buf = s & 1;
    if (s + 1 < nstages) load_stage(p_begin + (s + 1) * KP);
    const float* xa = &Xs[buf][lh][wci * 32 + li];
    const float* yb = &Ys[buf][lh][wco * 32 + li];
#pragma unroll
    for (int kk = 0; kk < KP; kk += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[kk * CT], yb[kk * CT], acc, 0, 0, 0);
    if (s + 1 < nstages) store_stage(buf ^ 1);
    __syncthreads();
  }

  // partial tile -> slab [ks][tap][Cin][Cout]; C/D layout: col = lane&31 (co), row = (r&3)+8*(r>>2)+4*lh (ci)
  float* slab = a.slab_w + ((size_t)ks * a.ntaps + tap) * a.Cin * d.Cout;
  const int co = co0 + wco * 32 + li;
  if (co < d.Cout) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + wci * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (ci < a.Cin) slab[(size_t)ci * d.Cout + co] = acc[r];
    }
  }

  if (do_bias) {
    // reduce the per-thread column sums over the pixel rows of the block through LDS (reuse Ys[0])
    __syncthreads();
    float* red = &Ys[0][0][0];
    if (Y_VEC) {
#pragma unroll
      for (int j = 0; j < 4; ++j) red[prow * CT + c4 + j] = bsum[j];
      __syncthreads();
      if (t < CT) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += red[r * CT + t];
        if (co0 + t < d.Cout) a.slab_b[(size_t)ks * d.Cout + co0 + t] = s;
      }
    } else {
      red[(t >> 6) * CT + (t & 63)] = bsum_s;
      __syncthreads();
      if (t < CT) {
        float s = red[t] + red[CT + t] + red[2 * CT + t] + red[3 * CT + t];
        if (co0 + t < d.Cout) a.slab_b[(size_t)ks * d.Cout + co0 + t] = s;
      }
    }
  }
}

// dw[tap*stap + ci*sk + co*sn] += sum_ks slab[ks][tap][ci][co] ;  db[co] += sum_ks slab_b[ks][co]
// The slab rows [taps*Cin*Cout | Cout] are contiguous and a multiple of 4 floats on the vector path: 256 threads =
// 16 float4 elements x 16 split lanes (1 KB per wave load), LDS tree in a fixed order -> deterministic.
template <int V>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab_w, const float* __restrict__ slab_b,
                                                            int ksplit, int ntaps, int Cin, int Cout, int64_t stap, int64_t sk,
                                                            int64_t sn, float* dw, float* db) {
  __shared__ float red[16][16 * 4];
  const int per = ntaps * Cin * Cout;
  const int e = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int idx = (blockIdx.x * 16 + e) * V;  // first float of this thread's element group
  float s[V];
  for (int j = 0; j < V; ++j) s[j] = 0.f;
  const float* src = nullptr;
  int64_t stride = 0;
  if (idx < per) {
    src = slab_w + idx;
    stride = per;
  } else if (db && idx < per + Cout) {
    src = slab_b + (idx - per);
    stride = Cout;
  }
  if (src) {
    // 8 independent loads per round trip, added in slab order (a rolled loop pays one memory latency per slab: 7.7 us for 256 slabs)
    for (int k0 = q; k0 < ksplit; k0 += 16 * 8) {
      if (V == 4) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + 16 * u;
          v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(k < ksplit ? k : q) * stride);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + 16 * u < ksplit)
            for (int j = 0; j < 4; ++j) s[j] += v[u][j];
      } else {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + 16 * u;
          v[u] = src[(size_t)(k < ksplit ? k : q) * stride];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + 16 * u < ksplit) s[0] += v[u];
      }
    }
  }
  for (int j = 0; j < V; ++j) red[q][e * 4 + j] = s[j];
  __syncthreads();
  if (q != 0 || !src) return;
  for (int j = 0; j < V; ++j) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][e * 4 + j];
    const int i = idx + j;
    if (i < per) {
      const int co = i % Cout, r2 = i / Cout, ci = r2 % Cin, tap = r2 / Cin;
      dw[tap * stap + ci * sk + co * sn] += t;
    } else if (i < per + Cout) {
      db[i - per] += t;
    }
  }
}

void wgrad_reduce_launch(const float* slab_w, const float* slab_b, int ksplit, int ntaps, int Cin, int Cout, int64_t stap,
                         int64_t sk, int64_t sn, float* dw, float* db, hipStream_t s) {
  const int per = ntaps * Cin * Cout, tot = per + (db ? Cout : 0);
  const bool v4 = (per % 4 == 0) && (Cout % 4 == 0) && ((reinterpret_cast<uintptr_t>(slab_w) & 15) == 0) &&
                  (!slab_b || (reinterpret_cast<uintptr_t>(slab_b) & 15) == 0);
  if (v4)
    hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((tot / 4 + 15) / 16), dim3(256), 0, s, slab_w, slab_b, ksplit, ntaps, Cin,
                       Cout, stap, sk, sn, dw, db);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((tot + 15) / 16), dim3(256), 0, s, slab_w, slab_b, ksplit, ntaps, Cin,
                       Cout, stap, sk, sn, dw, db);
}

struct ReduceArgs {
  const float* slab_w;
  const float* slab_b;
  int ksplit, ntaps, Cin, Cout;
  int64_t stap, sk, sn;
  float* dw;
  float* db;
};
constexpr int kMaxReduceGroup = 12;
struct ReduceGroup {
  ReduceArgs p[kMaxReduceGroup];
};

// blockIdx.y = problem; same element mapping as wgrad_reduce_kernel<4> (all grouped problems are float4-aligned)
__global__ __launch_bounds__(256) void wgrad_reduce_grouped_kernel(ReduceGroup g) {
  kernarg_warmup<(sizeof(ReduceGroup) < 1024 ? sizeof(ReduceGroup) : 1024)>();
  __shared__ float red[16][16 * 4];
  const ReduceArgs& a = g.p[blockIdx.y];
  const int per = a.ntaps * a.Cin * a.Cout;
  const int e = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int idx = (blockIdx.x * 16 + e) * 4;
  if ((int)blockIdx.x * 64 >= per + (a.slab_b ? a.Cout : 0)) return;  // uniform per workgroup
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const float* src = nullptr;
  int64_t stride = 0;
  if (idx < per) {
    src = a.slab_w + idx;
    stride = per;
  } else if (a.db && idx < per + a.Cout) {
    src = a.slab_b + (idx - per);
    stride = a.Cout;
  }
  if (src)
    for (int k0 = q; k0 < a.ksplit; k0 += 16 * 8) {  // 8 independent loads per round trip, added in slab order
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + 16 * u;
        v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(k < a.ksplit ? k : q) * stride);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + 16 * u < a.ksplit) s += v[u];
    }
  *reinterpret_cast<f32x4*>(&red[q][e * 4]) = s;
  __syncthreads();
  if (q != 0 || !src) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][e * 4 + j];
    const int i = idx + j;
    if (i < per) {
      const int co = i % a.Cout, r2 = i / a.Cout, ci = r2 % a.Cin, tap = r2 / a.Cin;
      a.dw[tap * a.stap + ci * a.sk + co * a.sn] += t;
    } else if (i < per + a.Cout) {
      a.db[i - per] += t;
    }
  }
}

void wgrad_reduce_grouped_launch(const ReduceArgs* r, int n, hipStream_t s) {
  ReduceGroup g;
  int max_tot = 0;
  for (int i = 0; i < n; ++i) {
    g.p[i] = r[i];
    const int tot = r[i].ntaps * r[i].Cin * r[i].Cout + (r[i].db ? r[i].Cout : 0);
    if (tot > max_tot) max_tot = tot;
  }
  for (int i = n; i < kMaxReduceGroup; ++i) g.p[i] = g.p[0];
  hipLaunchKernelGGL(wgrad_reduce_grouped_kernel, dim3((max_tot / 4 + 15) / 16, n), dim3(256), 0, s, g);
}

size_t conv_wgrad_wino_workspace(const lvae_conv_desc* d);
// whole-image tiles of the <= 8x8 levels on the bf16 matrix pipe (conv_wgrad_img.hip): up to 32 gradients per launch
size_t conv_wgrad_img_workspace(const lvae_conv_desc* d);
int conv_wgrad_img_kind(const lvae_conv_desc* d);
int conv_wgrad_img_grouped(const lvae_conv_desc* const* ds, const float* const* dy, float* const* dw, float* const* db,
                           void* const* workspace, int n, int kind, hipStream_t s);
int conv_wgrad_img_try(const lvae_conv_desc* d, const float* dy, float* dw, float* db, void* workspace, hipStream_t s);
size_t conv3x3_wgrad_bf16_workspace(const lvae_conv_desc* d);
int conv3x3_wgrad_bf16_try(const lvae_conv_desc* d, const float* dy, float* dw, float* db, void* workspace, hipStream_t s);
int conv_wgrad_wino_grouped(const lvae_conv_desc* const* ds, const float* const* dy, float* const* dw, float* const* db,
                            void* const* workspace, int n, hipStream_t s);
int conv_wgrad_tile_kind(const lvae_conv_desc* d);
int conv_wgrad_tile_grouped(const lvae_conv_desc* const* ds, const float* const* dy, float* const* dw, float* const* db,
                            void* const* workspace, int n, int kind, hipStream_t s);
int conv_wgrad_wino_try(const lvae_conv_desc* d, const float* dy, float* dw, float* db, void* workspace, hipStream_t s);
bool conv_wgrad_wino_apply_ok(const lvae_conv_desc* d);
int conv_wgrad_wino_apply_try(const lvae_conv_desc* d, const lvae_bn_apply* ap, float* dw, float* db, void* workspace, hipStream_t s);

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of the stem convolutions (5x5 stride 2 on the 1- or 3-channel image): the reduction dimension of the
// implicit GEMM is wide (all pixels) but its M side is only KH*KW*Cin <= 76 rows, far too thin for the MFMA tile kernels
// (the generic kernel spends 550 us on 0.6 GFLOP). One workgroup per image: the zero-padded image (<= 32 KB) and the
// image's dy tile sit in LDS; thread = (co, one of 4 k-groups) keeps 19 accumulators in registers and walks the output
// pixels (LDS broadcast read of x, one dy value per pixel). Partials go to the usual split-K slabs [image][k][co].
// ---------------------------------------------------------------------------------------------------------
struct ThinWgradArgs {
  lvae_conv_desc d;
  const float* dy;
  float* slab_w;
  float* slab_b;
  int K, PH, PW;  // K = KH*KW*Cin; padded image height / width
};

__global__ __launch_bounds__(256) void conv_wgrad_thin_kernel(ThinWgradArgs a) {
  kernarg_warmup<(sizeof(ThinWgradArgs) < 1024 ? sizeof(ThinWgradArgs) : 1024)>();
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const lvae_conv_desc& d = a.d;
  const int t = threadIdx.x, n = blockIdx.x;
  const int Cin = d.C1, npx = d.OH * d.OW;
  float* xs = smem;                                   // [PH][PW][Cin], zero border
  float* ys = smem + ((a.PH * a.PW * Cin + 3) & ~3);  // [npx][64]
  for (int i = t; i < a.PH * a.PW * Cin; i += 256) {
    const int ci = i % Cin, r = i / Cin, pw = r % a.PW, ph = r / a.PW;
    const int ih = ph - d.pad, iw = pw - d.pad;
    xs[i] = ((unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W) ? d.x[((size_t)(n * d.H + ih) * d.W + iw) * Cin + ci] : 0.f;
  }
  for (int i = t; i < npx * 64; i += 256) {
    const int co = i & 63, px = i >> 6;
    ys[i] = co < d.Cout ? a.dy[((size_t)n * npx + px) * d.Cout + co] : 0.f;
  }
  __syncthreads();
  const int co = t & 63, kg = t >> 6;
  int koff[19];
#pragma unroll
  for (int kk = 0; kk < 19; ++kk) {
    int k = kg * 19 + kk;
    if (k >= a.K) k = 0;  // surplus slots recompute k = 0; never stored
    const int ci = k % Cin, tap = k / Cin, kw = tap % d.KW, kh = tap / d.KW;
    koff[kk] = (kh * a.PW + kw) * Cin + ci;
  }
  float acc[19];
#pragma unroll
  for (int kk = 0; kk < 19; ++kk) acc[kk] = 0.f;
  float bsum = 0.f;
  for (int oh = 0; oh < d.OH; ++oh) {
    for (int ow = 0; ow < d.OW; ++ow) {
      const float g = ys[(oh * d.OW + ow) * 64 + co];
      const float* xb = xs + ((oh * d.stride) * a.PW + ow * d.stride) * Cin;
      bsum += g;
#pragma unroll
      for (int kk = 0; kk < 19; ++kk) acc[kk] += xb[koff[kk]] * g;
    }
  }
  if (co < d.Cout) {
    float* slab = a.slab_w + (size_t)n * a.K * d.Cout;
#pragma unroll
    for (int kk = 0; kk < 19; ++kk) {
      const int k = kg * 19 + kk;
      if (k < a.K) slab[(size_t)k * d.Cout + co] = acc[kk];
    }
    if (a.slab_b && kg == 0) a.slab_b[(size_t)n * d.Cout + co] = bsum;
  }
}

// workspace bytes of the thin path, 0 when not eligible
static size_t thin_wgrad_workspace(const lvae_conv_desc* d) {
  const int K = d->KH * d->KW * d->C1;
  if (d->gather != LVAE_GATHER_CONV || d->x2 != nullptr || d->C2 != 0 || d->in_scale != nullptr) return 0;
  if (K > 76 || d->Cout > 64 || d->OH * d->OW > 1024 || d->N < 32 || d->N > 65535) return 0;
  const size_t lds = ((size_t)((d->H + 2 * d->pad) * (d->W + 2 * d->pad) * d->C1 + 3) / 4 * 4 + (size_t)d->OH * d->OW * 64) * sizeof(float);
  if (lds > 160 * 1024) return 0;
  return (size_t)d->N * ((size_t)K * d->Cout + d->Cout) * sizeof(float);
}

size_t conv1x1_wgrad_workspace(const lvae_conv_desc* d);
int conv1x1_wgrad_try(const lvae_conv_desc* d, const float* dy, float* dw, float* db, void* workspace, hipStream_t s);
size_t conv_wgrad_tile_workspace(const lvae_conv_desc* d);
int conv_wgrad_tile_try(const lvae_conv_desc* d, const float* dy, float* dw, float* db, void* workspace, hipStream_t s);

static void wgrad_plan(const lvae_conv_desc* d, int& ksplit, int& px_per_split, int& ncit, int& ncot) {
  const int Cin = d->C1 + d->C2, M = d->N * d->OH * d->OW, ntaps = d->KH * d->KW;
  ncit = (Cin + CT - 1) / CT;
  ncot = (d->Cout + CT - 1) / CT;
  const int tiles = ntaps * ncit * ncot;
  int want = (512 + tiles - 1) / tiles;           // ~2 workgroups per CU in total
  if (want > 64) want = 64;                       // bound the slab traffic of small (1x1) filters
  int maxsplit = (M + 255) / 256;                 // at least 8 stages per workgroup
  ksplit = want < 1 ? 1 : want;
  if (ksplit > maxsplit) ksplit = maxsplit;
  if (ksplit < 1) ksplit = 1;
  px_per_split = (M + ksplit - 1) / ksplit;
  px_per_split = (px_per_split + KP - 1) / KP * KP;
  ksplit = (M + px_per_split - 1) / px_per_split;
}

}  // namespace lvae

using namespace lvae;

extern "C" size_t lvae_conv2d_wgrad_workspace(const lvae_conv_desc* d) {
  if (!d) return 0;
  const size_t img = conv_wgrad_img_workspace(d);   // fp32-stored operands of the <= 8x8 levels, either precision
  if (img) return img;
  const size_t bf = conv3x3_wgrad_bf16_workspace(d);   // precision = LVAE_PREC_BF16 descriptors that have a bf16 weight-gradient kernel
  if (bf) return bf;
  const size_t wino = conv_wgrad_wino_workspace(d);
  if (wino) return wino;
  const size_t direct = conv1x1_wgrad_workspace(d);
  if (direct) return direct;
  const size_t halo = conv_wgrad_tile_workspace(d);
  if (halo) return halo;
  const size_t thin = thin_wgrad_workspace(d);
  if (thin) return thin;
  int ksplit, pps, ncit, ncot;
  wgrad_plan(d, ksplit, pps, ncit, ncot);
  const size_t per = (size_t)d->KH * d->KW * (d->C1 + d->C2) * d->Cout + d->Cout;
  return (size_t)ksplit * per * sizeof(float);
}

extern "C" int32_t lvae_conv2d_wgrad_variant(const lvae_conv_desc* d) {
  if (!d) return LVAE_WGRAD_VARIANT_GENERIC;
  if (conv_wgrad_img_workspace(d)) return LVAE_WGRAD_VARIANT_IMG;
  if (conv3x3_wgrad_bf16_workspace(d)) return LVAE_WGRAD_VARIANT_BF16;
  if (conv_wgrad_wino_workspace(d)) return LVAE_WGRAD_VARIANT_WINO;
  if (conv1x1_wgrad_workspace(d)) return LVAE_WGRAD_VARIANT_DIRECT_1X1;
  if (conv_wgrad_tile_workspace(d)) return LVAE_WGRAD_VARIANT_TILE;
  if (thin_wgrad_workspace(d)) return LVAE_WGRAD_VARIANT_THIN;
  return LVAE_WGRAD_VARIANT_GENERIC;
}

extern "C" int lvae_conv2d_wgrad_f32(const lvae_conv_desc* d, const float* dy, float* dw, float* db, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  int rc = conv_desc_check(d, "lvae_conv2d_wgrad_f32");
  if (rc) return rc;
  LVAE_REQUIRE(dy && dw && workspace, LVAE_EINVAL, "lvae_conv2d_wgrad_f32: null dy/dw/workspace");
  LVAE_REQUIRE(workspace_bytes >= lvae_conv2d_wgrad_workspace(d), LVAE_EWORKSPACE,
               "lvae_conv2d_wgrad_f32: workspace %zu < %zu", workspace_bytes, lvae_conv2d_wgrad_workspace(d));
  static const bool halo_off = tune("LVAE_DISABLE_HALO", 0) != 0;
  if (!halo_off && conv_wgrad_img_workspace(d)) {
    const int hr = conv_wgrad_img_try(d, dy, dw, db, workspace, (hipStream_t)stream);
    if (hr != -1000) return hr;
  }
  if (!halo_off && conv3x3_wgrad_bf16_workspace(d)) {
    const int hr = conv3x3_wgrad_bf16_try(d, dy, dw, db, workspace, (hipStream_t)stream);
    if (hr != -1000) return hr;
  }
  LVAE_REQUIRE(d->x_dtype == LVAE_DT_F32 && d->y_dtype == LVAE_DT_F32, LVAE_EINVAL,
               "lvae_conv2d_wgrad_f32: bf16-stored x / dy need the bf16 weight-gradient kernel, which does not take this shape "
               "(lvae_resblock_bf16_storage(d) == 0)");
  if (!halo_off && conv_wgrad_wino_workspace(d)) {
    const int hr = conv_wgrad_wino_try(d, dy, dw, db, workspace, (hipStream_t)stream);
    if (hr != -1000) return hr;
  }
  if (!halo_off && conv1x1_wgrad_workspace(d)) {
    const int hr = conv1x1_wgrad_try(d, dy, dw, db, workspace, (hipStream_t)stream);
    if (hr != -1000) return hr;
  }
  if (!halo_off && conv_wgrad_tile_workspace(d)) {
    const int hr = conv_wgrad_tile_try(d, dy, dw, db, workspace, (hipStream_t)stream);
    if (hr != -1000) return hr;
  }
  if (!halo_off && thin_wgrad_workspace(d)) {
    ThinWgradArgs ta;
    ta.d = *d;
    ta.dy = dy;
    ta.K = d->KH * d->KW * d->C1;
    ta.PH = d->H + 2 * d->pad;
    ta.PW = d->W + 2 * d->pad;
    ta.slab_w = static_cast<float*>(workspace);
    ta.slab_b = db ? ta.slab_w + (size_t)d->N * ta.K * d->Cout : nullptr;
    const size_t lds = ((size_t)(ta.PH * ta.PW * d->C1 + 3) / 4 * 4 + (size_t)d->OH * d->OW * 64) * sizeof(float);
    static std::atomic<bool> attr_set{false};  // idempotent attribute write; the flag itself is race-free
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_thin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      LVAE_REQUIRE(e == hipSuccess, (int)e, "conv_wgrad_thin: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      attr_set = true;
    }
    hipLaunchKernelGGL(conv_wgrad_thin_kernel, dim3(d->N), dim3(256), lds, (hipStream_t)stream, ta);
    LVAE_LAUNCH_CHECK("conv_wgrad_thin");
    wgrad_reduce_launch(ta.slab_w, ta.slab_b, d->N, d->KH * d->KW, d->C1, d->Cout, d->w_stap, d->w_sk, d->w_sn, dw, db, (hipStream_t)stream);
    LVAE_LAUNCH_CHECK("conv2d_wgrad_reduce");
    return 0;
  }
  WgradArgs a;
  a.d = *d;
  a.dy = dy;
  a.M = d->N * d->OH * d->OW;
  a.ohw = d->OH * d->OW;
  a.Cin = d->C1 + d->C2;
  a.ntaps = d->KH * d->KW;
  wgrad_plan(d, a.ksplit, a.px_per_split, a.ncit, a.ncot);
  a.slab_w = static_cast<float*>(workspace);
  a.slab_b = db ? a.slab_w + (size_t)a.ksplit * a.ntaps * a.Cin * d->Cout : nullptr;
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool xv = (d->C1 % 4 == 0) && (d->C2 % 4 == 0) && al(d->x) && (!d->x2 || al(d->x2)) &&
                  (!d->in_scale || (al(d->in_scale) && al(d->in_shift)));
  const bool yv = (d->Cout % 4 == 0) && al(dy);
  hipStream_t s = (hipStream_t)stream;
  const int grid = a.ksplit * a.ncot * a.ncit * a.ntaps;
  if (xv && yv) hipLaunchKernelGGL((conv_wgrad_kernel<true, true>), dim3(grid), dim3(256), 0, s, a);
  else if (xv) hipLaunchKernelGGL((conv_wgrad_kernel<true, false>), dim3(grid), dim3(256), 0, s, a);
  else if (yv) hipLaunchKernelGGL((conv_wgrad_kernel<false, true>), dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_wgrad_kernel<false, false>), dim3(grid), dim3(256), 0, s, a);
  LVAE_LAUNCH_CHECK("conv2d_wgrad");
  wgrad_reduce_launch(a.slab_w, a.slab_b, a.ksplit, a.ntaps, a.Cin, d->Cout, d->w_stap, d->w_sk, d->w_sn, dw, db, s);
  LVAE_LAUNCH_CHECK("conv2d_wgrad_reduce");
  return 0;
}

extern "C" int32_t lvae_conv2d_wgrad_apply_ok(const lvae_conv_desc* d) {
  return d != nullptr && d->precision == LVAE_PREC_F32 && lvae_conv2d_wgrad_variant(d) == LVAE_WGRAD_VARIANT_WINO && conv_wgrad_wino_apply_ok(d) ? 1 : 0;
}

extern "C" int lvae_conv2d_wgrad_apply_f32(const lvae_conv_desc* d, const lvae_bn_apply* ap, float* dw, float* db, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  int rc = conv_desc_check(d, "lvae_conv2d_wgrad_apply_f32");
  if (rc) return rc;
  LVAE_REQUIRE(lvae_conv2d_wgrad_apply_ok(d), LVAE_EINVAL,
               "lvae_conv2d_wgrad_apply_f32: shape not supported (the Winograd-domain fp32 weight gradient of a 64 -> 64 layer, W = 16 or 32): "
               "use lvae_affine_act_bwd_parts_f32 + lvae_conv2d_wgrad_f32");
  LVAE_REQUIRE(ap && ap->parts && ap->rows > 0 && ap->M == (int64_t)d->N * d->H * d->W && ap->coef && ap->dh && ap->x && ap->out && ap->add == nullptr &&
                   ap->dh_bf16 == 0 && dw && workspace,
               LVAE_EINVAL, "lvae_conv2d_wgrad_apply_f32: needs parts / rows, M = N*H*W, the coefficient block, fp32 dh, x and out, no add");
  LVAE_REQUIRE(workspace_bytes >= lvae_conv2d_wgrad_workspace(d), LVAE_EWORKSPACE, "lvae_conv2d_wgrad_apply_f32: workspace %zu < %zu", workspace_bytes,
               lvae_conv2d_wgrad_workspace(d));
  rc = conv_wgrad_wino_apply_try(d, ap, dw, db, workspace, (hipStream_t)stream);
  LVAE_REQUIRE(rc != -1000, LVAE_EALIGN, "lvae_conv2d_wgrad_apply_f32: buffers must be 16-byte aligned");
  return rc;
}

extern "C" int lvae_conv2d_wgrad_bf16(const lvae_conv_desc* d, const float* dy, float* dw, float* db, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  LVAE_REQUIRE(d != nullptr, LVAE_EINVAL, "lvae_conv2d_wgrad_bf16: null descriptor");
  lvae_conv_desc dd = *d;
  dd.precision = LVAE_PREC_BF16;
  return lvae_conv2d_wgrad_f32(&dd, dy, dw, db, workspace, workspace_bytes, stream);
}

// n independent weight gradients; same result as n calls of lvae_conv2d_wgrad_f32. Launches that share a tile-kernel
// variant and are float4-aligned go out together (<= 12 per launch), everything else one by one.
extern "C" size_t lvae_conv2d_wgrad_grouped_workspace(const lvae_conv_desc* descs, int32_t n) {
  size_t tot = 0;
  for (int i = 0; descs && i < n; ++i) tot += (lvae_conv2d_wgrad_workspace(&descs[i]) + 255) / 256 * 256;
  return tot;
}

extern "C" int lvae_conv2d_wgrad_grouped_f32(const lvae_conv_desc* descs, const float* const* dy, float* const* dw,
                                             float* const* db, int32_t n, void* workspace, size_t workspace_bytes,
                                             void* stream) {
  LVAE_REQUIRE(descs && dy && dw && db && n > 0 && n <= 4096 && workspace, LVAE_EINVAL, "lvae_conv2d_wgrad_grouped_f32: bad arguments");
  LVAE_REQUIRE(workspace_bytes >= lvae_conv2d_wgrad_grouped_workspace(descs, n), LVAE_EWORKSPACE,
               "lvae_conv2d_wgrad_grouped_f32: workspace %zu < %zu", workspace_bytes, lvae_conv2d_wgrad_grouped_workspace(descs, n));
  static const bool halo_off = tune("LVAE_DISABLE_HALO", 0) != 0;
  std::vector<void*> ws(n);
  std::vector<int> kind(n);
  char* wp = static_cast<char*>(workspace);
  for (int i = 0; i < n; ++i) {
    int rc = conv_desc_check(&descs[i], "lvae_conv2d_wgrad_grouped_f32");
    if (rc) return rc;
    LVAE_REQUIRE(dy[i] && dw[i], LVAE_EINVAL, "lvae_conv2d_wgrad_grouped_f32: null dy/dw at %d", i);
    ws[i] = wp;
    wp += (lvae_conv2d_wgrad_workspace(&descs[i]) + 255) / 256 * 256;
    const bool al = (reinterpret_cast<uintptr_t>(dy[i]) & 15) == 0;
    const int img = !halo_off && al ? conv_wgrad_img_kind(&descs[i]) : -1;   // kinds 100 + ...: whole-image tiles (conv_wgrad_img.hip)
    const bool bf16 = !halo_off && img < 0 && conv3x3_wgrad_bf16_workspace(&descs[i]) != 0;   // launched one by one (lvae_conv2d_wgrad_f32 below)
    const bool wino = !halo_off && img < 0 && !bf16 && al && conv_wgrad_wino_workspace(&descs[i]) != 0;
    const bool groupable = !halo_off && img < 0 && !wino && !bf16 && al && descs[i].Cout % 4 == 0 && (descs[i].C1 + descs[i].C2) % 4 == 0 &&
                           conv1x1_wgrad_workspace(&descs[i]) == 0;
    // kinds 0-4: tile kernel variants; 5-7: Winograd kernel for W = 8 / 16 / 32 (grouped only while one problem leaves CUs idle)
    static const int64_t wino_group_max = tune("LVAE_WINO_GROUP_MAX_M", 65536);
    kind[i] = img >= 0 ? 100 + img
                       : (wino ? ((int64_t)descs[i].N * descs[i].H * descs[i].W < wino_group_max ? (descs[i].W == 8 ? 5 : (descs[i].W == 16 ? 6 : 7)) : -1)
                               : (groupable ? conv_wgrad_tile_kind(&descs[i]) : -1));
  }
  std::vector<char> done(n, 0);
  std::vector<int> kinds;   // the distinct kinds present, ascending
  for (int i = 0; i < n; ++i)
    if (kind[i] >= 0 && std::find(kinds.begin(), kinds.end(), kind[i]) == kinds.end()) kinds.push_back(kind[i]);
  std::sort(kinds.begin(), kinds.end());
  for (int k : kinds) {
    constexpr int kCap = 32;
    const int cap = k >= 100 ? kCap : 12;
    const lvae_conv_desc* gd[kCap];
    const float* gy[kCap];
    float* gw[kCap];
    float* gb[kCap];
    void* gs[kCap];
    int idx[kCap];
    int m = 0;
    auto flush = [&]() -> int {
      if (m == 0) return 0;
      int rc;
      if (k >= 100) rc = conv_wgrad_img_grouped(gd, gy, gw, gb, gs, m, k - 100, (hipStream_t)stream);   // (one problem too: the same kernel)
      else rc = m == 1 ? -1000
                       : (k >= 5 ? conv_wgrad_wino_grouped(gd, gy, gw, gb, gs, m, (hipStream_t)stream)
                                 : conv_wgrad_tile_grouped(gd, gy, gw, gb, gs, m, k, (hipStream_t)stream));
      if (rc == 0)
        for (int j = 0; j < m; ++j) done[idx[j]] = 1;
      m = 0;
      return rc == -1000 ? 0 : rc;
    };
    for (int i = 0; i < n; ++i) {
      if (kind[i] != k) continue;
      gd[m] = &descs[i]; gy[m] = dy[i]; gw[m] = dw[i]; gb[m] = db[i]; gs[m] = ws[i]; idx[m] = i;
      if (++m == cap) {
        int rc = flush();
        if (rc) return rc;
      }
    }
    int rc = flush();
    if (rc) return rc;
  }
  for (int i = 0; i < n; ++i) {
    if (done[i]) continue;
    int rc = lvae_conv2d_wgrad_f32(&descs[i], dy[i], dw[i], db[i], ws[i], lvae_conv2d_wgrad_workspace(&descs[i]), stream);
    if (rc) return rc;
  }
  return 0;
}
