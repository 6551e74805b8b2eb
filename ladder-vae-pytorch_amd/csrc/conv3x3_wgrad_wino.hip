// Weight gradient of the large 3x3 / stride 1 / pad 1 convolutions in the Winograd F(2x2,3x3) domain, fp32.
//
//   dg = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G          per (ci, co)
//
// i.e. 16 position GEMMs dU[p][ci][co] = sum_tiles V[p][tile][ci] * W[p][tile][co] with the tile index as the reduction
// dimension: 2.25x fewer fp32 MFMAs than the direct sum over pixels. V = B^T d B comes from the 4x4 input block of the
// tile (with the fused BatchNorm/activation prologue applied while the block is staged), W = A dY A^T from the 2x2 block
// of the output gradient; both transforms have coefficients 0/+-1, so they are exact adds.
//
// Workgroup = (range of 16-tile chunks, block of 32 input channels, group of 64 output channels), 8 waves: wave =
// (position row i, half of the chunk's tiles). The split is over INPUT channels so that the BatchNorm/activation prologue
// (the expensive part of staging) is done once per element; dY, which needs no arithmetic, is the operand staged twice.
// Each wave keeps dU[4i..4i+3][32 ci][64 co] in 128 accumulator registers; per MFMA k-step (two tiles, one per lane half)
// it reads two rows of the 4x4 block (8 ds_read_b32), the 2x2 dY block for both co halves (8 ds_read_b32), spends 22 VALU
// on the row-i transforms and issues 8 MFMAs. Chunks are double-buffered in LDS (pixel strides 48 / 80 floats keep the
// two lane halves on disjoint banks) and fetched one chunk ahead through registers. At the end the two tile halves are
// added through LDS and the workgroup writes one partial slab; conv_wgrad_wino_reduce sums the slabs in a fixed order
// (deterministic), applies G^T . G and accumulates into dw with the caller's strides.
#include <stdlib.h>

#include "lvae_common.h"

namespace lvae {

struct WgWinoArgs {
  lvae_conv_desc d;
  const float* dy;
  float* slab_w;  // [nranges][2 ci blocks][ncog][32 ci][16 positions][64 co]: 4 KB contiguous per input channel
  float* slab_b;  // [nranges][ncog][64] or nullptr
  int nranges, ncog, cpr, total_chunks, cpi;
  uint32_t m_cpi;
};

constexpr int WG_XS = 48;  // x halo pixel stride (floats, 32 channels + pad): 2 pixels = 96 = 32 mod 64 banks
constexpr int WG_DS = 80;  // dy pixel stride (floats, 64 channels + pad): 2 pixels = 160 = 32 mod 64 banks

template <int TPR>  // Winograd tiles per image row (W / 2): 4, 8 or 16; a chunk is 16 tiles = 16 / TPR tile rows
__global__ __launch_bounds__(512, 2) void conv_wgrad_wino_kernel(WgWinoArgs a) {
  kernarg_warmup<sizeof(WgWinoArgs)>();
  const int nwg_ = gridDim.x;
  constexpr bool AP = false;
  const lvae_bn_apply ap = lvae_bn_apply{};
#include "conv3x3_wgrad_wino_body.inc"
}

// the same kernel with the BatchNorm-backward apply that produces its dY operand in front (see the body's header)
struct WgWinoApArgs {
  WgWinoArgs w;
  lvae_bn_apply ap;
};
template <int TPR>
__global__ __launch_bounds__(512, 2) void conv_wgrad_wino_ap_kernel(WgWinoApArgs g) {
  kernarg_warmup<sizeof(WgWinoApArgs)>();
  const WgWinoArgs& a = g.w;
  const lvae_bn_apply& ap = g.ap;
  const int nwg_ = gridDim.x;
  constexpr bool AP = true;
#include "conv3x3_wgrad_wino_body.inc"
}

constexpr int kMaxWinoGroup = 12;
struct WgWinoGroup {
  WgWinoArgs p[kMaxWinoGroup];
};
static_assert(sizeof(WgWinoGroup) <= 4096, "kernel argument block");

// several independent problems in one launch (blockIdx.y = problem): the 8x8 level fills half the CUs per problem
template <int TPR>
__global__ __launch_bounds__(512, 2) void conv_wgrad_wino_grouped_kernel(WgWinoGroup g) {
  kernarg_warmup<(sizeof(WgWinoGroup) < 1024 ? sizeof(WgWinoGroup) : 1024)>();
  const WgWinoArgs& a = g.p[blockIdx.y];
  const int nwg_ = a.nranges * a.ncog * 2;
  if ((int)blockIdx.x >= nwg_) return;  // uniform per workgroup, before any barrier
  constexpr bool AP = false;
  const lvae_bn_apply ap = lvae_bn_apply{};
#include "conv3x3_wgrad_wino_body.inc"
}

struct WinoReduceArgs {
  const float* slab_w;
  const float* slab_b;
  int nranges, ncog, Cout, pad_;
  int64_t stap, sk, sn;
  float* dw;
  float* db;
};
struct WinoReduceGroup {
  WinoReduceArgs p[kMaxWinoGroup];
};

// dw[tap(a,b)][ci][co] += (G^T (sum_ranges slab) G)[a][b], 1024-thread workgroups, ranges summed in eight slices (slice = range mod 8,
// ascending inside a slice, then a balanced tree over the slices: the same order in both forms, so they agree bitwise).
//   HALF (single gradient): workgroup = (ci, half of a group of 64 co), threads = 8 float4 columns x 16 positions x 8 slices, so each
//     range contributes sixteen 128-byte pieces (whole cache lines). 128 workgroups per 64x64 gradient instead of 64: the pass is bound
//     by how many bytes a CU keeps in flight, not by HBM (64 workgroups: 9.7 us for 33.5 MB, 128: 9.0 us; 256 workgroups of 64-byte
//     pieces: 31 us).
//   !HALF (grouped launch, up to 12 gradients = 768 workgroups already): workgroup = (ci, 64 co), threads = 16 float4 columns x
//     16 positions x 4, each carrying two slices; halving these workgroups as well cost 12 us per launch.
template <bool HALF>
__device__ __forceinline__ void wino_reduce_body(const WinoReduceArgs& a, int bx, float* red, float (*red_b)[64]) {
  constexpr int CW = HALF ? 32 : 64;   // output channels per workgroup; red is [8][16][CW]
  const int ncog = a.ncog, nranges = a.nranges;
  const int half = HALF ? (bx & 1) : 0, cc = HALF ? (bx >> 1) : bx, ci = cc / ncog, cog = cc % ncog;
  const int t = threadIdx.x;
  const int q = HALF ? (t & 7) : (t & 15), p = HALF ? ((t >> 3) & 15) : ((t >> 4) & 15), rs = HALF ? (t >> 7) : (t >> 8);
  const float* src = a.slab_w + ((((size_t)(ci >> 5) * ncog + cog) * 32 + (ci & 31)) * 16 + p) * 64 + half * 32 + q * 4;
  const size_t rstride = (size_t)2 * ncog * 32 * 16 * 64;
  // slab rows p = 4 i + b, b < 3: the producer has applied the column half of G^T dU G (row 4 i + 3 is never written)
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if ((p & 3) != 3) {
    if (HALF) {
#pragma unroll 8
      for (int r = rs; r < nranges; r += 8) s += *reinterpret_cast<const f32x4*>(src + r * rstride);
      *reinterpret_cast<f32x4*>(red + (rs * 16 + p) * CW + q * 4) = s;
    } else {
      f32x4 s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int r = rs; r < nranges; r += 8) {
        s += *reinterpret_cast<const f32x4*>(src + r * rstride);
        if (r + 4 < nranges) s1 += *reinterpret_cast<const f32x4*>(src + (r + 4) * rstride);
      }
      *reinterpret_cast<f32x4*>(red + (rs * 16 + p) * CW + q * 4) = s;
      *reinterpret_cast<f32x4*>(red + ((rs + 4) * 16 + p) * CW + q * 4) = s1;
    }
  }
  const bool do_b = a.db != nullptr && ci == 0 && half == 0;
  if (do_b) {
    const int co = t & 63, slice = t >> 6;
    float v = 0.f;
#pragma unroll 4
    for (int r = slice; r < nranges; r += 16) v += a.slab_b[(size_t)(r * ncog + cog) * 64 + co];
    red_b[slice][co] = v;
  }
  __syncthreads();
  if (t < 3 * CW && cog * 64 + half * 32 + (t % CW) < a.Cout) {
    const int co = t % CW, ga = t / CW;  // output row a of G^T dU G
    float u[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const float* rp = red + (i * 4 + b) * CW + co;
        u[i][b] = ((rp[0] + rp[16 * CW]) + (rp[32 * CW] + rp[48 * CW])) + ((rp[64 * CW] + rp[80 * CW]) + (rp[96 * CW] + rp[112 * CW]));
      }
    // the row half: G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
    float* o = a.dw + (int64_t)ci * a.sk + (int64_t)(cog * 64 + half * 32 + co) * a.sn + (int64_t)(ga * 3) * a.stap;
#pragma unroll
    for (int b = 0; b < 3; ++b)
      o[b * a.stap] += ga == 0 ? u[0][b] + 0.5f * (u[1][b] + u[2][b]) : (ga == 1 ? 0.5f * (u[1][b] - u[2][b]) : 0.5f * (u[1][b] + u[2][b]) + u[3][b]);
  }
  if (do_b && t >= 256 && t < 320 && cog * 64 + t - 256 < a.Cout) {
    const int co = t - 256;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += red_b[k][co];
    a.db[cog * 64 + co] += v;
  }
}

__global__ __launch_bounds__(1024) void conv_wgrad_wino_reduce_kernel(WinoReduceArgs a) {
  kernarg_warmup<sizeof(WinoReduceArgs)>();
  __shared__ __attribute__((aligned(16))) float red[8 * 16 * 32];
  __shared__ float red_b[16][64];
  wino_reduce_body<true>(a, blockIdx.x, red, red_b);
}

__global__ __launch_bounds__(1024) void conv_wgrad_wino_reduce_grouped_kernel(WinoReduceGroup g) {
  kernarg_warmup<(sizeof(WinoReduceGroup) < 1024 ? sizeof(WinoReduceGroup) : 1024)>();
  const WinoReduceArgs& a = g.p[blockIdx.y];
  if ((int)blockIdx.x >= 64 * a.ncog) return;   // uniform per workgroup, before any barrier
  __shared__ __attribute__((aligned(16))) float red[8 * 16 * 64];
  __shared__ float red_b[16][64];
  wino_reduce_body<false>(a, blockIdx.x, red, red_b);
}

static bool al16g(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static bool wg_wino_plan(const lvae_conv_desc* d, WgWinoArgs& a) {
  static const bool off = tune("LVAE_DISABLE_WINO_WGRAD", 0) != 0 || tune("LVAE_DISABLE_WINO", 0) != 0;  // A/B switch (tuning builds only)
  if (off) return false;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->gather != LVAE_GATHER_CONV) return false;
  if (d->C1 != 64 || d->C2 != 0 || d->x2 != nullptr || d->Cout % 4 != 0 || d->Cout > 256) return false;
  if (d->OH != d->H || d->OW != d->W || (d->H & 1)) return false;
  if (d->W != 8 && d->W != 16 && d->W != 32) return false;
  const int tpr = d->W / 2, tr = 16 / tpr;
  if ((d->H / 2) % tr != 0) return false;
  if (!al16g(d->x) || !al16g(d->in_scale) || !al16g(d->in_shift)) return false;
  const int64_t M = (int64_t)d->N * d->H * d->W;
  static const int64_t min_m = tune("LVAE_WINO_WGRAD_MIN_M", 256 * 64);
  if (M < min_m || M * 256 >= ((int64_t)1 << 31)) return false;
  a.ncog = (d->Cout + 63) / 64;
  a.cpi = (d->H / 2) / tr;
  a.total_chunks = d->N * a.cpi;
  int nranges = 128 / a.ncog;  // 256 workgroups with the two input-channel blocks
  if (nranges < 1) nranges = 1;
  static const int min_cpr = (int)tune("LVAE_WINO_WGRAD_MIN_CPR", 4);
  if (nranges > a.total_chunks / min_cpr) nranges = a.total_chunks / min_cpr;  // slab traffic: at least min_cpr chunks per slab
  if (nranges < 1) nranges = 1;
  a.cpr = (a.total_chunks + nranges - 1) / nranges;
  a.nranges = (a.total_chunks + a.cpr - 1) / a.cpr;
  a.m_cpi = fastdiv_magic(a.cpi);
  return true;
}

size_t conv_wgrad_wino_workspace(const lvae_conv_desc* d) {
  WgWinoArgs a;
  if (!wg_wino_plan(d, a)) return 0;
  return (size_t)a.nranges * a.ncog * (2 * 16 * 32 * 64 + 64) * sizeof(float);
}

template <int TPR>
static int launch_wg_wino(const WgWinoArgs& a, hipStream_t s) {
  auto kern = conv_wgrad_wino_kernel<TPR>;
  constexpr int W = 2 * TPR, HW2 = W + 2, TR = 16 / TPR, HP = (2 * TR + 2) * HW2;
  size_t lds = (size_t)2 * (HP * WG_XS + 64 * WG_DS) * sizeof(float);
  const size_t lds_r = (size_t)(4 * 4 * 2 * 16 * 64 + 128) * sizeof(float);
  if (lds < lds_r) lds = lds_r;
  static std::atomic<bool> attr_set{false};  // idempotent attribute write; the flag itself is race-free
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      set_error("conv_wgrad_wino: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.nranges * a.ncog * 2), dim3(512), lds, s, a);
  LVAE_LAUNCH_CHECK("conv_wgrad_wino");
  return 0;
}

// returns -1000 when not eligible
int conv_wgrad_wino_try(const lvae_conv_desc* d, const float* dy, float* dw, float* db, void* workspace, hipStream_t s) {
  WgWinoArgs a;
  if (!wg_wino_plan(d, a) || !al16g(dy) || !al16g(workspace)) return -1000;
  a.d = *d;
  a.dy = dy;
  a.slab_w = static_cast<float*>(workspace);
  a.slab_b = db ? a.slab_w + (size_t)a.nranges * a.ncog * 2 * 16 * 32 * 64 : nullptr;
  int rc;
  if (d->W == 8) rc = launch_wg_wino<4>(a, s);
  else if (d->W == 16) rc = launch_wg_wino<8>(a, s);
  else rc = launch_wg_wino<16>(a, s);
  if (rc) return rc;
  const WinoReduceArgs ra{a.slab_w, a.slab_b, a.nranges, a.ncog, d->Cout, 0, d->w_stap, d->w_sk, d->w_sn, dw, db};
  hipLaunchKernelGGL(conv_wgrad_wino_reduce_kernel, dim3(128 * a.ncog), dim3(1024), 0, s, ra);
  LVAE_LAUNCH_CHECK("conv_wgrad_wino_reduce");
  return 0;
}

template <int TPR>
static int launch_wg_wino_ap(const WgWinoApArgs& g, hipStream_t s) {
  auto kern = conv_wgrad_wino_ap_kernel<TPR>;
  constexpr int W = 2 * TPR, HW2 = W + 2, TR = 16 / TPR, HP = (2 * TR + 2) * HW2;
  size_t lds = (size_t)2 * (HP * WG_XS + 64 * WG_DS) * sizeof(float);
  const size_t lds_r = (size_t)(4 * 4 * 2 * 16 * 64 + 128) * sizeof(float);
  if (lds < lds_r) lds = lds_r;
  static std::atomic<bool> attr_set{false};  // idempotent attribute write; the flag itself is race-free
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      set_error("conv_wgrad_wino_ap: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(g.w.nranges * g.w.ncog * 2), dim3(512), lds, s, g);
  LVAE_LAUNCH_CHECK("conv_wgrad_wino_ap");
  return 0;
}

// 1 when the weight gradient of `d` can take its dY operand from a deferred BatchNorm-backward apply (lvae_conv2d_wgrad_apply_f32)
bool conv_wgrad_wino_apply_ok(const lvae_conv_desc* d) {
  WgWinoArgs a;
  return d != nullptr && d->Cout == 64 && (d->W == 16 || d->W == 32) && d->x_dtype == LVAE_DT_F32 && d->y_dtype == LVAE_DT_F32 && wg_wino_plan(d, a);
}

// returns -1000 when not eligible
int conv_wgrad_wino_apply_try(const lvae_conv_desc* d, const lvae_bn_apply* ap, float* dw, float* db, void* workspace, hipStream_t s) {
  WgWinoApArgs g;
  WgWinoArgs& a = g.w;
  if (!conv_wgrad_wino_apply_ok(d) || !wg_wino_plan(d, a) || !al16g(workspace)) return -1000;
  if (!al16g(ap->parts) || !al16g(ap->coef) || !al16g(ap->dh) || !al16g(ap->x) || !al16g(ap->out) || !al16g(ap->drop)) return -1000;
  a.d = *d;
  a.dy = nullptr;
  a.slab_w = static_cast<float*>(workspace);
  a.slab_b = db ? a.slab_w + (size_t)a.nranges * a.ncog * 2 * 16 * 32 * 64 : nullptr;
  g.ap = *ap;
  int rc = d->W == 16 ? launch_wg_wino_ap<8>(g, s) : launch_wg_wino_ap<16>(g, s);
  if (rc) return rc;
  const WinoReduceArgs ra{a.slab_w, a.slab_b, a.nranges, a.ncog, d->Cout, 0, d->w_stap, d->w_sk, d->w_sn, dw, db};
  hipLaunchKernelGGL(conv_wgrad_wino_reduce_kernel, dim3(128 * a.ncog), dim3(1024), 0, s, ra);
  LVAE_LAUNCH_CHECK("conv_wgrad_wino_reduce");
  return 0;
}

// n <= kMaxWinoGroup eligible descriptors with the same image width, each with its own workspace; -1000 when one is not eligible
int conv_wgrad_wino_grouped(const lvae_conv_desc* const* ds, const float* const* dy, float* const* dw, float* const* db,
                            void* const* workspace, int n, hipStream_t s) {
  WgWinoGroup g;
  WinoReduceGroup rg;
  int max_wgs = 0, max_ncog = 0;
  const int W = ds[0]->W;
  for (int i = 0; i < n; ++i) {
    WgWinoArgs& a = g.p[i];
    if (!wg_wino_plan(ds[i], a) || !al16g(dy[i]) || !al16g(workspace[i]) || ds[i]->W != W) return -1000;
    a.d = *ds[i];
    a.dy = dy[i];
    a.slab_w = static_cast<float*>(workspace[i]);
    a.slab_b = db[i] ? a.slab_w + (size_t)a.nranges * a.ncog * 2 * 16 * 32 * 64 : nullptr;
    if (a.nranges * a.ncog * 2 > max_wgs) max_wgs = a.nranges * a.ncog * 2;
    if (a.ncog > max_ncog) max_ncog = a.ncog;
    rg.p[i] = WinoReduceArgs{a.slab_w, a.slab_b, a.nranges, a.ncog, ds[i]->Cout, 0, ds[i]->w_stap, ds[i]->w_sk, ds[i]->w_sn, dw[i], db[i]};
  }
  for (int i = n; i < kMaxWinoGroup; ++i) {
    g.p[i] = g.p[0];
    rg.p[i] = rg.p[0];
  }
  const int tpr = W / 2;
  const int hp = (2 * (16 / tpr) + 2) * (W + 2);
  size_t lds = (size_t)2 * (hp * WG_XS + 64 * WG_DS) * sizeof(float);
  const size_t lds_r = (size_t)(4 * 4 * 2 * 16 * 64 + 128) * sizeof(float);
  if (lds < lds_r) lds = lds_r;
  const void* kern = W == 8 ? (const void*)conv_wgrad_wino_grouped_kernel<4>
                            : (W == 16 ? (const void*)conv_wgrad_wino_grouped_kernel<8> : (const void*)conv_wgrad_wino_grouped_kernel<16>);
  static std::atomic<bool> attr_set[3] = {};
  const int slot = W == 8 ? 0 : (W == 16 ? 1 : 2);
  if (!attr_set[slot]) {
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      set_error("conv_wgrad_wino_grouped: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    attr_set[slot] = true;
  }
  if (W == 8) hipLaunchKernelGGL(conv_wgrad_wino_grouped_kernel<4>, dim3(max_wgs, n), dim3(512), lds, s, g);
  else if (W == 16) hipLaunchKernelGGL(conv_wgrad_wino_grouped_kernel<8>, dim3(max_wgs, n), dim3(512), lds, s, g);
  else hipLaunchKernelGGL(conv_wgrad_wino_grouped_kernel<16>, dim3(max_wgs, n), dim3(512), lds, s, g);
  LVAE_LAUNCH_CHECK("conv_wgrad_wino_grouped");
  hipLaunchKernelGGL(conv_wgrad_wino_reduce_grouped_kernel, dim3(64 * max_ncog, n), dim3(1024), 0, s, rg);
  LVAE_LAUNCH_CHECK("conv_wgrad_wino_reduce_grouped");
  return 0;
}

}  // namespace lvae
