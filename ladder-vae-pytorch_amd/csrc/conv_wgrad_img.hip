// Weight gradients of the low-resolution levels (H*W <= 64: the 8x8, 4x4 and 2x2 levels, 11 of the 15 stochastic layers of the CIFAR
// model) on the bf16 matrix pipe, "whole-image tiles", many gradients per launch.
//
//   dW[tap][ci][co] += sum_pixels T(x)[pixel + tap][ci] * dy[pixel][co],   db[co] += sum_pixels dy[pixel][co]
//   (T = the fused BatchNorm-apply + activation of the forward; replaces autograd's convolution_backward(weight, bias) of
//    lib/nn.py:83-87 (3x3, 64 -> 64) and lib/nn.py:118 (GateLayer2d's 1x1, 64 -> 128) at these levels)
//
// Why a kernel of its own (round 5). These gradients used to run on the fp32 MFMA: conv_wgrad_ws_grouped_kernel (4x4, 2x2 and every
// 1x1) spends ~27 us on a workgroup's ONE 64-pixel tile (7.7 us of v_mfma_f32_32x32x2_f32 — which shares the vector ALU's lanes — plus
// the staging latency), and the Winograd-domain kernel of the 8x8 level writes and re-reads 128 x 262 KB of partial slabs per gradient
// (15 us per gradient in the step, slab-traffic bound). Together 3.3 ms of a 34 ms step for ~1 % of its FLOPs
// (profiles/r04_last_step_by_grid.txt). Here:
//  * a tile is 64 pixels = WHOLE images (1 at 8x8, 4 at 4x4, 16 at 2x2), as in resblock_img.hip, so the zero ring of the patch is the
//    only halo and a tap shift is an address offset;
//  * the products run as six exact bf16-piece products per fp32 product on v_mfma_f32_32x32x16_bf16 (SPLIT = 3: the fp32-equivalent
//    form every fp32 kernel of the library uses, same parity tolerances) or with bf16 operands (SPLIT = 1, precision = LVAE_PREC_BF16):
//    864 / 144 MFMAs per 3x3 tile = 2.9 / 0.5 us on a CU, beside the vector ALU instead of on it;
//  * the reduction runs over PIXELS while the tensors are NHWC, so both operands are staged as [pixel][channel] bf16 planes and read
//    with the transposing ds_read_b64_tr_b16 (bf16_frag.h tr_frag), exactly as conv3x3_wgrad_bf16_kernel does;
//  * a workgroup is persistent over `tiles per workgroup` tiles (raw operands of the next tile prefetched into registers during the
//    MFMAs) and keeps its accumulators in registers, so a gradient writes ntiles / tpw slabs of 147 KB instead of one per tile:
//    8x8: 64 slabs = 9.4 MB (Winograd form: 33.5 MB), 4x4: 16, 2x2: 4;
//  * up to 32 gradients share a launch (blockIdx.y = problem; 112-byte argument blocks), and one grouped fixed-order reduce
//    (wgrad_reduce_grouped_kernel) sums the slabs into the gradient arena: deterministic, no float atomics.
// 512 threads: KIND 0 (3x3, Cin <= 64, Cout <= 64): wave = (ci half, co half, tap group {0-4 | 5-8}), <= 5 accumulator tiles of
// 32 ci x 32 co; KIND 1 (1x1, Cin <= 64, Cout <= 128): wave = (ci half, co quarter), one accumulator tile.
#include <stdlib.h>

#include "bf16_frag.h"
#include "lvae_common.h"

namespace lvae {

struct WgImgProb {
  const float* x;         // [N][H][W][Cin]
  const float* dy;        // [N][H][W][Cout]
  const float* in_scale;  // [Cin] or null
  const float* in_shift;
  float* slab_w;          // [nwg][ntap][Cin][Cout]
  float* slab_b;          // [nwg][Cout] or null
  int32_t N, HW, W, Cin, Cout, in_act, NI, halo_w, halo_h, halo_px, ntiles, nwg;
  uint32_t m_hw, m_w, m_per_img, m_halo_w;
};
static_assert(sizeof(WgImgProb) == 112, "argument block layout");
constexpr int kWgImgMax = 32;
struct WgImgGroup {
  WgImgProb p[kWgImgMax];
};
static_assert(sizeof(WgImgGroup) <= 4096, "kernel argument block");

constexpr int WGI_LDK = 72;    // bf16 elements per x row (64 channels + 8 pad = 144 bytes)
constexpr int WGI_LDD1 = 136;  // bf16 elements per dy row of the 1x1 kind (128 channels + 8 pad)

// XV: 32-pixel passes over the patch (halo_px <= 32 XV)
template <int KIND, int SPLIT, int XV>
__global__ __launch_bounds__(512) void wgrad_img_kernel(WgImgGroup g) {
  constexpr int LDK = WGI_LDK, LDD = KIND == 0 ? WGI_LDK : WGI_LDD1;
  constexpr int NTAP = KIND == 0 ? 9 : 1, NACC = KIND == 0 ? 5 : 1;
  constexpr int PAD = KIND == 0 ? 1 : 0;
  constexpr int DV = KIND == 0 ? 2 : 4;           // float4 of the dy tile per thread
  constexpr int DCH = KIND == 0 ? 16 : 32;        // float4 per dy row
  constexpr int DPX = 512 / DCH;                  // dy pixel rows per pass
#ifdef LVAE_WGI_DBG  // compile-time phase-skip mask of the profiling builds (tools/wgi_ab.sh); never defined in the product
  constexpr int dbg = LVAE_WGI_DBG;  // 1: no global loads, 2: no MFMAs, 4: no staging arithmetic / LDS writes, 8: fragments not read from LDS, 16: no slab stores
#else
  constexpr int dbg = 0;
#endif
  const WgImgProb& a = g.p[blockIdx.y];
  if ((int)blockIdx.x >= a.nwg) return;   // uniform per workgroup, before any barrier
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* Xs = reinterpret_cast<__bf16*>(smem_raw);   // [SPLIT][halo_px][LDK]
  const int x_plane = a.halo_px * LDK;
  __bf16* Ds = Xs + (size_t)SPLIT * x_plane;          // [SPLIT][64][LDD]
  constexpr int d_plane = 64 * LDD;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int cih = wave & 1;
  const int cob = KIND == 0 ? ((wave >> 1) & 1) : (wave >> 1);   // 32-channel block of the output channels
  const int tg = KIND == 0 ? (wave >> 2) : 0;
  const int tap0 = tg * 5, ntap = KIND == 0 ? (tg == 0 ? 5 : 4) : 1;
  const int li = lane & 31, lh = lane >> 5, G = lane >> 4, i16 = lane & 15;
  const int Cin = a.Cin, Cout = a.Cout, HW = a.HW, H = HW / a.W;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // transposed-read row of this lane for (k-step s, half-read rd): tile pixel 16 s + 8 (G >> 1) + 4 rd + (i16 >> 2) -> patch pixel of tap (0, 0)
  auto xrow_of = [&](int s, int rd) {
    const int p = 16 * s + 8 * (G >> 1) + 4 * rd + (i16 >> 2);
    if (KIND == 1) return p;
    const int img = fastdiv(p, a.m_hw), r = p - img * HW;
    const int ty = fastdiv(r, a.m_w), tx = r - ty * a.W;
    return (img * a.halo_h + ty) * a.halo_w + tx;
  };
  const int chx = cih * 32 + 16 * (G & 1) + 4 * (i16 & 3);   // channel offset of this lane's 8-byte piece in an x row
  const int chd = cob * 32 + 16 * (G & 1) + 4 * (i16 & 3);   // ... in a dy row
  const int drow0 = 8 * (G >> 1) + (i16 >> 2);

  // staging maps: thread -> (pixel, 4 channels); the raw operands of the next tile live in registers during the MFMAs
  const int c4 = (t & 15) * 4, px0 = t >> 4;
  const int c4d = (t & (DCH - 1)) * 4, pxd0 = t / DCH;
  const bool cx_ok = c4 < Cin, cd_ok = c4d < Cout;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = zero4;
  if (a.in_scale && cx_ok) {
    sc = *reinterpret_cast<const f32x4*>(a.in_scale + c4);
    sh = *reinterpret_cast<const f32x4*>(a.in_shift + c4);
  }
  const int per_img = a.halo_h * a.halo_w;
  f32x4 xr[XV], dr[DV];
  unsigned xok = 0, dok = 0;
  auto prefetch = [&](int tile) {
    const int n0 = tile * a.NI;
    xok = 0;
#pragma unroll
    for (int u = 0; u < XV; ++u) {
      const int px = px0 + 32 * u;
      const int img = fastdiv(px, a.m_per_img), r = px - img * per_img;
      const int hy = fastdiv(r, a.m_halo_w), hx = r - hy * a.halo_w;
      const int n = n0 + img, ih = hy - PAD, iw = hx - PAD;
      const bool ok = (px < a.halo_px) & (n < a.N) & ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)a.W) & cx_ok;
      const size_t off = ok ? ((size_t)n * HW + ih * a.W + iw) * Cin + c4 : 0;
      xr[u] = (dbg & 1) ? zero4 : *reinterpret_cast<const f32x4*>(a.x + off);
      xok |= ok ? (1u << u) : 0u;
    }
    dok = 0;
#pragma unroll
    for (int u = 0; u < DV; ++u) {
      const int p = pxd0 + DPX * u;
      const int img = fastdiv(p, a.m_hw);
      const bool ok = (n0 + img < a.N) & cd_ok;
      const size_t off = ok ? ((size_t)n0 * HW + p) * Cout + c4d : 0;
      dr[u] = (dbg & 1) ? zero4 : *reinterpret_cast<const f32x4*>(a.dy + off);
      dok |= ok ? (1u << u) : 0u;
    }
  };

  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f32x4 bsum = zero4;

  int tile = blockIdx.x;
  if (tile < a.ntiles) prefetch(tile);
  for (; tile < a.ntiles; tile += a.nwg) {
    // registers -> LDS planes (transform, exact split / rounding to bf16); rows that do not exist are zero
    if (!(dbg & 4))
#pragma unroll
    for (int u = 0; u < XV; ++u) {
      const int px = px0 + 32 * u;
      if (px < a.halo_px) {
        f32x4 w = zero4;
        if ((xok >> u) & 1u) {
          w = xr[u];
          if (a.in_scale) w = act_fwd4(w * sc + sh, a.in_act);
        }
        bf16x4 pl[SPLIT];
        split4<SPLIT>(w, pl);
#pragma unroll
        for (int q = 0; q < SPLIT; ++q) *reinterpret_cast<bf16x4*>(Xs + q * x_plane + px * LDK + c4) = pl[q];
      }
    }
    if (!(dbg & 4))
#pragma unroll
    for (int u = 0; u < DV; ++u) {
      const int p = pxd0 + DPX * u;
      const f32x4 w = ((dok >> u) & 1u) ? dr[u] : zero4;
      bsum += w;
      bf16x4 pl[SPLIT];
      split4<SPLIT>(w, pl);
#pragma unroll
      for (int q = 0; q < SPLIT; ++q) *reinterpret_cast<bf16x4*>(Ds + q * d_plane + p * LDD + c4d) = pl[q];
    }
    __syncthreads();
    if (tile + a.nwg < a.ntiles) prefetch(tile + a.nwg);

#pragma unroll 1
    for (int s = 0; s < 4; ++s) {   // k-steps of 16 pixels
      const int xr0 = xrow_of(s, 0), xr1 = xrow_of(s, 1);
      bf16x8 bfr[SPLIT];
#pragma unroll
      for (int q = 0; q < SPLIT; ++q)
        bfr[q] = (dbg & 8) ? __builtin_bit_cast(bf16x8, f32x4{(float)s, (float)q, 1.f, 2.f})
                           : tr_frag(Ds + q * d_plane + (16 * s + drow0) * LDD + chd, Ds + q * d_plane + (16 * s + drow0 + 4) * LDD + chd);
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        if (j < ntap) {
          const int tap = tap0 + j, kh = tap / 3, kw = tap - kh * 3;
          const int off = KIND == 0 ? kh * a.halo_w + kw : 0;
          bf16x8 afr[SPLIT];
#pragma unroll
          for (int q = 0; q < SPLIT; ++q)
            afr[q] = (dbg & 8) ? __builtin_bit_cast(bf16x8, f32x4{(float)(xr0 + off), (float)q, (float)xr1, 2.f})
                               : tr_frag(Xs + q * x_plane + (xr0 + off) * LDK + chx, Xs + q * x_plane + (xr1 + off) * LDK + chx);
          if (dbg & 2) {
#pragma unroll
            for (int q = 0; q < SPLIT; ++q) asm volatile("" ::"v"(afr[q]), "v"(bfr[q]));
          } else
          if (SPLIT == 1) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0], bfr[0], acc[j], 0, 0, 0);
          } else {  // piece products in ascending order of magnitude
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[SPLIT - 1], bfr[0], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0], bfr[SPLIT - 1], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1], bfr[1], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1], bfr[0], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0], bfr[1], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0], bfr[0], acc[j], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();  // the planes are read: the next tile overwrites them
  }

  // partial slab straight from the accumulators (row = ci, 32 consecutive co per lane half)
  float* sw = a.slab_w + (size_t)blockIdx.x * NTAP * Cin * Cout;
#pragma unroll
  for (int j = 0; j < NACC; ++j) {
    if (j < ntap) {
      const int tap = tap0 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = cih * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, co = cob * 32 + li;
        if (ci < Cin && co < Cout && !((dbg & 16) && acc[j][r] != 12345.f)) sw[((size_t)tap * Cin + ci) * Cout + co] = acc[j][r];
      }
    }
  }
  if (a.slab_b) {
    float* red = reinterpret_cast<float*>(smem_raw);   // [DPX pixel groups][4 DCH]
    *reinterpret_cast<f32x4*>(red + pxd0 * (4 * DCH) + c4d) = bsum;
    __syncthreads();
    if (t < 4 * DCH) {
      float v = 0.f;
#pragma unroll
      for (int gq = 0; gq < DPX; ++gq) v += red[gq * (4 * DCH) + t];
      if (t < Cout) a.slab_b[(size_t)blockIdx.x * Cout + t] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------
static bool al16i(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#ifndef LVAE_WGRAD_IMG
#define LVAE_WGRAD_IMG 1
#endif

// kind of the whole-image weight-gradient kernel a descriptor takes: -1 none; else KIND | SPLIT-is-1 << 1 | XV slot << 2
static int wgi_plan(const lvae_conv_desc* d, WgImgProb& a) {
  static const bool on = tune("LVAE_WGRAD_IMG", LVAE_WGRAD_IMG) != 0;   // A/B switch (tuning builds only)
  if (!on || d == nullptr) return -1;
  const bool k3 = d->KH == 3 && d->KW == 3 && d->pad == 1, k1 = d->KH == 1 && d->KW == 1 && d->pad == 0;
  if (!(k3 || k1) || d->stride != 1 || d->gather != LVAE_GATHER_CONV || d->x2 != nullptr || d->C2 != 0 || d->OH != d->H || d->OW != d->W) return -1;
  if (d->x_dtype != LVAE_DT_F32 || d->y_dtype != LVAE_DT_F32) return -1;
  const int HW = d->H * d->W;
  if (HW < 4 || HW > 64 || 64 % HW != 0 || d->N < 1) return -1;
  if (d->C1 > 64 || d->C1 % 4 != 0 || d->Cout % 4 != 0 || d->Cout > (k3 ? 64 : 128)) return -1;
  if (k1 && d->Cout <= 64 && d->C1 <= 32) return -1;   // tiny 1x1 problems: nothing to gain
  if (!al16i(d->x) || !al16i(d->in_scale) || !al16i(d->in_shift) || (d->in_scale != nullptr && d->in_shift == nullptr)) return -1;
  if ((int64_t)d->N * HW * 128 >= ((int64_t)1 << 31)) return -1;
  const int pad = k3 ? 1 : 0;
  a.N = d->N; a.HW = HW; a.W = d->W; a.Cin = d->C1; a.Cout = d->Cout; a.in_act = d->in_act;
  a.NI = 64 / HW;
  a.halo_h = d->H + 2 * pad;
  a.halo_w = d->W + 2 * pad;
  a.halo_px = a.NI * a.halo_h * a.halo_w;
  if (a.halo_px > 256) return -1;
  a.ntiles = (d->N + a.NI - 1) / a.NI;
  // tiles per workgroup: what a gradient costs is launch + first-tile latency per workgroup and 147 KB of slab per workgroup, against
  // ~5 us per further tile; but a gradient should still spread over enough CUs (whole step, one box: 2 -> 33.15 ms, 4 -> 32.69, 8 -> 32.57,
  // 16 -> 32.88; profiles/r05_wgrad_img_ab.txt)
  // Levels with few tiles (2x2 at batch 256: 16) keep at least 8 workgroups per gradient, so that a group of 32 fills the chip.
  static const int tpw_max = (int)tune("LVAE_WGRAD_IMG_TPW", 8);
  static const int min_wg = (int)tune("LVAE_WGRAD_IMG_MIN_WG", 8);
  int tpw = a.ntiles / min_wg;
  tpw = tpw < 1 ? 1 : (tpw > tpw_max ? tpw_max : tpw);
  int nwg = (a.ntiles + tpw - 1) / tpw;
  if (nwg > 256) nwg = 256;
  if (nwg < 1) nwg = 1;
  a.nwg = nwg;
  a.m_hw = fastdiv_magic(HW);
  a.m_w = fastdiv_magic(d->W);
  a.m_per_img = fastdiv_magic(a.halo_h * a.halo_w);
  a.m_halo_w = fastdiv_magic(a.halo_w);
  const int xv = a.halo_px <= 64 ? 0 : (a.halo_px <= 128 ? 1 : (a.halo_px <= 160 ? 2 : 3));
  return (k3 ? 0 : 1) | ((d->precision == LVAE_PREC_BF16 ? 1 : 0) << 1) | (xv << 2);
}

size_t conv_wgrad_img_workspace(const lvae_conv_desc* d) {
  WgImgProb a;
  if (wgi_plan(d, a) < 0) return 0;
  const size_t ntap = d->KH * d->KW;
  return (size_t)a.nwg * (ntap * a.Cin * a.Cout + a.Cout) * sizeof(float);
}

int conv_wgrad_img_kind(const lvae_conv_desc* d) {
  WgImgProb a;
  return wgi_plan(d, a);
}

struct ReduceArgs {
  const float* slab_w;
  const float* slab_b;
  int ksplit, ntaps, Cin, Cout;
  int64_t stap, sk, sn;
  float* dw;
  float* db;
};
void wgrad_reduce_grouped_launch(const ReduceArgs* r, int n, hipStream_t s);

template <int KIND, int SPLIT, int XV>
static int wgi_launch(const WgImgGroup& g, int n, int max_wgs, size_t lds, hipStream_t s) {
  auto kern = wgrad_img_kernel<KIND, SPLIT, XV>;
  static std::atomic<bool> attr_set{false};  // idempotent attribute write; the flag itself is race-free
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      set_error("conv_wgrad_img: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(max_wgs, n), dim3(512), lds, s, g);
  LVAE_LAUNCH_CHECK("conv_wgrad_img");
  return 0;
}

template <int KIND, int SPLIT>
static int wgi_launch_xv(const WgImgGroup& g, int n, int max_wgs, size_t lds, int xv, hipStream_t s) {
  if (KIND == 1) return wgi_launch<KIND, SPLIT, 2>(g, n, max_wgs, lds, s);   // no ring: the patch is the tile
  switch (xv) {
    case 0: return wgi_launch<KIND, SPLIT, 2>(g, n, max_wgs, lds, s);
    case 1: return wgi_launch<KIND, SPLIT, 4>(g, n, max_wgs, lds, s);
    case 2: return wgi_launch<KIND, SPLIT, 5>(g, n, max_wgs, lds, s);
    default: return wgi_launch<KIND, SPLIT, 8>(g, n, max_wgs, lds, s);
  }
}

// n <= kWgImgMax descriptors of one kind (conv_wgrad_img_kind), each with its own workspace: one launch + grouped fixed-order reduces.
// Returns -1000 when a descriptor is not eligible.
int conv_wgrad_img_grouped(const lvae_conv_desc* const* ds, const float* const* dy, float* const* dw, float* const* db,
                           void* const* workspace, int n, int kind, hipStream_t s) {
  WgImgGroup g;
  ReduceArgs r[kWgImgMax];
  int max_wgs = 0;
  size_t lds = 0;
  const int split = (kind >> 1) & 1 ? 1 : 3;
  for (int i = 0; i < n; ++i) {
    WgImgProb& a = g.p[i];
    if (wgi_plan(ds[i], a) != kind || !al16i(dy[i]) || !al16i(workspace[i])) return -1000;
    const int ntap = ds[i]->KH * ds[i]->KW;
    a.x = ds[i]->x;
    a.dy = dy[i];
    a.in_scale = ds[i]->in_scale;
    a.in_shift = ds[i]->in_shift;
    a.slab_w = static_cast<float*>(workspace[i]);
    a.slab_b = db[i] ? a.slab_w + (size_t)a.nwg * ntap * a.Cin * a.Cout : nullptr;
    if (a.nwg > max_wgs) max_wgs = a.nwg;
    const int ldd = (kind & 1) ? WGI_LDD1 : WGI_LDK;
    size_t l = (size_t)split * ((size_t)a.halo_px * WGI_LDK + 64 * ldd) * 2;
    if (l < 16 * 128 * 4) l = 16 * 128 * 4;   // bias-gradient reduction
    if (l > lds) lds = l;
    r[i] = ReduceArgs{a.slab_w, a.slab_b, a.nwg, ntap, a.Cin, a.Cout, ds[i]->w_stap, ds[i]->w_sk, ds[i]->w_sn, dw[i], db[i]};
  }
  for (int i = n; i < kWgImgMax; ++i) g.p[i] = g.p[0];
  const int xv = kind >> 2;
  int rc;
  if (kind & 1) rc = split == 1 ? wgi_launch_xv<1, 1>(g, n, max_wgs, lds, xv, s) : wgi_launch_xv<1, 3>(g, n, max_wgs, lds, xv, s);
  else rc = split == 1 ? wgi_launch_xv<0, 1>(g, n, max_wgs, lds, xv, s) : wgi_launch_xv<0, 3>(g, n, max_wgs, lds, xv, s);
  if (rc) return rc;
  for (int i0 = 0; i0 < n; i0 += 12) {
    wgrad_reduce_grouped_launch(r + i0, n - i0 < 12 ? n - i0 : 12, s);
    LVAE_LAUNCH_CHECK("conv_wgrad_img_reduce");
  }
  return 0;
}

// one gradient: the grouped launch with one problem (the same kernel, tiling and summation order: bitwise equal to the grouped call)
int conv_wgrad_img_try(const lvae_conv_desc* d, const float* dy, float* dw, float* db, void* workspace, hipStream_t s) {
  const int kind = conv_wgrad_img_kind(d);
  if (kind < 0) return -1000;
  return conv_wgrad_img_grouped(&d, &dy, &dw, &db, &workspace, 1, kind, s);
}

}  // namespace lvae
