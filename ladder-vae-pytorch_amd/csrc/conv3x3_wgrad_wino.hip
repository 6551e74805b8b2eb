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
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int W = 2 * TPR, HW2 = W + 2, TR = 16 / TPR, HROWS = 2 * TR + 2;
  constexpr int HP = HROWS * HW2;                      // halo pixels of a chunk
  constexpr int XSLOTS = (HP * 8 + 511) / 512;         // float4 slots per thread (32 channels per pixel)
  constexpr int BUF = HP * WG_XS + 64 * WG_DS;         // floats per LDS buffer
  const lvae_conv_desc& d = a.d;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5, pi = wave & 3, kg = wave >> 2;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int cib = bid & 1, cog = (bid >> 1) % a.ncog, range = (bid >> 1) / a.ncog;
  const int c_begin = range * a.cpr;
  const int c_end = min(a.total_chunks, c_begin + a.cpr);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- staging plan: x halo slots (pixel, float4 of this block's 32 channels), two dy slots (pixel, float4 of 64 channels)
  const int xc4 = (t & 7) * 4;
  int xrel[XSLOTS], xhy[XSLOTS];
  bool xcol_ok[XSLOTS], xin[XSLOTS];
#pragma unroll
  for (int u = 0; u < XSLOTS; ++u) {
    const int px = (t >> 3) + 64 * u;
    const int hy = px / HW2, hx = px - hy * HW2;
    xin[u] = px < HP;
    xhy[u] = hy;
    xcol_ok[u] = (hx >= 1) & (hx <= W);
    xrel[u] = (hy * W + hx) * 64 + cib * 32 + xc4;  // relative to the pixel (row oh0 - 1, column -1) of the image
  }
  const int dpx = t >> 4, dc4 = (t & 15) * 4;     // second slot: pixel + 32
  const int drel = dpx * d.Cout + cog * 64 + dc4;  // dy chunk rows are contiguous in memory (full image width)
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = zero4;
  if (d.in_scale) {
    sc = *reinterpret_cast<const f32x4*>(d.in_scale + cib * 32 + xc4);
    sh = *reinterpret_cast<const f32x4*>(d.in_shift + cib * 32 + xc4);
  }
  f32x4 xreg[XSLOTS], dreg[2];
  unsigned xok = 0;
  auto load_chunk = [&](int c) {
    const int n = fastdiv(c, a.m_cpi), cr = c - n * a.cpi;
    const int oh0 = cr * (2 * TR);
    const float* xb = d.x + ((int64_t)(n * d.H + oh0 - 1) * W - 1) * 64;
    xok = 0;
#pragma unroll
    for (int u = 0; u < XSLOTS; ++u) {
      const int ih = oh0 - 1 + xhy[u];
      const bool ok = xin[u] & xcol_ok[u] & ((unsigned)ih < (unsigned)d.H);
      const float* p = ok ? xb + xrel[u] : d.x;
      xreg[u] = *reinterpret_cast<const f32x4*>(p);
      xok |= ok ? (1u << u) : 0u;
    }
    const float* dyb = a.dy + (int64_t)(n * d.H + oh0) * W * d.Cout + drel;
    dreg[0] = *reinterpret_cast<const f32x4*>(dyb);
    dreg[1] = *reinterpret_cast<const f32x4*>(dyb + 32 * d.Cout);
  };
  auto store_chunk = [&](float* buf) {
#pragma unroll
    for (int u = 0; u < XSLOTS; ++u) {
      if (xin[u]) {
        f32x4 v = zero4;
        if ((xok >> u) & 1u) {
          v = xreg[u];
          if (d.in_scale) v = act_fwd4(v * sc + sh, d.in_act);
        }
        *reinterpret_cast<f32x4*>(buf + ((t >> 3) + 64 * u) * WG_XS + xc4) = v;
      }
    }
    *reinterpret_cast<f32x4*>(buf + HP * WG_XS + dpx * WG_DS + dc4) = dreg[0];
    *reinterpret_cast<f32x4*>(buf + HP * WG_XS + (dpx + 32) * WG_DS + dc4) = dreg[1];
  };

  // ---- per-lane read bases: tile = 8*kg + 2*ss + lh of the chunk
  // B^T rows: [1,0,-1,0], [0,1,1,0], [0,-1,1,0], [0,1,0,-1]  ->  t = d[ra] + sgn * d[rb]
  const int ra = pi == 0 ? 0 : (pi == 2 ? 2 : 1);
  const int rb = pi == 0 ? 2 : (pi == 1 ? 2 : (pi == 2 ? 1 : 3));
  const float sgn = pi == 1 ? 1.f : -1.f;
  // row i of A (4x2) = [[1,0],[1,1],[1,-1],[0,-1]]  ->  s = alpha * dy[0] + beta * dy[1]
  const float alpha = pi == 3 ? 0.f : 1.f;
  const float beta = pi == 0 ? 0.f : (pi == 1 ? 1.f : -1.f);
  int xbase_px, dbase_px;
  if (TPR == 4) {
    xbase_px = 4 * kg * HW2 + 2 * lh;
    dbase_px = 4 * kg * W + 2 * lh;
  } else if (TPR == 8) {
    xbase_px = 2 * kg * HW2 + 2 * lh;
    dbase_px = 2 * kg * W + 2 * lh;
  } else {
    xbase_px = 16 * kg + 2 * lh;
    dbase_px = 16 * kg + 2 * lh;
  }
  const int xoff_a = (xbase_px + ra * HW2) * WG_XS + li;
  const int xoff_b = (xbase_px + rb * HW2) * WG_XS + li;
  const int doff = HP * WG_XS + dbase_px * WG_DS + li;

  f32x16 acc[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][h][r] = 0.f;
  float dbacc[2] = {0.f, 0.f};

  if (c_begin < c_end) {
    load_chunk(c_begin);
    store_chunk(smem);
    if (c_begin + 1 < c_end) load_chunk(c_begin + 1);
  }
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    const float* buf = smem + ((c - c_begin) & 1) * BUF;
    const float* pa = buf + xoff_a;
    const float* pb = buf + xoff_b;
    const float* pd = buf + doff;
    // LDS reads of k-step ss+1 are issued before the MFMAs of k-step ss (their latency hides under 512 MFMA cycles)
    float rxa[4], rxb[4], ry[2][4];
    auto read_step = [&](int ss) {
      const int cx = TPR == 4 ? 2 * (ss >> 1) * HW2 + 4 * (ss & 1) : 4 * ss;  // pixels
      const int cd = TPR == 4 ? 2 * (ss >> 1) * W + 4 * (ss & 1) : 4 * ss;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        rxa[cc] = pa[(cx + cc) * WG_XS];
        rxb[cc] = pb[(cx + cc) * WG_XS];
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float* q = pd + cd * WG_DS + 32 * h;
        ry[h][0] = q[0];
        ry[h][1] = q[WG_DS];
        ry[h][2] = q[W * WG_DS];
        ry[h][3] = q[(W + 1) * WG_DS];
      }
    };
    read_step(0);
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
      float tt[4], vv[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) tt[cc] = __builtin_fmaf(sgn, rxb[cc], rxa[cc]);
      vv[0] = tt[0] - tt[2];
      vv[1] = tt[1] + tt[2];
      vv[2] = tt[2] - tt[1];
      vv[3] = tt[1] - tt[3];
      float ww[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float s0 = __builtin_fmaf(beta, ry[h][2], alpha * ry[h][0]), s1 = __builtin_fmaf(beta, ry[h][3], alpha * ry[h][1]);
        ww[h][0] = s0;
        ww[h][1] = s0 + s1;
        ww[h][2] = s0 - s1;
        ww[h][3] = -s1;
        dbacc[h] += ww[h][1];  // position (1,1) of A dY A^T is the plain sum of the 2x2 block (used from the pi == 1 waves)
      }
      __builtin_amdgcn_sched_barrier(0);
      if (ss + 1 < 4) read_step(ss + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[j][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[j], ww[h][j], acc[j][h], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (c + 1 < c_end) {
      store_chunk(smem + ((c + 1 - c_begin) & 1) * BUF);
      if (c + 2 < c_end) load_chunk(c + 2);
    }
    __syncthreads();
  }

  // ---- add the two tile halves through LDS (kg == 1 publishes, kg == 0 sums) and write the slab
  float* Rs = smem;  // [pi][j][h][r][64 lanes]
  if (kg == 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) Rs[(((pi * 4 + j) * 2 + h) * 16 + r) * 64 + lane] = acc[j][h][r];
    if (pi == 1) {
      Rs[4 * 4 * 2 * 16 * 64 + lane] = dbacc[0];
      Rs[4 * 4 * 2 * 16 * 64 + 64 + lane] = dbacc[1];
    }
  }
  __syncthreads();
  if (kg == 0) {
    float* slab = a.slab_w + (size_t)((range * 2 + cib) * a.ncog + cog) * 32 * 16 * 64 + 4 * pi * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ci = (r & 3) + 8 * (r >> 2) + 4 * lh;
          slab[((size_t)ci * 16 + j) * 64 + 32 * h + li] = acc[j][h][r] + Rs[(((pi * 4 + j) * 2 + h) * 16 + r) * 64 + lane];
        }
    if (pi == 1 && cib == 0 && a.slab_b) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v = dbacc[h] + Rs[4 * 4 * 2 * 16 * 64 + 64 * h + lane];
        v += __shfl_xor(v, 32, 64);
        if (lh == 0) a.slab_b[(size_t)(range * a.ncog + cog) * 64 + 32 * h + li] = v;
      }
    }
  }
}

// dw[tap(a,b)][ci][co] += (G^T (sum_ranges slab) G)[a][b]; one 1024-thread workgroup per (ci, group of 64 co): threads =
// 16 float4 columns x 16 positions x 4 range slices; each range contributes one contiguous 4 KB block; fixed summation order.
__global__ __launch_bounds__(1024) void conv_wgrad_wino_reduce_kernel(const float* __restrict__ slab_w, const float* __restrict__ slab_b,
                                                                       int nranges, int ncog, int64_t stap, int64_t sk, int64_t sn,
                                                                       float* dw, float* db) {
  __shared__ float red[4][16][64];
  __shared__ float red_b[16][64];
  const int ci = blockIdx.x / ncog, cog = blockIdx.x % ncog;
  const int t = threadIdx.x, q = t & 15, p = (t >> 4) & 15, rs = t >> 8;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const float* src = slab_w + ((((size_t)(ci >> 5) * ncog + cog) * 32 + (ci & 31)) * 16 + p) * 64 + q * 4;
  const size_t rstride = (size_t)2 * ncog * 32 * 16 * 64;
#pragma unroll 8
  for (int r = rs; r < nranges; r += 4) s += *reinterpret_cast<const f32x4*>(src + r * rstride);
  *reinterpret_cast<f32x4*>(&red[rs][p][q * 4]) = s;
  const bool do_b = db != nullptr && ci == 0;
  if (do_b) {
    const int co = t & 63, slice = t >> 6;
    float v = 0.f;
#pragma unroll 4
    for (int r = slice; r < nranges; r += 16) v += slab_b[(size_t)(r * ncog + cog) * 64 + co];
    red_b[slice][co] = v;
  }
  __syncthreads();
  if (t < 192) {
    const int co = t & 63, ga = t >> 6;  // output row a of G^T dU G
    float u[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) u[i][j] = (red[0][i * 4 + j][co] + red[1][i * 4 + j][co]) + (red[2][i * 4 + j][co] + red[3][i * 4 + j][co]);
    // G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
    float ra[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      ra[j] = ga == 0 ? u[0][j] + 0.5f * (u[1][j] + u[2][j]) : (ga == 1 ? 0.5f * (u[1][j] - u[2][j]) : 0.5f * (u[1][j] + u[2][j]) + u[3][j]);
    const float g0 = ra[0] + 0.5f * (ra[1] + ra[2]), g1 = 0.5f * (ra[1] - ra[2]), g2 = 0.5f * (ra[1] + ra[2]) + ra[3];
    float* o = dw + (int64_t)ci * sk + (int64_t)(cog * 64 + co) * sn + (int64_t)(ga * 3) * stap;
    o[0] += g0;
    o[stap] += g1;
    o[2 * stap] += g2;
  }
  if (do_b && t >= 256 && t < 320) {
    const int co = t - 256;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += red_b[k][co];
    db[cog * 64 + co] += v;
  }
}

static bool al16g(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static bool wg_wino_plan(const lvae_conv_desc* d, WgWinoArgs& a) {
  static const bool off = getenv("LVAE_DISABLE_WINO_WGRAD") != nullptr || getenv("LVAE_DISABLE_WINO") != nullptr;  // A/B switch
  if (off) return false;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->gather != LVAE_GATHER_CONV) return false;
  if (d->C1 != 64 || d->C2 != 0 || d->x2 != nullptr || d->Cout % 64 != 0 || d->Cout > 256) return false;
  if (d->OH != d->H || d->OW != d->W || (d->H & 1)) return false;
  if (d->W != 8 && d->W != 16 && d->W != 32) return false;
  const int tpr = d->W / 2, tr = 16 / tpr;
  if ((d->H / 2) % tr != 0) return false;
  if (!al16g(d->x) || !al16g(d->in_scale) || !al16g(d->in_shift)) return false;
  const int64_t M = (int64_t)d->N * d->H * d->W;
  static const int64_t min_m = getenv("LVAE_WINO_WGRAD_MIN_M") ? atoll(getenv("LVAE_WINO_WGRAD_MIN_M")) : 256 * 64;  // tuning switch
  if (M < min_m || M * 256 >= ((int64_t)1 << 31)) return false;
  a.ncog = d->Cout / 64;
  a.cpi = (d->H / 2) / tr;
  a.total_chunks = d->N * a.cpi;
  int nranges = 128 / a.ncog;  // 256 workgroups with the two input-channel blocks
  if (nranges < 1) nranges = 1;
  static const int min_cpr = getenv("LVAE_WINO_WGRAD_MIN_CPR") ? atoi(getenv("LVAE_WINO_WGRAD_MIN_CPR")) : 4;  // tuning switch
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
  static bool attr_set = false;
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
  hipLaunchKernelGGL(conv_wgrad_wino_reduce_kernel, dim3(64 * a.ncog), dim3(1024), 0, s, a.slab_w, a.slab_b, a.nranges,
                     a.ncog, d->w_stap, d->w_sk, d->w_sn, dw, db);
  LVAE_LAUNCH_CHECK("conv_wgrad_wino_reduce");
  return 0;
}

}  // namespace lvae
