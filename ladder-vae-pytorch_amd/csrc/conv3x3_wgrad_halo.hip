// Weight gradient of the 3x3 / stride 1 / pad 1 convolution with the input patch resident in LDS.
//
//   dW[kh][kw][ci][co] = sum_pixels T(x)[pixel + (kh-1, kw-1)][ci] * dy[pixel][co]
//
// A workgroup owns ONE kernel row kh (3 taps) and a strided subset of the pixel tiles. Per tile it stages the
// (TH+2) x (W+2) x Cin halo patch of T(x) (BatchNorm-apply + ELU recomputed once per element) and the
// 128 x 64 dy tile in LDS, then runs 3 MFMA chains (one per kw) over the tile's pixels: the A fragment of tap kw is
// the same LDS image read one pixel to the right, so x is fetched from HBM/L2 once per kernel row instead of once
// per tap. Partials go to the split-K slab of conv_wgrad.hip and are summed by wgrad_reduce_kernel (deterministic).
#include "lvae_common.h"

namespace lvae {

struct WHaloArgs {
  lvae_conv_desc d;
  const float* dy;
  float* slab_w;  // [ksplit][9][Cin][Cout]
  float* slab_b;  // [ksplit][Cout] or null
  int TH, TW, NI, tiles_h, halo_w, halo_h, halo_px, ntiles, ksplit, Cin, ncot;
};

template <int CIN_T>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_halo_kernel(WHaloArgs a) {
  constexpr int CIN4 = CIN_T / 4;
  constexpr int BMP = 128;  // pixels per tile (upper bound; tile_px <= BMP)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                                   // [halo_px][CIN_T]
  float* Ys = smem + (size_t)a.halo_px * CIN_T;       // [BMP][64]
  const lvae_conv_desc& d = a.d;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wci = wave >> 1, wco = wave & 1;
  int bid = blockIdx.x;
  const int kh = bid % 3;
  bid /= 3;
  const int cot = bid % a.ncot;
  const int ks = bid / a.ncot;
  const int co0 = cot * 64;
  const int Cin = a.Cin;
  const int tile_px = a.NI * a.TH * a.TW;
  const int per_img = a.halo_h * a.halo_w;
  const bool do_bias = a.slab_b != nullptr && kh == 0;

  f32x16 acc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

  const int total = a.halo_px * CIN4;
  for (int tile = ks; tile < a.ntiles; tile += a.ksplit) {
    const int th_idx = tile % a.tiles_h, ig = tile / a.tiles_h;
    const int n0 = ig * a.NI, oh0 = th_idx * a.TH;
    // ---- global -> registers: the 8 dy float4 and the first 8 halo float4 of this thread are all in flight together
    f32x4 yreg[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = t + 256 * u, p = idx >> 4, c4 = (idx & 15) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p < tile_px) {
        const int img = p / (a.TH * a.TW), r = p - img * (a.TH * a.TW);
        const int ty = r / a.TW, tx = r - ty * a.TW;
        const int n = n0 + img;
        if (n < d.N && co0 + c4 < d.Cout)
          v = *reinterpret_cast<const f32x4*>(a.dy + ((size_t)(n * d.H + oh0 + ty) * d.W + tx) * d.Cout + co0 + c4);
      }
      yreg[u] = v;
    }
    __syncthreads();  // previous tile's MFMAs are done with the LDS images
    for (int base = t; base < total; base += 256 * 8) {
      f32x4 v[8];
      int dst[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + 256 * u;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        dst[u] = -1;
        if (idx < total) {
          const int px = idx / CIN4, c4 = (idx - px * CIN4) * 4;
          const int img = px / per_img, r = px - img * per_img;
          const int hy = r / a.halo_w, hx = r - hy * a.halo_w;
          const int n = n0 + img, ih = oh0 + hy - 1, iw = hx - 1;
          dst[u] = px * CIN_T + c4;
          if (n < d.N && ih >= 0 && ih < d.H && iw >= 0 && iw < d.W && c4 < Cin) {
            v[u] = *reinterpret_cast<const f32x4*>(d.x + ((size_t)(n * d.H + ih) * d.W + iw) * Cin + c4);
            dst[u] |= 0x40000000;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (dst[u] >= 0 && (dst[u] & 0x40000000)) {
          dst[u] &= 0x3fffffff;
          if (d.in_scale) {
            const int c4 = dst[u] % CIN_T;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(d.in_scale + c4);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(d.in_shift + c4);
            f32x4 x = v[u] * sc + sh;
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = act_fwd(x[j], d.in_act);
            v[u] = x;
          }
        }
        if (dst[u] >= 0) *reinterpret_cast<f32x4*>(Xs + dst[u]) = v[u];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = t + 256 * u;
      if (do_bias) bsum += yreg[u];
      *reinterpret_cast<f32x4*>(Ys + (idx >> 4) * 64 + (idx & 15) * 4) = yreg[u];
    }
    __syncthreads();
    // ---- 3 taps x (tile_px / 2) MFMA steps; k index = pixel, lanes 32..63 take the odd pixel of each pair
    const int rows = a.NI * a.TH;
    for (int row = 0; row < rows; ++row) {
      const int img = row / a.TH, ty = row - img * a.TH;
      const float* xrow = Xs + (size_t)((img * a.halo_h + ty + kh) * a.halo_w + lh) * CIN_T + wci * 32 + li;
      const float* yrow = Ys + (size_t)(row * a.TW + lh) * 64 + wco * 32 + li;
#pragma unroll 4
      for (int tx = 0; tx < a.TW; tx += 2) {
        const float b = yrow[tx * 64];
        const float a0 = xrow[tx * CIN_T], a1 = xrow[(tx + 1) * CIN_T], a2 = xrow[(tx + 2) * CIN_T];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b, acc[2], 0, 0, 0);
      }
    }
  }

  // ---- slab [ks][tap][Cin][Cout]
  const int co = co0 + wco * 32 + li;
  if (co < d.Cout) {
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      float* slab = a.slab_w + ((size_t)ks * 9 + kh * 3 + kw) * Cin * d.Cout;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = wci * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ci < Cin) slab[(size_t)ci * d.Cout + co] = acc[kw][r];
      }
    }
  }
  if (do_bias) {
    __syncthreads();
    float* red = Ys;  // [16][64]
#pragma unroll
    for (int j = 0; j < 4; ++j) red[(t >> 4) * 64 + (t & 15) * 4 + j] = bsum[j];
    __syncthreads();
    if (t < 64) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += red[r * 64 + t];
      if (co0 + t < d.Cout) a.slab_b[(size_t)ks * d.Cout + co0 + t] = s;
    }
  }
}

static bool al16w(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static bool whalo_plan(const lvae_conv_desc* d, WHaloArgs& a) {
  const int Cin = d->C1;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->gather != LVAE_GATHER_CONV || d->x2 != nullptr ||
      d->OH != d->H || d->OW != d->W)
    return false;
  if (Cin > 64 || Cin % 4 != 0 || d->Cout % 4 != 0 || d->W % 2 != 0 || d->W > 128) return false;
  if (!al16w(d->x) || (d->in_scale && (!al16w(d->in_scale) || !al16w(d->in_shift)))) return false;
  const int cin_t = Cin <= 32 ? 32 : 64;
  int BM = 128;
  for (;;) {
    int TH = 1;
    for (int c = 1; c <= d->H; ++c)
      if (d->H % c == 0 && c * d->W <= BM) TH = c;
    int NI = BM / (TH * d->W);
    if (NI < 1) NI = 1;
    if (TH < d->H) NI = 1;
    if (NI > d->N) NI = d->N;
    a.TH = TH;
    a.TW = d->W;
    a.NI = NI;
    a.tiles_h = d->H / TH;
    a.halo_h = TH + 2;
    a.halo_w = d->W + 2;
    a.halo_px = NI * a.halo_h * a.halo_w;
    const size_t lds = ((size_t)a.halo_px * cin_t + 128 * 64) * sizeof(float);
    if (lds <= 160 * 1024) break;
    if (BM == 32) return false;
    BM /= 2;
  }
  a.Cin = Cin;
  a.ntiles = ((d->N + a.NI - 1) / a.NI) * a.tiles_h;
  a.ksplit = a.ntiles < 128 ? a.ntiles : 128;
  a.ncot = (d->Cout + 63) / 64;
  return true;
}

// workspace floats needed by the halo path, or 0 when the descriptor is not eligible
size_t conv3x3_wgrad_halo_workspace(const lvae_conv_desc* d) {
  WHaloArgs a;
  if (!whalo_plan(d, a)) return 0;
  return ((size_t)a.ksplit * (9 * (size_t)a.Cin * d->Cout + d->Cout)) * sizeof(float);
}

void wgrad_reduce_launch(const float* slab_w, const float* slab_b, int ksplit, int ntaps, int Cin, int Cout, int64_t stap,
                         int64_t sk, int64_t sn, float* dw, float* db, hipStream_t s);

// returns -1000 when not eligible
int conv3x3_wgrad_halo_try(const lvae_conv_desc* d, const float* dy, float* dw, float* db, void* workspace, hipStream_t s) {
  WHaloArgs a;
  if (!whalo_plan(d, a) || !al16w(dy)) return -1000;
  a.d = *d;
  a.dy = dy;
  a.slab_w = static_cast<float*>(workspace);
  a.slab_b = db ? a.slab_w + (size_t)a.ksplit * 9 * a.Cin * d->Cout : nullptr;
  const int cin_t = a.Cin <= 32 ? 32 : 64;
  const size_t lds = ((size_t)a.halo_px * cin_t + 128 * 64) * sizeof(float);
  auto k64 = conv3x3_wgrad_halo_kernel<64>;
  auto k32 = conv3x3_wgrad_halo_kernel<32>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k64), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(k32), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      set_error("conv3x3_wgrad_halo: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  const int grid = 3 * a.ncot * a.ksplit;
  if (cin_t == 64) hipLaunchKernelGGL(k64, dim3(grid), dim3(256), lds, s, a);
  else hipLaunchKernelGGL(k32, dim3(grid), dim3(256), lds, s, a);
  LVAE_LAUNCH_CHECK("conv3x3_wgrad_halo");
  wgrad_reduce_launch(a.slab_w, a.slab_b, a.ksplit, 9, a.Cin, d->Cout, d->w_stap, d->w_sk, d->w_sn, dw, db, s);
  LVAE_LAUNCH_CHECK("conv2d_wgrad_reduce");
  return 0;
}

}  // namespace lvae
