// Weight gradient of stride-1 "same" convolutions (3x3 pad 1, 1x1 pad 0): LDS-resident operands, wave-specialised.
//
//   dW[kh][kw][ci][co] = sum_pixels T(x)[pixel + (kh-p, kw-p)][ci] * dy[pixel][co]
//
// A workgroup owns ALL taps, one 64-wide co tile and a strided subset of the pixel tiles (so the x patch is fetched and
// transformed once, not once per kernel row); ONE workgroup per CU (the grid is sized to the CU count, which also bounds the split-K
// slab traffic). It has 8 waves with fixed roles:
//   * waves 4-7 (loaders): global -> registers -> LDS. They stage the (TH+2p) x (W+2p) x Cin patch of T(x)
//     (BatchNorm-apply + ELU recomputed once per element, channel-concat inputs read from two tensors, zero padding
//     materialised) and the dy tile of the NEXT tile into the idle LDS buffer;
//   * waves 0-3 (MFMA): one v_mfma_f32_32x32x2 chain per (kw, 64-channel ci block) over the CURRENT buffer. The A
//     fragment of tap kw is the same LDS image read kw pixels to the right, so x is fetched once per kernel row.
// One workgroup barrier per tile swaps the buffers, so a tile costs max(load, MFMA) instead of their sum (measured
// with phase-skip builds: in the single-role version the two phases added up, ~150 + ~175 us on a 32x32x64 layer).
// dy is stored with the row pitch of the x image (gap columns stay zero and contribute nothing), which makes the k-walk
// over an image linear and branch-free; the LDS reads of step s+1 are issued before the MFMAs of step s.
// Partials go to split-K slabs summed in a fixed order by wgrad_reduce_kernel (deterministic, no float atomics).
//
// Barrier protocol (T = tiles of this workgroup >= 1; both roles execute exactly T + 3 barriers):
//   all: B0 after zero-init | loaders: fill(i); bar  for i < T; bar; [bias partials]; bar
//                           | MFMA   : bar; compute(i); bar for i < T;       [slabs]; bar
#include <stdlib.h>

#include "lvae_common.h"

namespace lvae {

struct WTileArgs {
  lvae_conv_desc d;
  const float* dy;
  float* slab_w;  // [ksplit][KH*KW][Cin][Cout]
  float* slab_b;  // [ksplit][Cout] or null
  int TH, TW, NI, tiles_h, halo_w, halo_h, halo_px, ntiles, ksplit, Cin, ncot, pad, buf_floats;
  uint32_t m_thw, m_tw, m_per_img, m_halo_w, m_tiles_h;  // fastdiv magics
  int debug;  // profiling only: 1 = loaders idle, 2 = MFMA waves idle, 4 = loaders skip the x patch, 8 = loaders skip dy
};

template <int CIN_T, int NKW>
__global__ __launch_bounds__(512, 2) void conv_wgrad_ws_kernel(WTileArgs a) {
  constexpr int CIN4 = CIN_T / 4;
  constexpr int CIB = CIN_T > 64 ? CIN_T / 64 : 1;  // 64-channel ci blocks
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const lvae_conv_desc& d = a.d;
  const int tid = threadIdx.x;
  const bool loader = tid >= 256;
  const int t = tid & 255, lane = t & 63, wave = t >> 6;
  const int grp = blockIdx.x;
  const int cot = grp % a.ncot;
  const int ks = grp / a.ncot;
  const int co0 = cot * 64;
  const int Cin = a.Cin;
  const int tile_px = a.NI * a.TH * a.TW;
  const int per_img = a.halo_h * a.halo_w;
  const int x_f4 = a.halo_px * CIN4;          // float4 in the x image
  const int y_off = a.halo_px * CIN_T;        // dy image offset inside a buffer (floats)
  const bool do_bias = a.slab_b != nullptr;
  const int my_tiles = (a.ntiles - ks + a.ksplit - 1) / a.ksplit;  // >= 1: ksplit <= ntiles

  // zero both buffers once: the border / gap entries are never written again
  for (int i = tid; i < (2 * a.buf_floats) / 4; i += 512) *reinterpret_cast<f32x4*>(smem + i * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();  // B0

  if (loader) {
    // ------------------------------------------------------------------------------------------------ loaders
    const int c4x = (t % CIN4) * 4;  // 256 % CIN4 == 0: the channel group of a thread is the same for every element it stages
    const bool cx_ok = c4x < Cin;
    const bool first = c4x < d.C1 || d.x2 == nullptr;
    const float* xsrc = first ? d.x : d.x2;
    const int xcs = first ? d.C1 : d.C2, xco = first ? c4x : c4x - d.C1;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (d.in_scale && cx_ok) {
      sc = *reinterpret_cast<const f32x4*>(d.in_scale + c4x);
      sh = *reinterpret_cast<const f32x4*>(d.in_shift + c4x);
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 bsum = zero4;
    for (int it = 0; it < my_tiles; ++it) {
      const int tile = ks + it * a.ksplit;
      const int ig = fastdiv(tile, a.m_tiles_h), th_idx = tile - ig * a.tiles_h;
      const int n0 = ig * a.NI, oh0 = th_idx * a.TH;
      float* Xs = smem + (it & 1) * a.buf_floats;
      float* Ys = Xs + y_off;
      // ONE batch per tile: up to 16 float4 of the x patch and 8 of dy per thread are all in flight before the first
      // use (a dependent second batch costs another ~3 us round trip under load). Every lane loads from a clamped valid
      // address, so there are no exec-mask regions around the loads.
      for (int xbase = t; (xbase < x_f4 || xbase == t) && !(a.debug & 1); xbase += 256 * 16) {  // first pass: every thread (dy)
        const bool with_y = xbase == t && !(a.debug & 8);
        f32x4 xv[16], yv[8];
        unsigned xok = 0, yok = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int idx = xbase + 256 * u;
          const int px = idx / CIN4;
          const int img = fastdiv(px, a.m_per_img), r = px - img * per_img;
          const int hy = fastdiv(r, a.m_halo_w), hx = r - hy * a.halo_w;
          const int n = n0 + img, ih = oh0 + hy - a.pad, iw = hx - a.pad;
          const bool ok = (idx < x_f4) & (n < d.N) & ((unsigned)ih < (unsigned)d.H) & ((unsigned)iw < (unsigned)d.W) & cx_ok;
          const unsigned off = ok ? (unsigned)(((n * d.H + ih) * d.W + iw) * xcs + xco) : 0u;  // < 2^31 floats (host check)
          xv[u] = *reinterpret_cast<const f32x4*>(xsrc + off);
          xok |= ok ? (1u << u) : 0u;
        }
        if (with_y) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int idx = t + 256 * u, p = idx >> 4, c4 = (idx & 15) * 4;
            const int img = fastdiv(p, a.m_thw), r = p - img * (a.TH * a.TW);
            const int n = n0 + img;
            const bool ok = (p < tile_px) & (n < d.N) & (co0 + c4 < d.Cout);
            const unsigned off = ok ? (unsigned)(((n * d.H + oh0) * d.W + r) * d.Cout + co0 + c4) : 0u;
            yv[u] = *reinterpret_cast<const f32x4*>(a.dy + off);
            yok |= ok ? (1u << u) : 0u;
          }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int idx = xbase + 256 * u;
          if (idx < x_f4) {
            f32x4 w = zero4;  // zero padding is inserted AFTER BatchNorm + activation
            if ((xok >> u) & 1u) {
              w = xv[u];
              if (d.in_scale) w = act_fwd4(w * sc + sh, d.in_act);
            }
            *reinterpret_cast<f32x4*>(Xs + idx * 4) = w;
          }
        }
        if (with_y) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int idx = t + 256 * u, p = idx >> 4;
            if (p < tile_px) {
              const f32x4 w = ((yok >> u) & 1u) ? yv[u] : zero4;
              if (do_bias) bsum += w;
              const int row = fastdiv(p, a.m_tw), c = p - row * a.TW;
              *reinterpret_cast<f32x4*>(Ys + (size_t)(row * a.halo_w + c) * 64 + (idx & 15) * 4) = w;
            }
          }
        }
      }
      __syncthreads();  // buffer (it & 1) is published; the MFMA waves have finished tile it-1
    }
    __syncthreads();    // the MFMA waves have finished the last tile
    if (do_bias) {
      // a loader thread always holds the same 4 columns ((t & 15) * 4): reduce over the 16 row groups through LDS
#pragma unroll
      for (int j = 0; j < 4; ++j) smem[(t >> 4) * 64 + (t & 15) * 4 + j] = bsum[j];
    }
    __syncthreads();
    if (do_bias && t < 64) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += smem[r * 64 + t];
      if (co0 + t < d.Cout) a.slab_b[(size_t)ks * d.Cout + co0 + t] = s;
    }
    return;
  }

  // ---------------------------------------------------------------------------------------------------- MFMA waves
  const int li = lane & 31, lh = lane >> 5;
  const int wci = wave >> 1, wco = wave & 1;
  constexpr int NT = NKW * NKW;  // taps handled by this workgroup: all of them (x is staged and transformed once)
  f32x16 acc[NT][CIB];
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int c = 0; c < CIB; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[k][c][r] = 0.f;

  const int steps = (a.TH * a.halo_w) >> 1;
  // the loader wave on the same SIMD issues dense VALU / memory work: let the MFMA wave win issue arbitration
  __builtin_amdgcn_s_setprio(3);
  __syncthreads();  // tile 0 is staged
  for (int it = 0; it < my_tiles; ++it) {
    const float* Xs = smem + (it & 1) * a.buf_floats;
    const float* Ys = Xs + y_off;
    for (int img = 0; img < a.NI && !(a.debug & 2); ++img) {
      const float* xr = Xs + (size_t)(img * a.halo_h * a.halo_w + lh) * CIN_T + wci * 32 + li;
      const int rowp = a.halo_w * CIN_T;  // one halo row down = next kernel row
      const float* yr = Ys + (size_t)(img * a.TH * a.halo_w + lh) * 64 + wco * 32 + li;
      float b0, b1, a0[NT][CIB], a1[NT][CIB];
      auto fetch = [&](int s2, float (&A)[NT][CIB], float& B) {
        B = yr[s2 * 128];
#pragma unroll
        for (int k = 0; k < NT; ++k)
#pragma unroll
          for (int c = 0; c < CIB; ++c) A[k][c] = xr[(k / NKW) * rowp + (s2 * 2 + k % NKW) * CIN_T + c * 64];
      };
      auto mma = [&](const float (&A)[NT][CIB], float B) {
#pragma unroll
        for (int k = 0; k < NT; ++k)
#pragma unroll
          for (int c = 0; c < CIB; ++c) acc[k][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[k][c], B, acc[k][c], 0, 0, 0);
      };
      // two register sets; sched_barrier(0) pins "reads of s+1 before MFMAs of s" (hipcc otherwise sinks the reads next
      // to their use and waits lgkmcnt(0) there). The look-ahead of the last pair reads <= 6 pixels past the image:
      // inside the LDS allocation (slack added by the launcher), never used.
      fetch(0, a0, b0);
      int s2 = 0;
      for (; s2 + 2 <= steps; s2 += 2) {
        fetch(s2 + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(s2 + 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (s2 < steps) mma(a0, b0);
    }
    __syncthreads();  // done with buffer (it & 1); tile it+1 is staged in the other one
  }

  // ---- slab [ks][tap][Cin][Cout]
  const int co = co0 + wco * 32 + li;
  if (co < d.Cout) {
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      float* slab = a.slab_w + ((size_t)ks * NT + k) * Cin * d.Cout;
#pragma unroll
      for (int c = 0; c < CIB; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ci = c * 64 + wci * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (ci < Cin) slab[(size_t)ci * d.Cout + co] = acc[k][c][r];
        }
    }
  }
  __syncthreads();  // pairs with the loaders' bias-reduction barrier
}

static bool al16w(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int cin_tile(int Cin) { return Cin <= 32 ? 32 : (Cin <= 64 ? 64 : 128); }

static bool wtile_plan(const lvae_conv_desc* d, WTileArgs& a) {
  const int Cin = d->C1 + d->C2;
  const bool k3 = d->KH == 3 && d->KW == 3 && d->pad == 1, k1 = d->KH == 1 && d->KW == 1 && d->pad == 0;
  if (!(k3 || k1) || d->stride != 1 || d->gather != LVAE_GATHER_CONV || d->OH != d->H || d->OW != d->W) return false;
  if ((int64_t)d->N * d->H * d->W * (Cin > d->Cout ? Cin : d->Cout) >= ((int64_t)1 << 31)) return false;  // 32-bit element offsets
  if (Cin > 128 || (k3 && Cin > 64) || d->C1 % 4 != 0 || d->C2 % 4 != 0 || d->Cout % 4 != 0 || d->W % 2 != 0) return false;
  if (!al16w(d->x) || (d->x2 && !al16w(d->x2)) || (d->in_scale && (!al16w(d->in_scale) || !al16w(d->in_shift)))) return false;
  const int cin_t = cin_tile(Cin), pad = k3 ? 1 : 0;
  // tile: as many pixels as two LDS buffers allow (at most 128), whole rows, whole images when several fit
  int BM = 128;
  for (;;) {
    if (d->W <= BM) {
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
      a.halo_h = TH + 2 * pad;
      a.halo_w = d->W + 2 * pad;
      a.halo_px = NI * a.halo_h * a.halo_w;
      a.buf_floats = a.halo_px * cin_t + NI * TH * a.halo_w * 64 + 512;  // + look-ahead slack (dy side; the x side
                                                                            // look-ahead lands in the dy image)
      a.buf_floats = (a.buf_floats + 3) / 4 * 4;
      if ((size_t)2 * a.buf_floats * sizeof(float) <= 160 * 1024 && a.halo_px < 65536) break;
    }
    if (BM == 16) return false;
    BM /= 2;
  }
  a.pad = pad;
  a.Cin = Cin;
  a.m_thw = fastdiv_magic(a.TH * a.TW);
  a.m_tw = fastdiv_magic(a.TW);
  a.m_per_img = fastdiv_magic(a.halo_h * a.halo_w);
  a.m_halo_w = fastdiv_magic(a.halo_w);
  a.m_tiles_h = fastdiv_magic(a.tiles_h);
  a.ntiles = ((d->N + a.NI - 1) / a.NI) * a.tiles_h;
  a.ncot = (d->Cout + 63) / 64;
  int ks = 256 / a.ncot;  // one workgroup per CU: (co tiles) x ksplit ~ 256
  if (ks < 1) ks = 1;
  a.ksplit = a.ntiles < ks ? a.ntiles : ks;
  return a.ntiles < 65536;
}

// workspace bytes needed by this path, or 0 when the descriptor is not eligible
size_t conv_wgrad_tile_workspace(const lvae_conv_desc* d) {
  WTileArgs a;
  if (!wtile_plan(d, a)) return 0;
  return ((size_t)a.ksplit * ((size_t)d->KH * d->KW * a.Cin * d->Cout + d->Cout)) * sizeof(float);
}

void wgrad_reduce_launch(const float* slab_w, const float* slab_b, int ksplit, int ntaps, int Cin, int Cout, int64_t stap,
                         int64_t sk, int64_t sn, float* dw, float* db, hipStream_t s);

template <int CIN_T, int NKW>
static int launch_ws(const WTileArgs& a, hipStream_t s) {
  auto kern = conv_wgrad_ws_kernel<CIN_T, NKW>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      set_error("conv_wgrad_ws: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  const size_t lds = (size_t)2 * a.buf_floats * sizeof(float);
  hipLaunchKernelGGL(kern, dim3(a.ncot * a.ksplit), dim3(512), lds, s, a);
  LVAE_LAUNCH_CHECK("conv_wgrad_ws");
  return 0;
}

// returns -1000 when not eligible
int conv_wgrad_tile_try(const lvae_conv_desc* d, const float* dy, float* dw, float* db, void* workspace, hipStream_t s) {
  WTileArgs a;
  if (!wtile_plan(d, a) || !al16w(dy)) return -1000;
  a.d = *d;
  a.dy = dy;
  static const int dbg = getenv("LVAE_WG_DEBUG") ? atoi(getenv("LVAE_WG_DEBUG")) : 0;
  a.debug = dbg;
  const int ntaps = d->KH * d->KW;
  a.slab_w = static_cast<float*>(workspace);
  a.slab_b = db ? a.slab_w + (size_t)a.ksplit * ntaps * a.Cin * d->Cout : nullptr;
  const int cin_t = cin_tile(a.Cin);
  int rc;
  if (d->KH == 3) rc = cin_t == 32 ? launch_ws<32, 3>(a, s) : launch_ws<64, 3>(a, s);
  else rc = cin_t == 32 ? launch_ws<32, 1>(a, s) : (cin_t == 64 ? launch_ws<64, 1>(a, s) : launch_ws<128, 1>(a, s));
  if (rc) return rc;
  wgrad_reduce_launch(a.slab_w, a.slab_b, a.ksplit, ntaps, a.Cin, d->Cout, d->w_stap, d->w_sk, d->w_sn, dw, db, s);
  LVAE_LAUNCH_CHECK("conv2d_wgrad_reduce");
  return 0;
}

}  // namespace lvae
