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
  kernarg_warmup<(sizeof(WTileArgs) < 1024 ? sizeof(WTileArgs) : 1024)>();
  const int grp = blockIdx.x;
#include "conv3x3_wgrad_halo_body.inc"
}

// Several independent weight gradients in one launch (blockIdx.y = problem): the low-resolution levels fill 16-64 CUs per
// problem, and their launches are independent of everything but their own inputs.
constexpr int kMaxGroup = 12;
struct WTileGroup {
  WTileArgs p[kMaxGroup];
};
static_assert(sizeof(WTileGroup) <= 4096, "kernel argument block");

template <int CIN_T, int NKW>
__global__ __launch_bounds__(512, 2) void conv_wgrad_ws_grouped_kernel(WTileGroup g) {
  kernarg_warmup<(sizeof(WTileGroup) < 1024 ? sizeof(WTileGroup) : 1024)>();
  const WTileArgs& a = g.p[blockIdx.y];
  if ((int)blockIdx.x >= a.ncot * a.ksplit) return;  // uniform per workgroup, before any barrier
  const int grp = blockIdx.x;
#include "conv3x3_wgrad_halo_body.inc"
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
  static std::atomic<bool> attr_set{false};  // idempotent attribute write; the flag itself is race-free
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
  static const int dbg = lvae::debug_phase_switch("LVAE_WG_DEBUG");  // phase-skip builds (-DLVAE_PHASE_DEBUG) only; 0 in the product
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

struct ReduceArgs {
  const float* slab_w;
  const float* slab_b;
  int ksplit, ntaps, Cin, Cout;
  int64_t stap, sk, sn;
  float* dw;
  float* db;
};
void wgrad_reduce_grouped_launch(const ReduceArgs* r, int n, hipStream_t s);

template <int CIN_T, int NKW>
static int launch_ws_grouped(const WTileGroup& g, int n, int max_wgs, size_t lds, hipStream_t s) {
  auto kern = conv_wgrad_ws_grouped_kernel<CIN_T, NKW>;
  static std::atomic<bool> attr_set{false};  // idempotent attribute write; the flag itself is race-free
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      set_error("conv_wgrad_ws_grouped: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(max_wgs, n), dim3(512), lds, s, g);
  LVAE_LAUNCH_CHECK("conv_wgrad_ws_grouped");
  return 0;
}

// kind of the tile kernel a descriptor would use (-1: not eligible): 0 <32,3>, 1 <64,3>, 2 <32,1>, 3 <64,1>, 4 <128,1>
int conv_wgrad_tile_kind(const lvae_conv_desc* d) {
  WTileArgs a;
  if (!wtile_plan(d, a)) return -1;
  const int cin_t = cin_tile(a.Cin);
  if (d->KH == 3) return cin_t == 32 ? 0 : 1;
  return cin_t == 32 ? 2 : (cin_t == 64 ? 3 : 4);
}

// n <= kMaxGroup descriptors of the same kind, each with its own workspace: one launch + one grouped reduce
int conv_wgrad_tile_grouped(const lvae_conv_desc* const* ds, const float* const* dy, float* const* dw, float* const* db,
                            void* const* workspace, int n, int kind, hipStream_t s) {
  WTileGroup g;
  ReduceArgs r[kMaxGroup];
  static const int dbg = lvae::debug_phase_switch("LVAE_WG_DEBUG");  // phase-skip builds (-DLVAE_PHASE_DEBUG) only; 0 in the product
  int max_wgs = 0;
  size_t lds = 0;
  for (int i = 0; i < n; ++i) {
    WTileArgs& a = g.p[i];
    if (!wtile_plan(ds[i], a) || !al16w(dy[i])) return -1000;
    a.d = *ds[i];
    a.dy = dy[i];
    a.debug = dbg;
    const int ntaps = ds[i]->KH * ds[i]->KW;
    a.slab_w = static_cast<float*>(workspace[i]);
    a.slab_b = db[i] ? a.slab_w + (size_t)a.ksplit * ntaps * a.Cin * ds[i]->Cout : nullptr;
    if (a.ncot * a.ksplit > max_wgs) max_wgs = a.ncot * a.ksplit;
    const size_t l = (size_t)2 * a.buf_floats * sizeof(float);
    if (l > lds) lds = l;
    r[i] = ReduceArgs{a.slab_w, a.slab_b, a.ksplit, ntaps, a.Cin, ds[i]->Cout, ds[i]->w_stap, ds[i]->w_sk, ds[i]->w_sn, dw[i], db[i]};
  }
  for (int i = n; i < kMaxGroup; ++i) g.p[i] = g.p[0];
  int rc;
  switch (kind) {
    case 0: rc = launch_ws_grouped<32, 3>(g, n, max_wgs, lds, s); break;
    case 1: rc = launch_ws_grouped<64, 3>(g, n, max_wgs, lds, s); break;
    case 2: rc = launch_ws_grouped<32, 1>(g, n, max_wgs, lds, s); break;
    case 3: rc = launch_ws_grouped<64, 1>(g, n, max_wgs, lds, s); break;
    default: rc = launch_ws_grouped<128, 1>(g, n, max_wgs, lds, s);
  }
  if (rc) return rc;
  wgrad_reduce_grouped_launch(r, n, s);
  LVAE_LAUNCH_CHECK("conv2d_wgrad_reduce_grouped");
  return 0;
}

}  // namespace lvae
