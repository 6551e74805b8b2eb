// 3x3 / stride 1 / pad 1 convolution (forward and dgrad) with the input patch resident in LDS.
//
// The generic implicit-GEMM kernel (conv_igemm.hip) re-gathers the A tile from L2 for each of the 9 taps and
// recomputes the fused BatchNorm-apply + ELU 9 times per element; measured on MI355X it is L2-load bound at
// ~27 % of the fp32 MFMA peak. Here a workgroup stages the (TH+2) x (W+2) x Cin halo patch of its output tile
// ONCE (coalesced 16-B loads, input transform applied once per element, zero padding materialised), then runs the
// 9 taps x Cin reduction out of LDS: per tap only a 64 x Cin weight tile (16 KB, L2 resident) is streamed,
// double buffered against the MFMAs. A-fragments are ds_read_b128 at (pixel + tap offset) rows, padded to 68 floats.
//
// Tile: BM output pixels = NI images x TH rows x TW (= W) columns, BN = 64 output channels, 4 waves as 2(M) x 2(N).
#include <stdlib.h>

#include "lvae_common.h"

namespace lvae {

int conv_desc_check(const lvae_conv_desc* d, const char* who);

struct HaloArgs {
  lvae_conv_desc d;
  int TH, TW, NI, tiles_h, halo_w, halo_h, halo_px, ntn, flip, Cin, debug;
  uint32_t m_thw, m_tw, m_per_img, m_halo_w;  // fastdiv magics
};

template <int BM, int CIN_T, bool B_KCONTIG>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(HaloArgs a) {
  kernarg_warmup<(sizeof(HaloArgs) < 1024 ? sizeof(HaloArgs) : 1024)>();
  constexpr int LDA = CIN_T + 4;   // A / k-contiguous B row stride (floats)
  constexpr int LDN = 64;          // n-contiguous B row stride
  // BM = 32 (low-resolution layers): the 4 waves are 2 (co halves) x 2 (K groups); both K groups work on every stage
  // concurrently (each takes half of the stage's 32 channels) and are summed through LDS in the epilogue: the dependent
  // MFMA chain per wave halves (144 instead of 288) and the tile count doubles.
  constexpr bool KG = BM == 32;
  constexpr int WMT = KG ? 32 : BM / 2, MI = WMT / 32;
  constexpr int CIN4 = CIN_T / 4;
  // reduction channels per stage: half a tap at Cin = 64 keeps LDS <= 80 KB -> 2 WGs/CU on the large tiles; the 32-pixel tile
  // (low-resolution levels, latency bound, small halo) takes whole taps: 9 barrier-separated stages instead of 18
  constexpr int KS = (BM == 32 && CIN_T >= 64) ? 64 : 32;
  constexpr int NP = KS / 16;      // float4 of a weight stage per thread
  constexpr int KS4 = KS / 4;
  constexpr int LDB = KS + 4;      // k-contiguous B row stride
  constexpr int SPT = CIN_T / KS;  // stages per tap
  constexpr int BBUF = 64 * LDB;   // floats per B buffer (covers both layouts: 64 x 36 >= 32 x 64)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + (size_t)a.halo_px * LDA;

  const lvae_conv_desc& d = a.d;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = KG ? 0 : wave >> 1, kg = KG ? wave >> 1 : 0, wn = wave & 1, li = lane & 31, lh = lane >> 5;

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_n = bid % a.ntn;
  const int tm = bid / a.ntn;
  const int th_idx = tm % a.tiles_h, ig = tm / a.tiles_h;
  const int n0 = ig * a.NI, oh0 = th_idx * a.TH, co0 = tile_n * 64;
  const int Cin = a.Cin;

  // ---- weight tile of one stage (tap, 32-channel half): global -> register ring (RING stages in flight) -> LDS.
  // One stage is only 16-32 MFMAs per wave (0.4-0.9 us), shorter than an L2 round trip, so loads run RING stages ahead.
  constexpr int RING = 3;
  f32x4 breg[RING][NP];
  auto load_b = [&](int stage, f32x4 (&r)[NP]) {
    const int tap = stage / SPT, k0 = (stage - tap * SPT) * KS;
    const float* wt = d.w + (int64_t)tap * d.w_stap;
    if (B_KCONTIG) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int n = t / KS4 + (256 / KS4) * p, k = k0 + (t % KS4) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (co0 + n < d.Cout && k < Cin) v = *reinterpret_cast<const f32x4*>(wt + (int64_t)(co0 + n) * d.w_sn + k);
        r[p] = v;
      }
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int k = k0 + (t >> 4) + 16 * p, n = (t & 15) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < Cin && co0 + n < d.Cout) v = *reinterpret_cast<const f32x4*>(wt + (int64_t)k * d.w_sk + co0 + n);
        r[p] = v;
      }
    }
  };
  auto store_b = [&](int buf, const f32x4 (&r)[NP]) {
    float* Bb = Bs + buf * BBUF;
    if (B_KCONTIG) {
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<f32x4*>(Bb + (t / KS4 + (256 / KS4) * p) * LDB + (t % KS4) * 4) = r[p];
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<f32x4*>(Bb + ((t >> 4) + 16 * p) * LDN + (t & 15) * 4) = r[p];
    }
  };

  constexpr int NST = 9 * SPT;
#pragma unroll
  for (int q = 0; q < RING; ++q) load_b(q, breg[q]);

  // ---- halo patch: every (pixel, 4 channels) once, transform fused, zeros outside the image / batch.
  // idx = t + 256u: the channel group c4 of a thread is the same for every element (256 % CIN4 == 0), so scale/shift
  // are loaded once; loads use clamped addresses + select (no exec-mask regions).
  if (!(a.debug & 1)) {
    const int per_img = a.halo_h * a.halo_w;
    const int total = a.halo_px * CIN4;
    const int c4 = (t % CIN4) * 4;
    const bool c_ok = c4 < Cin;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (d.in_scale && c_ok) {
      sc = *reinterpret_cast<const f32x4*>(d.in_scale + c4);
      sh = *reinterpret_cast<const f32x4*>(d.in_shift + c4);
    }
    const float* xc = d.x + c4;
    const int px0 = t / CIN4;
    for (int base = 0; base < total; base += 256 * 8) {
      f32x4 v[8];
      unsigned okm = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + t + 256 * u;
        const int px = base / CIN4 + px0 + u * (256 / CIN4);
        const int img = fastdiv(px, a.m_per_img), r = px - img * per_img;
        const int hy = fastdiv(r, a.m_halo_w), hx = r - hy * a.halo_w;
        const int n = n0 + img, ih = oh0 + hy - 1, iw = hx - 1;
        const bool ok = (idx < total) & (n < d.N) & ((unsigned)ih < (unsigned)d.H) & ((unsigned)iw < (unsigned)d.W) & c_ok;
        const unsigned off = ok ? (unsigned)(((n * d.H + ih) * d.W + iw) * Cin) : 0u;  // < 2^31 floats (checked on the host)
        v[u] = *reinterpret_cast<const f32x4*>(xc + off);
        okm |= ok ? (1u << u) : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + t + 256 * u;
        if (idx < total) {
          f32x4 w = {0.f, 0.f, 0.f, 0.f};
          if ((okm >> u) & 1u) {
            w = v[u];
            if (d.in_scale) w = act_fwd4(w * sc + sh, d.in_act);
          }
          const int px = base / CIN4 + px0 + u * (256 / CIN4);
          *reinterpret_cast<f32x4*>(As + px * LDA + c4) = w;
        }
      }
    }
  }
  store_b(0, breg[0]);

  // ---- per-lane halo row of its A-fragment pixels
  const int tile_px = a.NI * a.TH * a.TW;
  int hbase[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    int p = wm * WMT + mi * 32 + li;
    if (p >= tile_px) p = 0;  // masked rows read a valid address; their results are never stored
    const int img = fastdiv(p, a.m_thw), r = p - img * (a.TH * a.TW);
    const int ty = fastdiv(r, a.m_tw), tx = r - ty * a.TW;
    hbase[mi] = ((img * a.halo_h + ty) * a.halo_w + tx) * LDA + 4 * lh;
  }

  f32x16 acc[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;

  __syncthreads();

#pragma unroll
  for (int st = 0; st < NST; ++st) {
    if (a.debug & 4) break;
    const int buf = st & 1;
    if (st + RING < NST) load_b(st + RING, breg[st % RING]);  // slot st % RING was stored to LDS one stage ago
    const int tap = st / SPT, k0 = (st - tap * SPT) * KS;
    const int kh = tap / 3, kw = tap - kh * 3;
    const int dh = a.flip ? 2 - kh : kh, dw = a.flip ? 2 - kw : kw;
    const int tapoff = (dh * a.halo_w + dw) * LDA + k0;
    const float* Bb = Bs + buf * BBUF;
#pragma unroll
    for (int q8 = 0; q8 < (KG ? KS / 16 : KS / 8); ++q8) {
      const int kk = (KG ? kg * (KS / 2) : 0) + q8 * 8;
      f32x4 af[MI], bf;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const f32x4*>(As + hbase[mi] + tapoff + kk);
      if (B_KCONTIG) {
        bf = *reinterpret_cast<const f32x4*>(Bb + (wn * 32 + li) * LDB + kk + 4 * lh);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = Bb[(kk + 4 * lh + j) * LDN + wn * 32 + li];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[j], acc[mi], 0, 0, 0);
    }
    if (st + 1 < NST) store_b(buf ^ 1, breg[(st + 1) % RING]);
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS tile [BM][64+4] (the halo image is dead after the last barrier) -> each
  // thread stores 16 contiguous bytes of an NHWC row: whole 256-B rows per 16 lanes instead of 4-byte scatters.
  constexpr int LDO = 68;
  float* Os = smem;
  if (!(a.debug & 2)) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Os[kg * BM * LDO + (wm * WMT + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDO + wn * 32 + li] = acc[mi][r];
    __syncthreads();
    // tile pixel p of a tile that covers whole rows / whole images is pixel (n0*H + oh0)*W + p of the tensor: the output
    // pointer is linear in p (16 pixel rows per pass), only the image index needs a division (for N bound and dropout)
    const int c4 = (t & 15) * 4, col = co0 + c4;
    f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = st1, piv = st1;  // BatchNorm partials of the stored values (d.stats_out)
    if (d.stats_out && col < d.Cout) piv = *reinterpret_cast<const f32x4*>(d.stats_pivot + col);
    f32x4 bsh = piv, bmu = piv, brs = piv;  // LVAE_STATS_BN_BWD: piv = scale, then shift, mean, rstd of the [4][Cout] block
    if (d.stats_out && d.stats_mode == LVAE_STATS_BN_BWD && col < d.Cout) {
      bsh = *reinterpret_cast<const f32x4*>(d.stats_pivot + d.Cout + col);
      bmu = *reinterpret_cast<const f32x4*>(d.stats_pivot + 2 * d.Cout + col);
      brs = *reinterpret_cast<const f32x4*>(d.stats_pivot + 3 * d.Cout + col);
    }
    if (col < d.Cout) {
      f32x4 bias = {0.f, 0.f, 0.f, 0.f};
      if (d.bias) bias = *reinterpret_cast<const f32x4*>(d.bias + col);  // Cout % 4 == 0 on this path
      const int p0 = t >> 4;
      float* yp = d.y + ((size_t)(n0 * d.H + oh0) * d.W + p0) * d.Cout + col;
      const float* op = Os + p0 * LDO + c4;
      const int nvalid = min(tile_px, (d.N - n0) * a.TH * a.TW);  // pixels of images that exist
#pragma unroll
      for (int q = 0; q < BM / 16; ++q) {
        const int p = p0 + 16 * q;
        if (p < nvalid) {
          f32x4 v = *reinterpret_cast<const f32x4*>(op + q * 16 * LDO) + bias;
          if (KG) v += *reinterpret_cast<const f32x4*>(op + BM * LDO + q * 16 * LDO);  // second K group
          if (d.out_scale) {
            const int n = n0 + fastdiv(p, a.m_thw);
            v = v * *reinterpret_cast<const f32x4*>(d.out_scale + (size_t)n * d.Cout + col);
          }
          v = act_fwd4(v, d.out_act);
          store_wt4(yp + (size_t)q * 16 * d.Cout, v);
          if (d.stats_mode == LVAE_STATS_BN_BWD) {
            if (d.stats_out) {
              const f32x4 xv = *reinterpret_cast<const f32x4*>(d.stats_x + (size_t)((n0 * d.H + oh0) * d.W + p) * d.Cout + col);
              const f32x4 ag = act_grad4(xv * piv + bsh, d.stats_act);
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float gj = v[j] * ag[j];
                st1[j] += gj;
                st2[j] += gj * (xv[j] - bmu[j]) * brs[j];
              }
            }
          } else {
            const f32x4 dl = v - piv;
            st1 += dl;
            st2 += dl * dl;
          }
        }
      }
    }
    if (d.stats_out) {  // 16 pixel groups x 64 channels -> one row of partials per pixel tile (fixed order)
      __syncthreads();  // the staging tile is dead
      float* red = smem;
      *reinterpret_cast<f32x4*>(red + (t >> 4) * 64 + c4) = st1;
      *reinterpret_cast<f32x4*>(red + 1024 + (t >> 4) * 64 + c4) = st2;
      __syncthreads();
      if (t < 128) {
        const int c = t & 63, which = t >> 6;
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) v += red[which * 1024 + r * 64 + c];
        if (co0 + c < d.Cout) d.stats_out[((size_t)tm * 2 + which) * d.Cout + co0 + c] = v;
      }
    }
  }
}

constexpr int kHaloNotEligible = -1000;

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// tile geometry for a W-wide image: rows per tile and images per tile so that NI*TH*W <= BM
static bool halo_plan(int N, int H, int W, int BM, int cin_t, HaloArgs& a) {
  if (W > BM) return false;
  int TH = 1;
  for (int c = 1; c <= H; ++c)
    if (H % c == 0 && c * W <= BM) TH = c;
  int NI = BM / (TH * W);
  if (NI < 1) NI = 1;
  if (TH < H) NI = 1;  // partial images are not packed with others
  if (NI > N) NI = N;
  a.TH = TH;
  a.TW = W;
  a.NI = NI;
  a.tiles_h = H / TH;
  a.halo_h = TH + 2;
  a.halo_w = W + 2;
  a.halo_px = NI * a.halo_h * a.halo_w;
  a.m_thw = fastdiv_magic(TH * W);
  a.m_tw = fastdiv_magic(W);
  a.m_per_img = fastdiv_magic(a.halo_h * a.halo_w);
  a.m_halo_w = fastdiv_magic(a.halo_w);
  const size_t lds = ((size_t)a.halo_px * (cin_t + 4) + 2 * 64 * 36) * sizeof(float);
  return lds <= 160 * 1024;
}

template <int BM, int CIN_T, bool B_KCONTIG>
static int launch_halo(HaloArgs a, hipStream_t s) {
  auto kern = conv3x3_halo_kernel<BM, CIN_T, B_KCONTIG>;
  static std::atomic<bool> attr_set{false};  // idempotent attribute write; the flag itself is race-free
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) {
      set_error("conv3x3_halo: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  constexpr int ks = (BM == 32 && CIN_T >= 64) ? 64 : 32;
  size_t lds = ((size_t)a.halo_px * (CIN_T + 4) + 2 * 64 * (ks + 4)) * sizeof(float);
  const size_t lds_out = (size_t)(BM == 32 ? 2 : 1) * BM * 68 * sizeof(float);  // epilogue staging tile(s)
  if (lds < lds_out) lds = lds_out;
  const int img_groups = (a.d.N + a.NI - 1) / a.NI;
  a.ntn = (a.d.Cout + 63) / 64;
  hipLaunchKernelGGL(kern, dim3(img_groups * a.tiles_h * a.ntn), dim3(256), lds, s, a);
  LVAE_LAUNCH_CHECK("conv3x3_halo");
  return 0;
}

// eligibility + tile plan: returns the tile size (128 / 64 / 32), 0 when the descriptor does not fit this kernel
static int halo_select(const lvae_conv_desc* d, HaloArgs& a, bool& ncontig_out) {
  const int Cin = d->C1;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->x2 != nullptr || d->OH != d->H || d->OW != d->W) return 0;
  if (Cin > 64 || Cin % 4 != 0 || !al16(d->x) || !al16(d->w) || d->w_stap % 4 != 0) return 0;
  if ((int64_t)d->N * d->H * d->W * Cin >= ((int64_t)1 << 31)) return 0;  // 32-bit element offsets in the halo loader
  if (d->in_scale && (!al16(d->in_scale) || !al16(d->in_shift))) return 0;
  const bool kcontig = d->w_sk == 1 && d->w_sn % 4 == 0;
  const bool ncontig = d->w_sn == 1 && d->w_sk % 4 == 0 && d->Cout % 4 == 0;
  if (!kcontig && !ncontig) return 0;
  if (d->Cout % 4 != 0 || !al16(d->y) || (d->bias && !al16(d->bias)) || (d->out_scale && !al16(d->out_scale))) return 0;
  if (d->stats_out && (!al16(d->stats_pivot) || d->stats_pivot == nullptr)) return 0;
  const int cin_t = Cin <= 32 ? 32 : 64;
  const int64_t M = (int64_t)d->N * d->H * d->W;
  a.d = *d;
  a.Cin = Cin;
  a.flip = d->gather == LVAE_GATHER_TRANSPOSED ? 1 : 0;
  int BM = M >= 128 * 192 ? 128 : (M > 64 * 256 ? 64 : 32);
  while (!halo_plan(d->N, d->H, d->W, BM, cin_t, a)) {
    if (BM == 32) return 0;
    BM /= 2;
  }
  ncontig_out = ncontig;
  return BM;
}

// rows of BatchNorm partials a launch writes (one per pixel tile), 0 when this kernel would not run
int conv3x3_halo_stats_rows(const lvae_conv_desc* d) {
  HaloArgs a;
  bool nc;
  if (!halo_select(d, a, nc)) return 0;
  return ((d->N + a.NI - 1) / a.NI) * a.tiles_h;
}

// returns kHaloNotEligible when the descriptor does not fit this kernel (the caller then uses the generic one)
int conv3x3_halo_try(const lvae_conv_desc* d, hipStream_t s) {
  HaloArgs a;
  bool ncontig = false;
  const int BM = halo_select(d, a, ncontig);
  if (BM == 0) return kHaloNotEligible;
  const int cin_t = d->C1 <= 32 ? 32 : 64;
  static const int dbg = lvae::debug_phase_switch("LVAE_HALO_DEBUG");  // phase-skip builds (-DLVAE_PHASE_DEBUG) only; 0 in the product
  a.debug = dbg;
  // n-contiguous weights take precedence when both hold (Cin == 1 cannot reach here)
  const bool kc = !ncontig;
  if (BM == 128) {
    if (cin_t == 64) return kc ? launch_halo<128, 64, true>(a, s) : launch_halo<128, 64, false>(a, s);
    return kc ? launch_halo<128, 32, true>(a, s) : launch_halo<128, 32, false>(a, s);
  }
  if (BM == 64) {
    if (cin_t == 64) return kc ? launch_halo<64, 64, true>(a, s) : launch_halo<64, 64, false>(a, s);
    return kc ? launch_halo<64, 32, true>(a, s) : launch_halo<64, 32, false>(a, s);
  }
  if (cin_t == 64) return kc ? launch_halo<32, 64, true>(a, s) : launch_halo<32, 64, false>(a, s);
  return kc ? launch_halo<32, 32, true>(a, s) : launch_halo<32, 32, false>(a, s);
}

}  // namespace lvae
