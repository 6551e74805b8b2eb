// 3x3 / stride 1 / pad 1 convolution (forward and dgrad) on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
// gfx950 issues a bf16 MFMA 16x faster than the fp32 one (1024 vs 64 FLOP/clk/SIMD), and the fp32 MFMA is what bounds the
// large 3x3 layers of the fp32 model. Two precisions share this kernel:
//
//   SPLIT = 1  bf16 operands: activations and weights are rounded to bf16 at the MFMA input, products are exact, sums fp32
//              (the arithmetic of the reference under torch.autocast(bfloat16); BASELINE configs[1], [3], [4]).
//   SPLIT = 3  fp32-equivalent: every operand is split exactly into three bf16 pieces a = a1 + a2 + a3 (8 + 8 + 8 significant
//              bits; each remainder is exact in fp32) and the product is formed from the six piece pairs of order <= 2^-16:
//              a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1 — what is dropped is below 2^-24 of the product, i.e. below the
//              rounding an fp32 multiply-add makes anyway. Six bf16 MFMAs cost 6/16 of one fp32 MFMA of the same shape:
//              2.7x the fp32-MFMA rate, and (unlike the fp32 MFMA) they leave the vector ALU free for the splits.
//
// Structure (as conv3x3_halo.hip): a workgroup stages the halo patch of its 128-pixel output tile once — input transform
// (BatchNorm-apply + activation) applied once per element, then split — as SPLIT bf16 planes in LDS (A fragments: 16-byte
// ds_read_b128 of 8 consecutive reduction channels, rows padded to 144 bytes: conflict free). The weights never touch LDS: they
// are pre-split once per optimizer step into planes stored in MFMA FRAGMENT ORDER (bf_weight_kernel), so a wave's B fragment is
// one coalesced 1 KB load from L2, prefetched two k-steps ahead in registers. With no weights in LDS there is no barrier in
// the reduction loop and the patch alone (<= 80 KB) lets two workgroups share a CU: one's loads overlap the other's MFMAs.
// Epilogue through LDS with bias, Dropout2d scale, activation and the BatchNorm statistics / BatchNorm-backward sums of
// conv3x3_halo.hip.
#include <string.h>

#include "bf16_frag.h"
#include "lvae_common.h"

namespace lvae {

struct BfArgs {
  lvae_conv_desc d;
  const __bf16* Wp;  // pre-split weights in fragment order [tap][Cout tile][k-step 4][plane][n half 2][lane 64][8] (bf_weight_kernel)
  int TH, TW, NI, tiles_h, halo_w, halo_h, halo_px, ntn, flip, Cin;
  uint32_t m_thw, m_tw, m_per_img, m_halo_w, m_tiles_h;
  int bm;     // output pixels per workgroup (64 | 128)
  int debug;  // phase-skip builds only (-DLVAE_PHASE_DEBUG): 1 = no halo staging, 2 = no epilogue, 4 = no MFMAs, 8 = no weight staging
};

constexpr int kBfNotEligible = -1000;
constexpr int BF_LDK = 72;   // bf16 elements per LDS row (64 channels + 8 pad = 144 bytes)

// MI: 32-pixel MFMA row blocks per wave; the workgroup tile is BM = 64 * MI output pixels x 64 output channels, 4 waves 2(M) x 2(N)
// PRE (SPLIT == 1 only): the wave's whole B slice is fetched up front (144 registers, two workgroups per CU) — for layers whose grid is a
// single round of workgroups, where the per-tile latency chain is the launch time; larger grids keep the three-deep ring (87 registers,
// five workgroups per CU hide the fragment latency by occupancy instead).
// XB / YB (SPLIT == 1, bf16 storage of the tensors inside a residual block): x / y are bfloat16 and move as 16-byte pieces - a thread
// stages 8 channels of a pixel (one 16-byte load, one ds_write_b128) and the epilogue stores 8 channels per lane (one 16-byte
// write-through store) - so a bf16 tensor costs half the memory instructions of an fp32 one, not the same number at half the width
// (8-byte accesses measured SLOWER than fp32 storage: 19.6 vs 17.9 us at 256x16x16). Mixed cases the training step does not produce
// (bf16 in / fp32 out) take the generic path with the descriptor's run-time dtype flags.
template <int SPLIT, int MI, bool PRE = false, bool XB = false, bool YB = false>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16_kernel(BfArgs a) {
  kernarg_warmup<(sizeof(BfArgs) < 1024 ? sizeof(BfArgs) : 1024)>();
  static_assert(SPLIT == 1 || (!XB && !YB), "bf16 storage exists for the bf16-operand form only");
  constexpr int BM = 64 * MI, LDK = BF_LDK;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* As = reinterpret_cast<__bf16*>(smem_raw);                    // [SPLIT][halo_px][LDK]
  const int a_plane = a.halo_px * LDK;

  const lvae_conv_desc& d = a.d;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;

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

  // ---- weights: this wave's B fragments of k-step s = tap * 4 + ks, one 16-byte load per plane (1 KB per wave, contiguous)
  bf16x8 bq[3][SPLIT];
  const __bf16* wp_lane = a.Wp + (size_t)tile_n * 4 * SPLIT * 1024 + (size_t)wn * 512 + (size_t)lane * 8;
  auto load_b = [&](int step, bf16x8 (&r)[SPLIT]) {
    const int tap = step >> 2, ks = step & 3;
    const __bf16* src = wp_lane + ((size_t)(tap * a.ntn) * 4 + ks) * SPLIT * 1024;
#pragma unroll
    for (int p = 0; p < SPLIT; ++p) r[p] = *reinterpret_cast<const bf16x8*>(src + p * 1024);
  };
  // SPLIT == 1: a k-step is only MI MFMAs (64-128 cycles), far shorter than an L2 round trip, so a two-step ring stalls on every
  // step (measured: 20 us per tile, ~10 of them waiting for fragments). The wave's whole B slice — 36 fragments, 144 registers —
  // is fetched up front instead, while the patch is being staged.
  bf16x8 ball[PRE ? 36 : 1];
  if (PRE) {
#pragma unroll
    for (int st = 0; st < 36; ++st) {
      bf16x8 tmp[SPLIT];
      load_b(st, tmp);
      ball[PRE ? st : 0] = tmp[0];
    }
  } else if (!(a.debug & 8)) {
    load_b(0, bq[0]);
    load_b(1, bq[1]);
  }

  // ---- halo patch, bf16-stored x: every (pixel, 8 channels) once: one 16-byte load, transform in fp32, one 16-byte LDS store
  if (XB) {
    const int per_img = a.halo_h * a.halo_w;
    const int total = a.halo_px * 8;
    const int c8 = (t & 7) * 8;
    const bool c_ok = c8 < Cin;   // Cin % 8 == 0 (launcher)
    f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sh0 = {0.f, 0.f, 0.f, 0.f}, sh1 = sh0;
    if (d.in_scale && c_ok) {
      sc0 = *reinterpret_cast<const f32x4*>(d.in_scale + c8);
      sc1 = *reinterpret_cast<const f32x4*>(d.in_scale + c8 + 4);
      sh0 = *reinterpret_cast<const f32x4*>(d.in_shift + c8);
      sh1 = *reinterpret_cast<const f32x4*>(d.in_shift + c8 + 4);
    }
    const __bf16* xb = reinterpret_cast<const __bf16*>(d.x);
    const int px0 = t >> 3;
    for (int base = 0; base < total; base += 256 * 8) {
      bf16x8 v[8];
      unsigned okm = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + t + 256 * u;
        const int px = (base >> 3) + px0 + u * 32;
        const int img = fastdiv(px, a.m_per_img), r = px - img * per_img;
        const int hy = fastdiv(r, a.m_halo_w), hx = r - hy * a.halo_w;
        const int n = n0 + img, ih = oh0 + hy - 1, iw = hx - 1;
        const bool ok = (idx < total) & (n < d.N) & ((unsigned)ih < (unsigned)d.H) & ((unsigned)iw < (unsigned)d.W) & c_ok;
        const unsigned off = ok ? (unsigned)(((n * d.H + ih) * d.W + iw) * Cin + c8) : 0u;
        v[u] = *reinterpret_cast<const bf16x8*>(xb + off);
        okm |= ok ? (1u << u) : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + t + 256 * u;
        if (idx < total) {
          bf16x8 w = v[u];
          if (d.in_scale) {
            f32x4 lo = {(float)w[0], (float)w[1], (float)w[2], (float)w[3]}, hi = {(float)w[4], (float)w[5], (float)w[6], (float)w[7]};
            lo = act_fwd4(lo * sc0 + sh0, d.in_act);
            hi = act_fwd4(hi * sc1 + sh1, d.in_act);
            w = to_bf16x8(lo, hi);
          }
          if (!((okm >> u) & 1u)) w = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
          const int px = (base >> 3) + px0 + u * 32;
          *reinterpret_cast<bf16x8*>(As + px * LDK + c8) = w;
        }
      }
    }
  }
  // ---- halo patch: every (pixel, 4 channels) once; transform fused, split into planes, zeros outside the image / batch
  if (!XB && !(a.debug & 1)) {
    const int per_img = a.halo_h * a.halo_w;
    const int total = a.halo_px * 16;
    const int c4 = (t & 15) * 4;
    const bool c_ok = c4 < Cin;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (d.in_scale && c_ok) {
      sc = *reinterpret_cast<const f32x4*>(d.in_scale + c4);
      sh = *reinterpret_cast<const f32x4*>(d.in_shift + c4);
    }
    const float* xc = d.x;
    const bool xbf = d.x_dtype == LVAE_DT_BF16;
    const int px0 = t >> 4;
    for (int base = 0; base < total; base += 256 * 8) {
      f32x4 v[8];
      unsigned okm = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + t + 256 * u;
        const int px = (base >> 4) + px0 + u * 16;
        const int img = fastdiv(px, a.m_per_img), r = px - img * per_img;
        const int hy = fastdiv(r, a.m_halo_w), hx = r - hy * a.halo_w;
        const int n = n0 + img, ih = oh0 + hy - 1, iw = hx - 1;
        const bool ok = (idx < total) & (n < d.N) & ((unsigned)ih < (unsigned)d.H) & ((unsigned)iw < (unsigned)d.W) & c_ok;
        const unsigned off = ok ? (unsigned)(((n * d.H + ih) * d.W + iw) * Cin + c4) : 0u;
        v[u] = load4_dt(xc, off, xbf);
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
          const int px = (base >> 4) + px0 + u * 16;
          bf16x4 pl[SPLIT];
          split4<SPLIT>(w, pl);
#pragma unroll
          for (int p = 0; p < SPLIT; ++p) *reinterpret_cast<bf16x4*>(As + p * a_plane + px * LDK + c4) = pl[p];
        }
      }
    }
  }
  // ---- per-lane halo row of its A-fragment pixels (element offset inside a plane)
  const int tile_px = a.NI * a.TH * a.TW;
  int hbase[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    int p = wm * (32 * MI) + mi * 32 + li;
    if (p >= tile_px) p = 0;
    const int img = fastdiv(p, a.m_thw), r = p - img * (a.TH * a.TW);
    const int ty = fastdiv(r, a.m_tw), tx = r - ty * a.TW;
    hbase[mi] = ((img * a.halo_h + ty) * a.halo_w + tx) * LDK + 8 * lh;
  }

  f32x16 acc[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;

  __syncthreads();

  // 36 k-steps (9 taps x 4 blocks of 16 channels), no barrier: B two steps ahead from L2, A one step ahead from LDS
  bf16x8 af[2][MI][SPLIT];
  auto load_a = [&](int step, bf16x8 (&fa)[MI][SPLIT]) {
    const int tap = step >> 2, ks = step & 3;
    const int kh = tap / 3, kw = tap - kh * 3;
    const int dh = a.flip ? 2 - kh : kh, dw = a.flip ? 2 - kw : kw;
    const int off = (dh * a.halo_w + dw) * LDK + ks * 16;
#pragma unroll
    for (int p = 0; p < SPLIT; ++p)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) fa[mi][p] = *reinterpret_cast<const bf16x8*>(As + p * a_plane + hbase[mi] + off);
  };
  if (!(a.debug & 4)) load_a(0, af[0]);
#pragma unroll
  for (int step = 0; step < 36; ++step) {
    if (a.debug & 4) break;
    const int cur = step & 1;
    if (!PRE && step + 2 < 36 && !(a.debug & 8)) load_b(step + 2, bq[(step + 2) % 3]);
    if (step + 1 < 36) load_a(step + 1, af[cur ^ 1]);
    if (PRE) bq[step % 3][0] = ball[PRE ? step : 0];
    const bf16x8 (&bf)[SPLIT] = bq[step % 3];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      if (SPLIT == 1) {
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][mi][0], bf[0], acc[mi], 0, 0, 0);
      } else {
        // smallest terms first
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][mi][SPLIT - 1], bf[0], acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][mi][0], bf[SPLIT - 1], acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][mi][1], bf[1], acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][mi][1], bf[0], acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][mi][0], bf[1], acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][mi][0], bf[0], acc[mi], 0, 0, 0);
      }
    }
  }
  __syncthreads();  // every wave is done with the patch: LDS becomes the output staging tile

  // ---- epilogue (as conv3x3_halo.hip): accumulators -> LDS tile [BM][68] floats -> 16-byte row stores
  if (a.debug & 2) return;
  constexpr int LDO = 68;
  float* Os = reinterpret_cast<float*>(smem_raw);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      Os[(wm * (32 * MI) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDO + wn * 32 + li] = acc[mi][r];
  __syncthreads();
  if (YB) {
    // ---- bf16-stored y: 8 channels per lane, one 16-byte write-through store per pixel; statistics from the fp32 values
    const int c8 = (t & 7) * 8, col = co0 + c8;   // Cout % 8 == 0 (launcher)
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 st1[2] = {z4, z4}, st2[2] = {z4, z4}, piv[2] = {z4, z4}, bsh[2] = {z4, z4}, bmu[2] = {z4, z4}, brs[2] = {z4, z4}, bias[2] = {z4, z4};
    const bool bwd = d.stats_mode == LVAE_STATS_BN_BWD;
    if (col < d.Cout) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (d.stats_out) piv[h] = *reinterpret_cast<const f32x4*>(d.stats_pivot + col + 4 * h);
        if (d.stats_out && bwd) {
          bsh[h] = *reinterpret_cast<const f32x4*>(d.stats_pivot + d.Cout + col + 4 * h);
          bmu[h] = *reinterpret_cast<const f32x4*>(d.stats_pivot + 2 * d.Cout + col + 4 * h);
          brs[h] = *reinterpret_cast<const f32x4*>(d.stats_pivot + 3 * d.Cout + col + 4 * h);
        }
        if (d.bias) bias[h] = *reinterpret_cast<const f32x4*>(d.bias + col + 4 * h);
      }
      const int p0 = t >> 3;
      __bf16* yb16 = reinterpret_cast<__bf16*>(d.y);
      const size_t ybase = ((size_t)(n0 * d.H + oh0) * d.W) * d.Cout + col;
      const bool sxbf = d.stats_x_dtype == LVAE_DT_BF16;
      const int nvalid = min(tile_px, (d.N - n0) * a.TH * a.TW);
      // the BatchNorm input rows of the backward sums are requested BEFORE the first store of this pass: a wave's memory counter retires in
      // order, so a load issued behind a write-through store waits for that store's trip to memory (conv3x3_wino.hip, round 4)
      // (all rows requested in one go — a dtype branch per row made each row its own round trip — and the Dropout2d mask rows with them:
      // loaded between the stores they were three more dependent round trips, `load; s_waitcnt vmcnt(0)` each)
      f32x4 sx0[BM / 32], sx1[BM / 32], mk0[BM / 32], mk1[BM / 32];
      bf16x8 sxq[BM / 32];
      const bool want_sx = bwd && d.stats_out;
      if (want_sx && sxbf) {
#pragma unroll
        for (int q = 0; q < BM / 32; ++q) {
          const int p = p0 + 32 * q, pc = p < nvalid ? p : 0;
          sxq[q] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(d.stats_x) + (size_t)((n0 * d.H + oh0) * d.W + pc) * d.Cout + col);
        }
      } else if (want_sx) {
#pragma unroll
        for (int q = 0; q < BM / 32; ++q) {
          const int p = p0 + 32 * q, pc = p < nvalid ? p : 0;
          const size_t xo = (size_t)((n0 * d.H + oh0) * d.W + pc) * d.Cout + col;
          sx0[q] = *reinterpret_cast<const f32x4*>(d.stats_x + xo);
          sx1[q] = *reinterpret_cast<const f32x4*>(d.stats_x + xo + 4);
        }
      }
#pragma unroll
      for (int q = 0; q < BM / 32; ++q) {
        const int p = p0 + 32 * q, pc = p < nvalid ? p : 0;
        mk0[q] = mk1[q] = f32x4{1.f, 1.f, 1.f, 1.f};
        if (d.out_scale) {
          const int n = n0 + fastdiv(pc, a.m_thw);
          mk0[q] = *reinterpret_cast<const f32x4*>(d.out_scale + (size_t)n * d.Cout + col);
          mk1[q] = *reinterpret_cast<const f32x4*>(d.out_scale + (size_t)n * d.Cout + col + 4);
        }
      }
      if (want_sx && sxbf) {
#pragma unroll
        for (int q = 0; q < BM / 32; ++q) {
          sx0[q] = f32x4{(float)sxq[q][0], (float)sxq[q][1], (float)sxq[q][2], (float)sxq[q][3]};
          sx1[q] = f32x4{(float)sxq[q][4], (float)sxq[q][5], (float)sxq[q][6], (float)sxq[q][7]};
        }
      }
#pragma unroll
      for (int q = 0; q < BM / 32; ++q) {
        const int p = p0 + 32 * q;
        if (p < nvalid) {
          f32x4 v[2];
          v[0] = (*reinterpret_cast<const f32x4*>(Os + p * LDO + c8) + bias[0]) * mk0[q];
          v[1] = (*reinterpret_cast<const f32x4*>(Os + p * LDO + c8 + 4) + bias[1]) * mk1[q];
          v[0] = act_fwd4(v[0], d.out_act);
          v[1] = act_fwd4(v[1], d.out_act);
          const bf16x8 o = to_bf16x8(v[0], v[1]);
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(yb16 + ybase + (size_t)p * d.Cout), "v"(o) : "memory");
          if (bwd) {
            if (d.stats_out) {
              const f32x4 xv[2] = {sx0[q], sx1[q]};
const f32x4 ag[2] = {act_grad4(xv[0] * piv[0] + bsh[0], d.stats_act), act_grad4(xv[1] * piv[1] + bsh[1], d.stats_act)};
#pragma unroll
              for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const float gj = v[h][j] * ag[h][j];
                  st1[h][j] += gj;
                  st2[h][j] += gj * (xv[h][j] - bmu[h][j]) * brs[h][j];
                }
            }
          } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const f32x4 dl = v[h] - piv[h];
              st1[h] += dl;
              st2[h] += dl * dl;
            }
          }
        }
      }
    }
    if (d.stats_out) {  // 32 pixel groups x 64 channels -> one row of partials per pixel tile (fixed order)
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem_raw);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        *reinterpret_cast<f32x4*>(red + (t >> 3) * 64 + c8 + 4 * h) = st1[h];
        *reinterpret_cast<f32x4*>(red + 2048 + (t >> 3) * 64 + c8 + 4 * h) = st2[h];
      }
      __syncthreads();
      if (t < 128) {
        const int c = t & 63, which = t >> 6;
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) v += red[which * 2048 + r * 64 + c];
        if (co0 + c < d.Cout) d.stats_out[((size_t)tm * 2 + which) * d.Cout + co0 + c] = v;
      }
    }
    return;
  }
  const int c4 = (t & 15) * 4, col = co0 + c4;
  f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = st1, piv = st1;
  if (d.stats_out && col < d.Cout) piv = *reinterpret_cast<const f32x4*>(d.stats_pivot + col);
  f32x4 bsh = piv, bmu = piv, brs = piv;
  if (d.stats_out && d.stats_mode == LVAE_STATS_BN_BWD && col < d.Cout) {
    bsh = *reinterpret_cast<const f32x4*>(d.stats_pivot + d.Cout + col);
    bmu = *reinterpret_cast<const f32x4*>(d.stats_pivot + 2 * d.Cout + col);
    brs = *reinterpret_cast<const f32x4*>(d.stats_pivot + 3 * d.Cout + col);
  }
  if (col < d.Cout) {
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (d.bias) bias = *reinterpret_cast<const f32x4*>(d.bias + col);
    const int p0 = t >> 4;
    const size_t ybase = ((size_t)(n0 * d.H + oh0) * d.W + p0) * d.Cout + col;  // element offset of this thread's first output
    const bool ybf = d.y_dtype == LVAE_DT_BF16, sxbf = d.stats_x_dtype == LVAE_DT_BF16;
    const float* op = Os + p0 * LDO + c4;
    const int nvalid = min(tile_px, (d.N - n0) * a.TH * a.TW);
    f32x4 sxr[BM / 16], mkr[BM / 16];   // BatchNorm input rows of the backward sums and the mask rows, requested before the first store of the pass (see above)
    if (d.stats_mode == LVAE_STATS_BN_BWD && d.stats_out) {
      if (sxbf) {
#pragma unroll
        for (int q = 0; q < BM / 16; ++q) {
          const int p = p0 + 16 * q, pc = p < nvalid ? p : 0;
          sxr[q] = load4_dt(d.stats_x, (size_t)((n0 * d.H + oh0) * d.W + pc) * d.Cout + col, true);
        }
      } else {
#pragma unroll
        for (int q = 0; q < BM / 16; ++q) {
          const int p = p0 + 16 * q, pc = p < nvalid ? p : 0;
          sxr[q] = load4_dt(d.stats_x, (size_t)((n0 * d.H + oh0) * d.W + pc) * d.Cout + col, false);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < BM / 16; ++q) {
      const int p = p0 + 16 * q, pc = p < nvalid ? p : 0;
      mkr[q] = f32x4{1.f, 1.f, 1.f, 1.f};
      if (d.out_scale) mkr[q] = *reinterpret_cast<const f32x4*>(d.out_scale + (size_t)(n0 + fastdiv(pc, a.m_thw)) * d.Cout + col);
    }
    // consumed before the first store too: rows only used under `p < nvalid` stay pending on the path that skips them, and hipcc then drains
    // the whole counter (the stores with it) when their registers are reused in a later row
#pragma unroll
    for (int q = 0; q < BM / 16; ++q) asm volatile("" ::"v"(mkr[q]));
    if (d.stats_mode == LVAE_STATS_BN_BWD && d.stats_out) {
#pragma unroll
      for (int q = 0; q < BM / 16; ++q) asm volatile("" ::"v"(sxr[q]));
    }
#pragma unroll
    for (int q = 0; q < BM / 16; ++q) {
      const int p = p0 + 16 * q;
      if (p < nvalid) {
        f32x4 v = (*reinterpret_cast<const f32x4*>(op + q * 16 * LDO) + bias) * mkr[q];
        v = act_fwd4(v, d.out_act);
        store4_dt(d.y, ybase + (size_t)q * 16 * d.Cout, v, ybf);
        if (d.stats_mode == LVAE_STATS_BN_BWD) {
          if (d.stats_out) {
            const f32x4 xv = sxr[q];
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
  if (d.stats_out) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_raw);
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

// (A persistent two-workgroups-per-CU form of this kernel, prefetching the next tile's patch during the MFMAs, was built in round 2 and
// measured slower on the CIFAR-15 step, 33.8 vs 33.1 ms: at 256x16x16 there is one tile per workgroup anyway. Removed.)

// ---------------------------------------------------------------------------------------------------------------------------
// Weight gradient with bf16 operands:  dW[tap][ci][co] += sum_pixels T(x)[pixel + tap][ci] * dy[pixel][co],  db[co] += sum dy
// (T = the fused BatchNorm-apply + activation of the forward). The reduction runs over PIXELS, so both MFMA operands need 8
// consecutive pixels of one channel per lane while the tensors are NHWC: the x patch and the dy tile are staged as [pixel][channel]
// bf16 images exactly like the forward kernel's, and the fragments are read with ds_read_b64_tr_b16 (a 4-pixel x 16-channel block
// per 16 lanes, delivered channel-major: the hardware transpose; semantics probed on the device with tools/tr_probe.hip). A tap
// shift only changes which pixel rows are addressed, so there is no alignment problem.
// Persistent: one 512-thread workgroup per CU loops over 128- or 64-pixel tiles; wave = (ci half, co half, tap group {0-4 | 5-8}) keeps
// its <= 5 accumulator tiles (32 ci x 32 co each) in registers across all tiles; the raw x / dy of the next tile are prefetched
// into registers during the MFMAs. Per-workgroup partial slabs [workgroup][tap][ci][co] + [workgroup][co], fixed-order reduce.
// ---------------------------------------------------------------------------------------------------------------------------
struct BfWgArgs {
  lvae_conv_desc d;
  const float* dy;
  float* slab_w;   // [nwg][9][Cin][Cout]
  float* slab_b;   // [nwg][Cout] or null
  int TH, TW, NI, tiles_h, halo_w, halo_h, halo_px, ntiles, Cin, bm, split;
  uint32_t m_thw, m_tw, m_per_img, m_halo_w, m_tiles_h;
};

// SPLIT = 1: bf16 operands. SPLIT = 3: both operands split exactly into three bf16 pieces and the six piece products of order <= 2^-16
// accumulated (conv3x3_bf16_kernel's fp32-equivalent form). Unlike the forward kernel this one is bound by its slab traffic, not by
// MFMAs (40 per wave and tile); the six-fold matrix work of SPLIT = 3 was hoped to hide under the memory time but does not (bfwg_form
// below has the measurement), so SPLIT = 3 is an opt-in form that the parity tests keep exercised.
// XB / DB: x / dy are bf16-stored (residual-block internals under compute_dtype bf16) and are staged as 16-byte pieces: a thread takes 8
// channels of a pixel (half the memory instructions of the fp32 form; see conv3x3_bf16_kernel). Without them the descriptor's run-time
// dtype flags are honoured with 4-channel accesses.
template <int MI, int SPLIT, bool XB = false, bool DB = false>  // tile = 64 * MI pixels
__global__ __launch_bounds__(512) void conv3x3_wgrad_bf16_kernel(BfWgArgs a) {
  kernarg_warmup<(sizeof(BfWgArgs) < 1024 ? sizeof(BfWgArgs) : 1024)>();
  static_assert(SPLIT == 1 || (!XB && !DB), "bf16 storage exists for the bf16-operand form only");
  constexpr int BM = 64 * MI, LDK = BF_LDK, KS = BM / 16;   // k-steps of 16 pixels per tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* Xs = reinterpret_cast<__bf16*>(smem_raw);          // [SPLIT][halo_px][LDK]
  const int x_plane = a.halo_px * LDK;
  __bf16* Ds = Xs + (size_t)SPLIT * x_plane;                  // [SPLIT][BM][LDK]
  constexpr int d_plane = BM * LDK;
  constexpr int UNR = SPLIT == 1 ? 2 : 1;  // k-steps in flight (register budget)
  const lvae_conv_desc& d = a.d;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int cih = wave & 1, coh = (wave >> 1) & 1, tg = wave >> 2;   // ci half, co half, tap group
  const int tap0 = tg * 5, ntap = tg == 0 ? 5 : 4;
  const int li = lane & 31, lh = lane >> 5, G = lane >> 4, i16 = lane & 15;
  const int Cin = a.Cin, co0 = blockIdx.y * 64;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- transposed-read row of this lane for (k-step s, half-read rd): tile pixel 16 s + 8 (G >> 1) + 4 rd + (i16 >> 2) -> halo pixel
  // index of tap (0, 0); recomputed per k-step (two magic divisions) rather than held in 16 registers
  auto xrow_of = [&](int s, int rd) {
    const int p = 16 * s + 8 * (G >> 1) + 4 * rd + (i16 >> 2);
    const int img = fastdiv(p, a.m_thw), r = p - img * (a.TH * a.TW);
    const int ty = fastdiv(r, a.m_tw), tx = r - ty * a.TW;
    return (img * a.halo_h + ty) * a.halo_w + tx;
  };
  const int chx = cih * 32 + 16 * (G & 1) + 4 * (i16 & 3);   // channel offset of this lane's 8-byte piece in an x row
  const int chd = coh * 32 + 16 * (G & 1) + 4 * (i16 & 3);   // ... in a dy row
  const int drow0 = 8 * (G >> 1) + (i16 >> 2);

  // ---- staging maps: thread -> (pixel, 4 channels); raw operands of the next tile live in registers during the MFMAs
  constexpr int XV = XB ? 1 : 7;        // float4 of the x patch per thread (halo_px <= 224 checked on the host)
  constexpr int DV = DB ? 1 : BM / 32;  // float4 of the dy tile per thread
  constexpr int XV8 = XB ? 4 : 1;       // 8-channel pieces of a bf16-stored x patch per thread (64 pixels per pass)
  constexpr int DV8 = DB ? BM / 64 : 1; // ... of a bf16-stored dy tile
  const int c4 = (t & 15) * 4, px0 = t >> 4;
  const int c8 = (t & 7) * 8, px8 = t >> 3;
  bf16x8 xr8[XV8], dr8[DV8];
  f32x4 sc1 = {1.f, 1.f, 1.f, 1.f}, sh1 = {0.f, 0.f, 0.f, 0.f}, sc0 = sc1, sh0 = sh1;   // 8-channel mapping: scale / shift of c8 .. c8 + 7
  if (XB && a.d.in_scale && c8 < a.Cin) {
    sc0 = *reinterpret_cast<const f32x4*>(a.d.in_scale + c8);
    sc1 = *reinterpret_cast<const f32x4*>(a.d.in_scale + c8 + 4);
    sh0 = *reinterpret_cast<const f32x4*>(a.d.in_shift + c8);
    sh1 = *reinterpret_cast<const f32x4*>(a.d.in_shift + c8 + 4);
  }
  const bool cx_ok = c4 < Cin, cd_ok = co0 + c4 < d.Cout;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = zero4;
  if (d.in_scale && cx_ok) {
    sc = *reinterpret_cast<const f32x4*>(d.in_scale + c4);
    sh = *reinterpret_cast<const f32x4*>(d.in_shift + c4);
  }
  const int per_img = a.halo_h * a.halo_w;
  f32x4 xr[XV], dr[DV];
  unsigned xok = 0, dok = 0;
  const bool xbf = d.x_dtype == LVAE_DT_BF16, dybf = d.y_dtype == LVAE_DT_BF16;
  auto prefetch = [&](int tile) {
    const int ig = fastdiv(tile, a.m_tiles_h), th_idx = tile - ig * a.tiles_h;
    const int n0 = ig * a.NI, oh0 = th_idx * a.TH;
    xok = 0;
    if (XB) {
#pragma unroll
      for (int u = 0; u < XV8; ++u) {
        const int px = px8 + 64 * u;
        const int img = fastdiv(px, a.m_per_img), r = px - img * per_img;
        const int hy = fastdiv(r, a.m_halo_w), hx = r - hy * a.halo_w;
        const int n = n0 + img, ih = oh0 + hy - 1, iw = hx - 1;
        const bool ok = (px < a.halo_px) & (n < d.N) & ((unsigned)ih < (unsigned)d.H) & ((unsigned)iw < (unsigned)d.W) & (c8 < Cin);
        const size_t off = ok ? ((size_t)(n * d.H + ih) * d.W + iw) * Cin + c8 : 0;
        xr8[u] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(d.x) + off);
        xok |= ok ? (1u << u) : 0u;
      }
    } else
#pragma unroll
    for (int u = 0; u < XV; ++u) {
      const int px = px0 + 32 * u;
      const int img = fastdiv(px, a.m_per_img), r = px - img * per_img;
      const int hy = fastdiv(r, a.m_halo_w), hx = r - hy * a.halo_w;
      const int n = n0 + img, ih = oh0 + hy - 1, iw = hx - 1;
      const bool ok = (px < a.halo_px) & (n < d.N) & ((unsigned)ih < (unsigned)d.H) & ((unsigned)iw < (unsigned)d.W) & cx_ok;
      const size_t off = ok ? ((size_t)(n * d.H + ih) * d.W + iw) * Cin + c4 : 0;
      xr[u] = load4_dt(d.x, off, xbf);
      xok |= ok ? (1u << u) : 0u;
    }
    dok = 0;
    const int tile_px = a.NI * a.TH * a.TW;
    if (DB) {
#pragma unroll
      for (int u = 0; u < DV8; ++u) {
        const int p = px8 + 64 * u;
        const int img = fastdiv(p, a.m_thw);
        const bool ok = (p < tile_px) & (n0 + img < d.N) & (co0 + c8 < d.Cout);
        const size_t off = ok ? ((size_t)(n0 * d.H + oh0) * d.W + p) * d.Cout + co0 + c8 : 0;
        dr8[u] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(a.dy) + off);
        dok |= ok ? (1u << u) : 0u;
      }
    } else
#pragma unroll
    for (int u = 0; u < DV; ++u) {
      const int p = px0 + 32 * u;
      const int img = fastdiv(p, a.m_thw);
      const bool ok = (p < tile_px) & (n0 + img < d.N) & cd_ok;
      const size_t off = ok ? ((size_t)(n0 * d.H + oh0) * d.W + p) * d.Cout + co0 + c4 : 0;
      dr[u] = load4_dt(a.dy, off, dybf);
      dok |= ok ? (1u << u) : 0u;
    }
  };

  f32x16 acc[5];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f32x4 bsum = zero4, bsum8 = zero4;   // bias-gradient partials (bsum8: channels c8 + 4 .. c8 + 7 of the 8-channel mapping)

  int tile = blockIdx.x;
  if (tile < a.ntiles) prefetch(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    // ---- registers -> LDS images (transform + round to bf16); rows that do not exist are zero
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    if (XB) {
#pragma unroll
      for (int u = 0; u < XV8; ++u) {
        const int px = px8 + 64 * u;
        if (px < a.halo_px) {
          bf16x8 w = xr8[u];
          if (d.in_scale) {
            f32x4 lo = {(float)w[0], (float)w[1], (float)w[2], (float)w[3]}, hi = {(float)w[4], (float)w[5], (float)w[6], (float)w[7]};
            lo = act_fwd4(lo * sc0 + sh0, d.in_act);
            hi = act_fwd4(hi * sc1 + sh1, d.in_act);
            w = to_bf16x8(lo, hi);
          }
          if (!((xok >> u) & 1u)) w = zero8;
          *reinterpret_cast<bf16x8*>(Xs + px * LDK + c8) = w;
        }
      }
    } else
#pragma unroll
    for (int u = 0; u < XV; ++u) {
      const int px = px0 + 32 * u;
      if (px < a.halo_px) {
        f32x4 w = zero4;
        if ((xok >> u) & 1u) {
          w = xr[u];
          if (d.in_scale) w = act_fwd4(w * sc + sh, d.in_act);
        }
        bf16x4 pl[SPLIT];
        split4<SPLIT>(w, pl);
#pragma unroll
        for (int q = 0; q < SPLIT; ++q) *reinterpret_cast<bf16x4*>(Xs + q * x_plane + px * LDK + c4) = pl[q];
      }
    }
    if (DB) {
#pragma unroll
      for (int u = 0; u < DV8; ++u) {
        const int p = px8 + 64 * u;
        const bf16x8 w = ((dok >> u) & 1u) ? dr8[u] : zero8;
        bsum += f32x4{(float)w[0], (float)w[1], (float)w[2], (float)w[3]};
        bsum8 += f32x4{(float)w[4], (float)w[5], (float)w[6], (float)w[7]};
        *reinterpret_cast<bf16x8*>(Ds + p * LDK + c8) = w;
      }
    } else
#pragma unroll
    for (int u = 0; u < DV; ++u) {
      const int p = px0 + 32 * u;
      const f32x4 w = ((dok >> u) & 1u) ? dr[u] : zero4;
      bsum += w;
      bf16x4 pl[SPLIT];
      split4<SPLIT>(w, pl);
#pragma unroll
      for (int q = 0; q < SPLIT; ++q) *reinterpret_cast<bf16x4*>(Ds + q * d_plane + p * LDK + c4) = pl[q];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) prefetch(tile + gridDim.x);

#pragma unroll UNR
    for (int s = 0; s < KS; ++s) {
      const int xr0 = xrow_of(s, 0), xr1 = xrow_of(s, 1);
      bf16x8 bfr[SPLIT];
#pragma unroll
      for (int q = 0; q < SPLIT; ++q)
        bfr[q] = tr_frag(Ds + q * d_plane + (16 * s + drow0) * LDK + chd, Ds + q * d_plane + (16 * s + drow0 + 4) * LDK + chd);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        if (j < ntap) {
          const int tap = tap0 + j, kh = tap / 3, kw = tap - kh * 3;
          const int off = kh * a.halo_w + kw;
          bf16x8 afr[SPLIT];
#pragma unroll
          for (int q = 0; q < SPLIT; ++q) afr[q] = tr_frag(Xs + q * x_plane + (xr0 + off) * LDK + chx, Xs + q * x_plane + (xr1 + off) * LDK + chx);
          if (SPLIT == 1) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0], bfr[0], acc[j], 0, 0, 0);
          } else {  // smallest terms first
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
    __syncthreads();  // the images are read: the next tile overwrites them
  }

  // ---- partial slabs straight from the accumulators (row = ci, 32 consecutive co per lane half)
  float* sw = a.slab_w + (size_t)blockIdx.x * 9 * Cin * d.Cout;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    if (j < ntap) {
      const int tap = tap0 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = cih * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, co = co0 + coh * 32 + li;
        if (ci < Cin && co < d.Cout) sw[((size_t)tap * Cin + ci) * d.Cout + co] = acc[j][r];
      }
    }
  }
  if (a.slab_b) {
    float* red = reinterpret_cast<float*>(smem_raw);   // [32 | 64 pixel groups][64]
    constexpr int NG = DB ? 64 : 32;
    if (DB) {
      *reinterpret_cast<f32x4*>(red + px8 * 64 + c8) = bsum;
      *reinterpret_cast<f32x4*>(red + px8 * 64 + c8 + 4) = bsum8;
    } else {
      *reinterpret_cast<f32x4*>(red + px0 * 64 + c4) = bsum;
    }
    __syncthreads();
    if (t < 64) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g) v += red[g * 64 + t];
      if (co0 + t < d.Cout) a.slab_b[(size_t)blockIdx.x * d.Cout + co0 + t] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Half-slab form of the bf16 weight gradient (round 4; 64 -> 64 channels, 128-pixel tiles, >= 512 tiles). The form above gives every
// workgroup the whole [9][64][64] gradient, so its 256 workgroups write 256 x 147 KB = 37.7 MB of partial slabs per launch and the reduce
// reads them back: ~19 of the 34 us a 256x16x16 gradient costs in the step, for 2 us of MFMAs; and a tile costs it ~5 us (operands fetched
// one tile ahead, two barriers, 4-5 taps per wave), so fewer workgroups with more tiles each lost (2 tiles per workgroup 33.3 ms / step,
// 4: 33.7, 8: 36.1). Here a workgroup owns the 32 INPUT channels `cih` of the gradient over twice the pixels: 128 pixel ranges x 2 halves
// = 256 workgroups, 128 slabs, 18.8 MB written and re-read. It stages only its half of the x patch - the BatchNorm + ELU of the staging is
// vector-ALU work that must not be replicated: a first build with (ci half, co half) quadrants and four waves spent 13 of its 39 us there -
// and the whole dy tile; waves = co half x tap group {0,1,2} {3,4} {5,6} {7,8} (<= 3 accumulator tiles), the LDS images are double-buffered
// (one barrier per tile) and the raw operands are fetched TWO tiles ahead in two register sets.
// The two halves of a pixel range read the same dy lines: same XCD, consecutive dispatch slots. Slab layout and the fixed-order reduce are
// the ones of the form above (the two halves of a range fill one slab): deterministic.
// ---------------------------------------------------------------------------------------------------------------------------
#ifndef LVAE_BFH_LDK
#define LVAE_BFH_LDK 72
#endif
constexpr int BFH_LDK = LVAE_BFH_LDK;   // bf16 elements per LDS row of this kernel's images

template <bool XB, bool DB>
struct BfHRegs {
  bf16x8 x8[XB ? 2 : 1];
  f32x4 x4[XB ? 1 : 4];
  bf16x8 d8[DB ? 2 : 1];
  f32x4 d4[DB ? 1 : 4];
  unsigned xok, dok;
};

template <bool XB, bool DB>
__global__ __launch_bounds__(512) void conv3x3_wgrad_bf16h_kernel(BfWgArgs a) {
  kernarg_warmup<(sizeof(BfWgArgs) < 1024 ? sizeof(BfWgArgs) : 1024)>();
  constexpr int BM = 128, LDK = BFH_LDK, KS = BM / 16;
#ifdef LVAE_BFH_DBG  // compile-time phase-skip mask of the profiling builds (tools/bfq_phase.sh); never defined in the product
  constexpr int dbg = LVAE_BFH_DBG;  // 1: no k-steps, 2: no staging, 4: no global loads, 8: no slab stores, 16: no input transform
#else
  constexpr int dbg = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int img_elems = (a.halo_px + BM) * LDK;   // one image: x patch [halo_px][LDK] (32 channels used) then dy tile [BM][LDK]
  __bf16* const buf0 = reinterpret_cast<__bf16*>(smem_raw);
  const lvae_conv_desc& d = a.d;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int coh = wave & 1, tg = wave >> 1;
  const int tap0 = tg == 0 ? 0 : 2 * tg + 1, ntap = tg == 0 ? 3 : 2;
  const int li = lane & 31, lh = lane >> 5, G = lane >> 4, i16 = lane & 15;
  // workgroup -> (pixel range, ci half): XCD = range % 8, the two halves of a range are consecutive workgroups of that XCD
  const int nrange = gridDim.x >> 1;
  int range, cih;
  if ((nrange & 7) == 0) {
    const int xcd = blockIdx.x & 7, g = blockIdx.x >> 3;
    cih = g & 1;
    range = (g >> 1) * 8 + xcd;
  } else {
    cih = blockIdx.x & 1;
    range = blockIdx.x >> 1;
  }
  const int Cin = a.Cin;   // == 64 == d.Cout (host)
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  const int chx = 16 * (G & 1) + 4 * (i16 & 3);   // channel offset of this lane's 8-byte piece inside the staged x half
  const int chd = coh * 32 + chx;                  // ... inside a dy row
  const int drow0 = 8 * (G >> 1) + (i16 >> 2);
  // element offset (inside an image) of this lane's two transposed reads of k-step s at tap (0, 0): the same for every tile, so the two
  // magic divisions per read are paid once per kernel (16 registers) instead of once per k-step and tile
  int xoff[KS][2];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int p = 16 * s + 8 * (G >> 1) + 4 * rd + (i16 >> 2);
      const int img = fastdiv(p, a.m_thw), r = p - img * (a.TH * a.TW);
      const int ty = fastdiv(r, a.m_tw), tx = r - ty * a.TW;
      xoff[s][rd] = ((img * a.halo_h + ty) * a.halo_w + tx) * LDK + chx;
    }
  int tapoff[3];   // wave-uniform
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int tap = tap0 + (j < ntap ? j : 0), kh = tap / 3, kw = tap - kh * 3;
    tapoff[j] = (kh * a.halo_w + kw) * LDK;
  }

  // ---- staging maps (512 threads). x half, bf16-stored: (pixel t >> 2 + 128 u, 8 channels), fp32-stored: (pixel t >> 3 + 64 u, 4 channels);
  // dy, bf16-stored: (pixel t >> 3 + 64 u, 8 channels), fp32-stored: (pixel t >> 4 + 32 u, 4 channels)
  const int xcl = XB ? (t & 3) * 8 : (t & 7) * 4, xpx = XB ? t >> 2 : t >> 3;
  const int dcl = DB ? (t & 7) * 8 : (t & 15) * 4, dpx = DB ? t >> 3 : t >> 4;
  const int xc = cih * 32 + xcl;   // global channel of this thread's x piece
  f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sh0 = zero4, sc1 = sc0, sh1 = zero4;
  if (d.in_scale) {
    sc0 = *reinterpret_cast<const f32x4*>(d.in_scale + xc);
    sh0 = *reinterpret_cast<const f32x4*>(d.in_shift + xc);
    if (XB) {
      sc1 = *reinterpret_cast<const f32x4*>(d.in_scale + xc + 4);
      sh1 = *reinterpret_cast<const f32x4*>(d.in_shift + xc + 4);
    }
  }
  const int per_img = a.halo_h * a.halo_w;
  const bool xbf = d.x_dtype == LVAE_DT_BF16, dybf = d.y_dtype == LVAE_DT_BF16;
  // tile-invariant part of the staging loads: image / row of the slot inside the tile, element offset relative to the tile's (n0, oh0, 0)
  constexpr int XU = XB ? 2 : 4, DU = DB ? 2 : 4;
  int ximg[XU], xhy[XU], xrel[XU], dimg[DU], drel[DU];
  unsigned xcolok = 0;
#pragma unroll
  for (int u = 0; u < XU; ++u) {
    const int px = xpx + (XB ? 128 : 64) * u;
    const int img = fastdiv(px, a.m_per_img), r = px - img * per_img;
    const int hy = fastdiv(r, a.m_halo_w), hx = r - hy * a.halo_w;
    ximg[u] = img;
    xhy[u] = hy - 1;
    xrel[u] = ((img * d.H + hy - 1) * d.W + hx - 1) * Cin + xc;
    xcolok |= ((px < a.halo_px) & ((unsigned)(hx - 1) < (unsigned)d.W)) ? (1u << u) : 0u;
  }
#pragma unroll
  for (int u = 0; u < DU; ++u) {
    const int p = dpx + (DB ? 64 : 32) * u;
    dimg[u] = fastdiv(p, a.m_thw);
    drel[u] = p * d.Cout + dcl;
  }
  auto prefetch = [&](int tile, BfHRegs<XB, DB>& R) {
    const int ig = fastdiv(tile, a.m_tiles_h), th_idx = tile - ig * a.tiles_h;
    const int n0 = ig * a.NI, oh0 = th_idx * a.TH;
    const int tbase = (n0 * d.H + oh0) * d.W;   // pixel index of the tile's first output pixel (M * 64 < 2^31: host)
    R.xok = 0;
    R.dok = 0;
    if (dbg & 4) {
#pragma unroll
      for (int u = 0; u < (XB ? 2 : 1); ++u) R.x8[u] = zero8;
#pragma unroll
      for (int u = 0; u < (XB ? 1 : 4); ++u) R.x4[u] = zero4;
#pragma unroll
      for (int u = 0; u < (DB ? 2 : 1); ++u) R.d8[u] = zero8;
#pragma unroll
      for (int u = 0; u < (DB ? 1 : 4); ++u) R.d4[u] = zero4;
      R.xok = R.dok = tile & 1;
      return;
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const bool ok = ((xcolok >> u) & 1u) & (n0 + ximg[u] < d.N) & ((unsigned)(oh0 + xhy[u]) < (unsigned)d.H);
      const unsigned off = ok ? (unsigned)(tbase * Cin + xrel[u]) : 0u;
      if (XB) R.x8[XB ? u : 0] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(d.x) + off);
      else R.x4[XB ? 0 : u] = load4_dt(d.x, off, xbf);
      R.xok |= ok ? (1u << u) : 0u;
    }
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const bool ok = n0 + dimg[u] < d.N;
      const unsigned off = ok ? (unsigned)(tbase * d.Cout + drel[u]) : 0u;
      if (DB) R.d8[DB ? u : 0] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(a.dy) + off);
      else R.d4[DB ? 0 : u] = load4_dt(a.dy, off, dybf);
      R.dok |= ok ? (1u << u) : 0u;
    }
  };

  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f32x4 bsum = zero4, bsum8 = zero4;   // bias-gradient partials of this thread's dy channels (bsum8: channels + 4 .. + 7 of an 8-channel piece)

  // registers -> LDS image (transform + round to bf16; rows that do not exist are zero)
  auto stage = [&](BfHRegs<XB, DB>& R, __bf16* Xs) {
    if (dbg & 2) return;
    __bf16* Ds = Xs + a.halo_px * LDK;
#pragma unroll
    for (int u = 0; u < (XB ? 2 : 4); ++u) {
      const int px = xpx + (XB ? 128 : 64) * u;
      if (px < a.halo_px) {
        if (XB) {
          bf16x8 w = R.x8[XB ? u : 0];
          if (d.in_scale && !(dbg & 16)) {
            f32x4 lo = {(float)w[0], (float)w[1], (float)w[2], (float)w[3]}, hi = {(float)w[4], (float)w[5], (float)w[6], (float)w[7]};
            lo = act_fwd4(lo * sc0 + sh0, d.in_act);
            hi = act_fwd4(hi * sc1 + sh1, d.in_act);
            w = to_bf16x8(lo, hi);
          }
          if (!((R.xok >> u) & 1u)) w = zero8;
          *reinterpret_cast<bf16x8*>(Xs + px * LDK + xcl) = w;
        } else {
          f32x4 w = zero4;
          if ((R.xok >> u) & 1u) {
            w = R.x4[XB ? 0 : u];
            if (d.in_scale && !(dbg & 16)) w = act_fwd4(w * sc0 + sh0, d.in_act);
          }
          *reinterpret_cast<bf16x4*>(Xs + px * LDK + xcl) = to_bf16x4(w);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < (DB ? 2 : 4); ++u) {
      const int p = dpx + (DB ? 64 : 32) * u;
      if (DB) {
        const bf16x8 w = ((R.dok >> u) & 1u) ? R.d8[DB ? u : 0] : zero8;
        bsum += f32x4{(float)w[0], (float)w[1], (float)w[2], (float)w[3]};
        bsum8 += f32x4{(float)w[4], (float)w[5], (float)w[6], (float)w[7]};
        *reinterpret_cast<bf16x8*>(Ds + p * LDK + dcl) = w;
      } else {
        const f32x4 w = ((R.dok >> u) & 1u) ? R.d4[DB ? 0 : u] : zero4;
        bsum += w;
        *reinterpret_cast<bf16x4*>(Ds + p * LDK + dcl) = to_bf16x4(w);
      }
    }
  };
  auto mma_tile = [&](const __bf16* Xs) {
    if (dbg & 1) return;
    const __bf16* Ds = Xs + a.halo_px * LDK;
    const __bf16* Dl = Ds + drow0 * LDK + chd;
    // the 48 read addresses of a tile are loop-invariant; hoisted out of the tile loop (as hipcc does) they cost 96 registers and spill.
    // Passing the scalar tap offsets through an empty asm per tile keeps each address one v_add next to its read.
    int tb[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      tb[j] = tapoff[j];
      asm volatile("" : "+s"(tb[j]));
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bf16x8 bfr = tr_frag(Dl + 16 * s * LDK, Dl + (16 * s + 4) * LDK);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (j < ntap) {
          const bf16x8 afr = tr_frag(Xs + xoff[s][0] + tb[j], Xs + xoff[s][1] + tb[j]);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, acc[j], 0, 0, 0);
        }
      }
      if (s & 1) __builtin_amdgcn_sched_barrier(0);   // fragments of two k-steps in flight, not of all eight (registers)
    }
  };

  // tiles range, range + nrange, ...: two register sets, two LDS images
  BfHRegs<XB, DB> R0, R1;
  const int stride = nrange;
  int tile = range;
  if (tile < a.ntiles) prefetch(tile, R0);
  if (tile + stride < a.ntiles) prefetch(tile + stride, R1);
  for (; tile < a.ntiles; tile += 2 * stride) {
    stage(R0, buf0);
    __syncthreads();   // image 0 is published; every wave has left the k-steps of the tile before last, which read image 0
    if (tile + 2 * stride < a.ntiles) prefetch(tile + 2 * stride, R0);
    mma_tile(buf0);
    if (tile + stride >= a.ntiles) break;
    stage(R1, buf0 + img_elems);
    __syncthreads();
    if (tile + 3 * stride < a.ntiles) prefetch(tile + 3 * stride, R1);
    mma_tile(buf0 + img_elems);
  }

  // ---- this half of the range's slab straight from the accumulators (row = ci, 32 consecutive co per lane half)
  float* sw = a.slab_w + (size_t)range * 9 * Cin * d.Cout;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (j < ntap) {
      const int tap = tap0 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = cih * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, co = coh * 32 + li;
        if (!(dbg & 8) || acc[j][r] == 12345.678f) sw[((size_t)tap * Cin + ci) * d.Cout + co] = acc[j][r];
      }
    }
  }
  if (a.slab_b && cih == 0) {   // uniform per workgroup
    __syncthreads();            // every wave is done with the images
    float* red = reinterpret_cast<float*>(smem_raw);   // [pixel groups][64]
    constexpr int NG = DB ? 64 : 32;
    if (DB) {
      *reinterpret_cast<f32x4*>(red + dpx * 64 + dcl) = bsum;
      *reinterpret_cast<f32x4*>(red + dpx * 64 + dcl + 4) = bsum8;
    } else {
      *reinterpret_cast<f32x4*>(red + dpx * 64 + dcl) = bsum;
    }
    __syncthreads();
    if (t < 64) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g) v += red[g * 64 + t];
      a.slab_b[(size_t)range * d.Cout + t] = v;
    }
  }
}

static bool al16b(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- weight pre-split: piece `plane` of w[tap][k][n] (any strides) in MFMA B-fragment order, zero beyond K / N
__device__ __forceinline__ void bf_prep_element(const BfPrepEntry& e, int idx) {
  // idx -> (n, k): consecutive threads take consecutive k (the destination rows are k-contiguous)
  const int k = idx & 63, n = idx >> 6;
  if (n >= e.Npad) return;
  const int ntn = e.Npad >> 6, tn = n >> 6, nn = n & 63;
  for (int tap = 0; tap < 9; ++tap) {
    float r = (n < e.N && k < e.K) ? e.w[tap * e.stap + (int64_t)k * e.sk + (int64_t)n * e.sn] : 0.f;
    // fragment order: [tap][Cout tile][k-step][plane][n half][lane = (k % 16 / 8) * 32 + n % 32][k % 8]
    __bf16* dst = e.U + ((size_t)((tap * ntn + tn) * 4 + (k >> 4)) * e.kind * 2 + (nn >> 5)) * 512 + (((k >> 3) & 1) * 32 + (nn & 31)) * 8 + (k & 7);
    for (int p = 0; p < e.kind; ++p) {
      const __bf16 b = (__bf16)r;
      dst[(size_t)p * 1024] = b;
      r -= (float)b;
    }
  }
}

// 1x1 gate convolution of the fused residual-block kernels (resblock_img.hip), kind = 32 + planes: w[k][n] (K <= 128 reduction channels,
// N <= 128 outputs, any strides) as planes in fragment order [k-step K/16][32-column tile N/32][plane][lane = (k % 16 / 8) * 32 + n % 32][k % 8]
__device__ __forceinline__ void gate_prep_element(const BfPrepEntry& e, int idx) {
  const int planes = e.kind & 31;
  const int k = idx % e.K, n = idx / e.K;
  if (n >= e.N) return;
  float r = e.w[(int64_t)k * e.sk + (int64_t)n * e.sn];
  __bf16* dst = e.U + ((size_t)((k >> 4) * (e.N >> 5) + (n >> 5)) * planes) * 512 + (((k >> 3) & 1) * 32 + (n & 31)) * 8 + (k & 7);
  for (int p = 0; p < planes; ++p) {
    const __bf16 b = (__bf16)r;
    dst[(size_t)p * 512] = b;
    r -= (float)b;
  }
}

__global__ __launch_bounds__(256) void bf_weight_kernel(BfPrepEntry e) {
  if (e.kind & 32) gate_prep_element(e, blockIdx.x * 256 + threadIdx.x);
  else bf_prep_element(e, blockIdx.x * 256 + threadIdx.x);
}

// the entries of a batched pre-transform table (lvae_conv2d_prepare_weights) whose kind != 0; the Winograd kernel takes the others
__global__ __launch_bounds__(256) void bf_weight_batched_kernel(const BfPrepEntry* __restrict__ entries) {
  const BfPrepEntry e = entries[blockIdx.y];
  if (e.kind == 33 || e.kind == 35) {   // 1x1 gate weights of the fused residual-block kernels
    gate_prep_element(e, blockIdx.x * 256 + threadIdx.x);
    return;
  }
  if (e.kind != 1 && e.kind != 3) return;   // 0: Winograd fp32, 16: Winograd six-product form (conv3x3_wino.hip)
  bf_prep_element(e, blockIdx.x * 256 + threadIdx.x);
}

int conv3x3_bf16_prepare_batched(const void* entries, int n, int npad, hipStream_t s) {
  hipLaunchKernelGGL(bf_weight_batched_kernel, dim3((npad * 64 + 255) / 256, n), dim3(256), 0, s, static_cast<const BfPrepEntry*>(entries));
  LVAE_LAUNCH_CHECK("bf_weight_batched");
  return 0;
}

size_t conv3x3_bf16_workspace(const lvae_conv_desc* d, int split) {
  const int ntn = (d->Cout + 63) / 64;
  return (size_t)9 * ntn * split * 4096 * sizeof(__bf16);  // 9 taps x Cout tiles x planes x (64 x 64)
}

void conv3x3_bf16_prep_entry(const lvae_conv_desc* d, int split, void* entry) {
  BfPrepEntry e;
  e.w = d->w;
  e.U = static_cast<__bf16*>(d->workspace);
  e.stap = d->w_stap;
  e.sk = d->w_sk;
  e.sn = d->w_sn;
  e.K = d->C1;
  e.N = d->Cout;
  e.Npad = (d->Cout + 63) / 64 * 64;
  e.flip = 0;
  e.Kpad = 64;
  e.kind = split;
  memcpy(entry, &e, sizeof(e));
}

// one launch that writes the pre-split planes of ONE descriptor (the fallback of a launch whose workspace is not ready; the batched
// form above is what a training step uses)
// table entry / single launch for a 1x1 gate convolution d (C1 = reduction channels, Cout = outputs): planes = 1 | 3
size_t resblock_gate_ws_bytes(const lvae_conv_desc* d, int planes) { return (size_t)d->C1 * d->Cout * planes * sizeof(__bf16); }
void resblock_gate_prep_entry(const lvae_conv_desc* d, int planes, void* entry) {
  BfPrepEntry e;
  e.w = d->w;
  e.U = static_cast<__bf16*>(d->workspace);
  e.stap = 0;
  e.sk = d->w_sk;
  e.sn = d->w_sn;
  e.K = d->C1;
  e.N = d->Cout;
  e.Npad = d->Cout;
  e.flip = 0;
  e.Kpad = d->C1;
  e.kind = 32 + planes;
  memcpy(entry, &e, sizeof(e));
}
int resblock_gate_prepare_single(const lvae_conv_desc* d, int planes, hipStream_t s) {
  BfPrepEntry e;
  resblock_gate_prep_entry(d, planes, &e);
  hipLaunchKernelGGL(bf_weight_kernel, dim3((e.K * e.N + 255) / 256), dim3(256), 0, s, e);
  LVAE_LAUNCH_CHECK("gate_weight");
  return 0;
}

int conv3x3_bf16_prepare_single(const lvae_conv_desc* d, int split, hipStream_t s) {
  BfPrepEntry e;
  conv3x3_bf16_prep_entry(d, split, &e);
  hipLaunchKernelGGL(bf_weight_kernel, dim3((e.Npad * 64 + 255) / 256), dim3(256), 0, s, e);
  LVAE_LAUNCH_CHECK("bf_weight");
  return 0;
}

static size_t bf_lds_bytes(int split, int halo_px, int bm) {
  const size_t in = (size_t)split * halo_px * BF_LDK * 2;
  const size_t out = (size_t)bm * 68 * 4;
  return in > out ? in : out;
}

// tile geometry for a W-wide image (as conv3x3_halo.hip): rows per tile and images per tile so that NI*TH*W <= 128
static bool bf_plan_bm(const lvae_conv_desc* d, int split, int BM, BfArgs& a) {
  const int N = d->N, H = d->H, W = d->W;
  if (W > BM) return false;
  int TH = 1;
  for (int c = 1; c <= H; ++c)
    if (H % c == 0 && c * W <= BM) TH = c;
  int NI = BM / (TH * W);
  if (NI < 1) NI = 1;
  if (TH < H) NI = 1;
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
  a.bm = BM;
  return bf_lds_bytes(split, a.halo_px, BM) <= 160 * 1024;
}

// 128-pixel tiles when their patch leaves room for two workgroups per CU (<= 80 KB), else 64-pixel tiles
static bool bf_plan(const lvae_conv_desc* d, int split, BfArgs& a) {
  if (bf_plan_bm(d, split, 128, a) && bf_lds_bytes(split, a.halo_px, 128) <= 80 * 1024 &&
      (int64_t)((d->N + a.NI - 1) / a.NI) * a.tiles_h * ((d->Cout + 63) / 64) >= 256)  // ... and the grid still fills the chip
    return true;
  if (bf_plan_bm(d, split, 64, a)) return true;
  return bf_plan_bm(d, split, 128, a);
}

static bool bf_select(const lvae_conv_desc* d, int split, BfArgs& a) {
  const int Cin = d->C1;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->x2 != nullptr || d->OH != d->H || d->OW != d->W) return false;
  if (d->in_fold != nullptr) return false;
  if (Cin > 64 || Cin % 4 != 0 || d->Cout % 4 != 0) return false;
  if ((int64_t)d->N * d->H * d->W * Cin >= ((int64_t)1 << 31)) return false;
  if (!al16b(d->x) || !al16b(d->y) || !al16b(d->bias) || !al16b(d->out_scale) || !al16b(d->in_scale) || !al16b(d->in_shift) ||
      !al16b(d->stats_pivot) || !al16b(d->stats_x))
    return false;
  a.d = *d;
  a.Cin = Cin;
  a.flip = d->gather == LVAE_GATHER_TRANSPOSED ? 1 : 0;
  return bf_plan(d, split, a);
}

bool conv3x3_bf16_eligible(const lvae_conv_desc* d, int split) {
  BfArgs a;
  return bf_select(d, split, a);
}

// Which bf16-matrix-pipe form a 3x3 descriptor takes: 1 = bf16 operands (precision LVAE_PREC_BF16), 3 = the fp32-equivalent
// six-product split for the large fp32 layers (form = LVAE_FORM_SIX_PRODUCT_DIRECT only), 0 = neither.
int conv3x3_bf16_form(const lvae_conv_desc* d) {
  if (d->precision == LVAE_PREC_BF16) return conv3x3_bf16_eligible(d, 1) ? 1 : 0;
  // Off by default: measured on MI355X (profiles/r02_conv3x3_forms.txt) the six-product form only ties Winograd-fp32 at 16x16
  // (32.3 vs 31.1 us) and loses at 32x32 (122 vs 100 us): under a dense bf16-MFMA load the chip holds ~1.6 GHz, so 6/16 of the
  // fp32-MFMA cycles is worth ~280 TFLOP/s fp32-equivalent against Winograd's 16/36 on the fp32 pipe. Only when the descriptor
  // asks for it (form = LVAE_FORM_SIX_PRODUCT_DIRECT; the parity tests do).
  const bool split_on = d->form == LVAE_FORM_SIX_PRODUCT_DIRECT;
  static const int64_t min_m = tune("LVAE_F32_SPLIT_MIN_M", 256 * 64);
  if (!split_on || (int64_t)d->N * d->H * d->W < min_m) return 0;
  return conv3x3_bf16_eligible(d, 3) ? 3 : 0;
}

int conv3x3_bf16_stats_rows(const lvae_conv_desc* d, int split) {
  BfArgs a;
  if (d->workspace == nullptr || (size_t)d->workspace_bytes < conv3x3_bf16_workspace(d, split) || !bf_select(d, split, a)) return 0;
  return ((d->N + a.NI - 1) / a.NI) * a.tiles_h;
}

template <int SPLIT, int MI, bool PRE = false, bool XB = false, bool YB = false>
static int launch_bf(BfArgs a, hipStream_t s) {
  auto kern = conv3x3_bf16_kernel<SPLIT, MI, PRE, XB, YB>;
  static std::atomic<bool> attr_set{false};  // idempotent attribute write; the flag itself is race-free
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      set_error("conv3x3_bf16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  const int img_groups = (a.d.N + a.NI - 1) / a.NI;
  a.ntn = (a.d.Cout + 63) / 64;
  a.d.in_fold = nullptr;
  hipLaunchKernelGGL(kern, dim3(img_groups * a.tiles_h * a.ntn), dim3(256), bf_lds_bytes(SPLIT, a.halo_px, 64 * MI), s, a);
  LVAE_LAUNCH_CHECK("conv3x3_bf16");
  return 0;
}

// split = 1: bf16 operands; split = 3: fp32-equivalent six-product form. `d->workspace` holds the pre-split weights
// (conv3x3_bf16_workspace bytes; written here first unless d->workspace_ready). Returns kBfNotEligible when the descriptor or the
// scratch does not fit.
int conv3x3_bf16_try(const lvae_conv_desc* d, int split, hipStream_t s) {
  BfArgs a;
  if (d->workspace == nullptr || (size_t)d->workspace_bytes < conv3x3_bf16_workspace(d, split) || !al16b(d->workspace)) return kBfNotEligible;
  if (!bf_select(d, split, a)) return kBfNotEligible;
  if (!d->workspace_ready) {
    BfPrepEntry e;
    conv3x3_bf16_prep_entry(d, split, &e);
    hipLaunchKernelGGL(bf_weight_kernel, dim3((e.Npad * 64 + 255) / 256), dim3(256), 0, s, e);
  }
  a.Wp = static_cast<const __bf16*>(d->workspace);
  static const int dbg = lvae::debug_phase_switch("LVAE_BF16_DEBUG");  // phase-skip builds (-DLVAE_PHASE_DEBUG) only; 0 in the product
  a.debug = dbg;
  a.ntn = (d->Cout + 63) / 64;
  // (a persistent form of this kernel was built in round 2 and measured slower, 33.8 vs 33.1 ms/step: removed)
  if (split == 1) {
    const int64_t wgs = (int64_t)((d->N + a.NI - 1) / a.NI) * a.tiles_h * a.ntn;
    const bool pre = wgs <= 512;  // one round of two workgroups per CU (measured: 16x16 19.9 -> 18.0 us, 8x8 13.8 -> 9.8; 32x32 53 -> 65 with it)
    const bool xb = d->x_dtype == LVAE_DT_BF16, yb = d->y_dtype == LVAE_DT_BF16;
    if (yb && d->C1 % 8 == 0 && d->Cout % 8 == 0) {   // bf16-stored residual-block tensors: the 16-byte forms
      if (xb) {
        if (a.bm == 128) return pre ? launch_bf<1, 2, true, true, true>(a, s) : launch_bf<1, 2, false, true, true>(a, s);
        return pre ? launch_bf<1, 1, true, true, true>(a, s) : launch_bf<1, 1, false, true, true>(a, s);
      }
      if (a.bm == 128) return pre ? launch_bf<1, 2, true, false, true>(a, s) : launch_bf<1, 2, false, false, true>(a, s);
      return pre ? launch_bf<1, 1, true, false, true>(a, s) : launch_bf<1, 1, false, false, true>(a, s);
    }
    if (a.bm == 128) return pre ? launch_bf<1, 2, true>(a, s) : launch_bf<1, 2>(a, s);
    return pre ? launch_bf<1, 1, true>(a, s) : launch_bf<1, 1>(a, s);
  }
  return a.bm == 128 ? launch_bf<3, 2>(a, s) : launch_bf<3, 1>(a, s);
}

// ---- bf16 weight gradient: host side
// 0: not this kernel; 1: bf16 operands (precision LVAE_PREC_BF16). (The fp32-equivalent six-product form of this kernel, SPLIT = 3, was
// built in round 2 and measured slower than the Winograd-domain fp32 kernel, 39.5 vs 37.9 ms/step: no longer instantiated.)
static int bfwg_form(const lvae_conv_desc* d) {
  static const bool off = tune("LVAE_DISABLE_BF16_WGRAD", 0) != 0;  // A/B switch (tuning builds only)
  if (off) return 0;
  return d->precision == LVAE_PREC_BF16 ? 1 : 0;
}

static bool bfwg_plan(const lvae_conv_desc* d, BfWgArgs& a) {
  const int split = bfwg_form(d);
  if (split == 0) return false;
  a.split = split;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->gather != LVAE_GATHER_CONV || d->x2 != nullptr) return false;
  if (d->OH != d->H || d->OW != d->W || d->C1 > 64 || d->C1 % 4 != 0 || d->Cout % 4 != 0) return false;
  if (!al16b(d->x) || !al16b(d->in_scale) || !al16b(d->in_shift)) return false;
  const int64_t M = (int64_t)d->N * d->H * d->W;
  if (M < 256 * 64 || M * 64 >= ((int64_t)1 << 31)) return false;
  // Layers of exactly 16 k pixels with fp32-stored operands (the 8x8 level at batch 256, whose blocks run the fused whole-image launches with
  // fp32 storage) keep the grouped fp32 Winograd weight gradient: twelve gradients per launch at 9.7 + 2.9 us each (measured in the fp32 step)
  // against 14.1 + 8.3 us for one launch + one slab reduce per gradient here. bf16-stored operands still need this kernel.
  static const int64_t min_m_f32 = tune("LVAE_BF16_WGRAD_MIN_M", 256 * 64 + 1);
  if (M < min_m_f32 && d->x_dtype != LVAE_DT_BF16 && d->y_dtype != LVAE_DT_BF16) return false;
  BfArgs g;
  int bm = 128;
  if (!bf_plan_bm(d, 1, 128, g) || g.halo_px > 224 || (g.NI * g.TH * g.TW) % 16 != 0 ||
      (int64_t)((d->N + g.NI - 1) / g.NI) * g.tiles_h < 256) {   // fewer tiles than CUs: 64-pixel tiles
    bm = 64;
    if (!bf_plan_bm(d, 1, 64, g) || g.halo_px > 224 || (g.NI * g.TH * g.TW) % 16 != 0) return false;
  }
  if (g.NI * g.TH * g.TW != bm) return false;   // whole 16-pixel k-steps only (image widths 8, 16, 32 ...)
  if ((size_t)split * (g.halo_px + bm) * BF_LDK * 2 > 160 * 1024) return false;
  a.TH = g.TH; a.TW = g.TW; a.NI = g.NI; a.tiles_h = g.tiles_h; a.halo_w = g.halo_w; a.halo_h = g.halo_h; a.halo_px = g.halo_px;
  a.m_thw = g.m_thw; a.m_tw = g.m_tw; a.m_per_img = g.m_per_img; a.m_halo_w = g.m_halo_w; a.m_tiles_h = fastdiv_magic(g.tiles_h);
  a.ntiles = ((d->N + g.NI - 1) / g.NI) * g.tiles_h;
  a.Cin = d->C1;
  a.bm = bm;
  return true;
}

// workgroups = split-K ranges = partial slabs (147 KB each for 64 -> 64): the slab write + reduce traffic, not the MFMAs, is what this
// kernel costs; measured on the CIFAR-15 step (bf16 mode): 2 tiles per workgroup 33.3 ms, 4 -> 33.7, 8 -> 36.1
static int bfwg_nwg(const BfWgArgs& a) {
  static const int tpw = (int)tune("LVAE_BF16_WGRAD_TPW", 2);
  int n = (a.ntiles + tpw - 1) / tpw;
  if (n > 256) n = 256;
  if (n < 1) n = 1;
  return n;
}

// pixel ranges (= slabs) of the half-slab form, 0 when the layer keeps the form above: 64 -> 64 channels, 128-pixel tiles, at least 512 of
// them (4 per workgroup at 256x16x16, 16 at 256x32x32)
#ifndef LVAE_BF16_WGRAD_HALF
#define LVAE_BF16_WGRAD_HALF 1
#endif
static int bfwg_half_ranges(const lvae_conv_desc* d, const BfWgArgs& a) {
  static const bool on = tune("LVAE_BF16_WGRAD_HALF", LVAE_BF16_WGRAD_HALF) != 0;  // A/B switch (tuning builds only)
  if (!on || a.split != 1 || d->C1 != 64 || d->Cout != 64 || a.bm != 128 || a.ntiles < 512) return 0;
  if ((size_t)2 * (a.halo_px + 128) * BFH_LDK * 2 > 159 * 1024) return 0;
  // both operands fp32-stored (the layers outside the bf16-storage blocks): two register sets of 4-channel pieces do not fit beside the
  // accumulators (20 registers spilled, 30.5 us against 30.1 us for the form above)
  if (d->x_dtype != LVAE_DT_BF16 && d->y_dtype != LVAE_DT_BF16) return 0;
  return 128;
}

size_t conv3x3_wgrad_bf16_workspace(const lvae_conv_desc* d) {
  BfWgArgs a;
  if (!bfwg_plan(d, a)) return 0;
  const int q = bfwg_half_ranges(d, a);
  return (size_t)(q ? q : bfwg_nwg(a)) * ((size_t)9 * d->C1 * d->Cout + d->Cout) * sizeof(float);
}

void wgrad_reduce_launch(const float* slab_w, const float* slab_b, int ksplit, int ntaps, int Cin, int Cout, int64_t stap, int64_t sk,
                         int64_t sn, float* dw, float* db, hipStream_t s);

// returns kBfNotEligible when the descriptor does not take this kernel
int conv3x3_wgrad_bf16_try(const lvae_conv_desc* d, const float* dy, float* dw, float* db, void* workspace, hipStream_t s) {
  BfWgArgs a;
  if (!bfwg_plan(d, a) || !al16b(dy) || !al16b(workspace)) return kBfNotEligible;
  a.d = *d;
  a.d.in_fold = nullptr;
  a.dy = dy;
  const bool xb = d->x_dtype == LVAE_DT_BF16, dyb = d->y_dtype == LVAE_DT_BF16;
  if (const int nr = bfwg_half_ranges(d, a)) {
    a.slab_w = static_cast<float*>(workspace);
    a.slab_b = db ? a.slab_w + (size_t)nr * 9 * d->C1 * d->Cout : nullptr;
    const size_t qlds = (size_t)2 * (a.halo_px + 128) * BFH_LDK * 2;
    static const hipError_t attr = [] {
      hipError_t e = hipFuncSetAttribute((const void*)conv3x3_wgrad_bf16h_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_wgrad_bf16h_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_wgrad_bf16h_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
      return e;
    }();
    if (attr != hipSuccess) {
      set_error("conv3x3_wgrad_bf16: hipFuncSetAttribute failed: %s", hipGetErrorString(attr));
      return (int)attr;
    }
    const dim3 qgrid(2 * nr);
    if (xb && dyb) hipLaunchKernelGGL((conv3x3_wgrad_bf16h_kernel<true, true>), qgrid, dim3(512), qlds, s, a);
    else if (dyb) hipLaunchKernelGGL((conv3x3_wgrad_bf16h_kernel<false, true>), qgrid, dim3(512), qlds, s, a);
    else hipLaunchKernelGGL((conv3x3_wgrad_bf16h_kernel<true, false>), qgrid, dim3(512), qlds, s, a);
    LVAE_LAUNCH_CHECK("conv3x3_wgrad_bf16h");
    wgrad_reduce_launch(a.slab_w, a.slab_b, nr, 9, d->C1, d->Cout, d->w_stap, d->w_sk, d->w_sn, dw, db, s);
    LVAE_LAUNCH_CHECK("conv3x3_wgrad_bf16_reduce");
    return 0;
  }
  const int nwg = bfwg_nwg(a);
  a.slab_w = static_cast<float*>(workspace);
  a.slab_b = db ? a.slab_w + (size_t)nwg * 9 * d->C1 * d->Cout : nullptr;
  size_t lds = (size_t)a.split * (a.halo_px + a.bm) * BF_LDK * 2;
  if (lds < 32 * 64 * 4) lds = 32 * 64 * 4;
  const dim3 grid(nwg, (d->Cout + 63) / 64);
  if (lds < 64 * 64 * 4) lds = 64 * 64 * 4;   // bias-gradient reduction of the 8-channel mapping
  const bool c8ok = d->C1 % 8 == 0 && d->Cout % 8 == 0;
  if (dyb && c8ok && xb) {
    if (a.bm == 128) hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<2, 1, true, true>), grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<1, 1, true, true>), grid, dim3(512), lds, s, a);
  } else if (dyb && c8ok) {
    if (a.bm == 128) hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<2, 1, false, true>), grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<1, 1, false, true>), grid, dim3(512), lds, s, a);
  } else if (a.bm == 128) hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<2, 1>), grid, dim3(512), lds, s, a);
  else hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<1, 1>), grid, dim3(512), lds, s, a);
  LVAE_LAUNCH_CHECK("conv3x3_wgrad_bf16");
  wgrad_reduce_launch(a.slab_w, a.slab_b, nwg, 9, d->C1, d->Cout, d->w_stap, d->w_sk, d->w_sn, dw, db, s);
  LVAE_LAUNCH_CHECK("conv3x3_wgrad_bf16_reduce");
  return 0;
}

}  // namespace lvae
