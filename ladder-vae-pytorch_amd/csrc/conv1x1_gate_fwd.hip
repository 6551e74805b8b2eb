// GateLayer2d forward of the 64-channel residual blocks (lib/nn.py:118-126 and the residual add of lib/nn.py:99) as a persistent kernel:
//   ab = x . W + bias            x [M][64], W [64][128]; ab written to d->y when the backward needs it
//   out[m][c] = act(ab[m][c]) * sigmoid(ab[m][64 + c]) + res[m][c]
// and, when d->stats_out is set, the BatchNorm partials of `out` (the next residual block's first BatchNorm), one row per workgroup.
//
// The layer is bound by its 85 MB of HBM traffic at 256x16x16 (x and res in, ab and out out; 7 us of fp32 MFMA). conv1x1_kernel ran it as
// 1024 independent 64-pixel workgroups (stage x and W in LDS, MFMA, stage the result in LDS, epilogue), about 1.3 rounds of what the chip
// holds at once, each round paying the whole dependent chain: 29 us. Here
//   * at most 512 workgroups (two per CU), each walking over tiles blockIdx.x, blockIdx.x + gridDim.x, ...;
//   * W lives in registers for the life of the workgroup (64 VGPRs per lane: the B operand of every v_mfma_f32_32x32x2_f32 this wave
//     issues), so the only LDS traffic is the x tile, double-buffered: tile i+1 is fetched while tile i is in the MFMAs and the epilogue;
//   * the a- and b-halves of a channel sit in the same lane (wave wn owns columns wn*32 + lane and 64 + wn*32 + lane), so bias, gate and
//     statistics are lane-local arithmetic on the accumulators and the epilogue needs no workgroup barrier. Two store forms: (WT, the
//     default) each output passes through a 5 KB LDS strip private to the wave, which turns the accumulator layout into 16-byte rows
//     for write-through stores (20.3 us at 256x16x16); (!WT) one dword per lane straight from the accumulators - an accumulator
//     register across a wave is two 128-byte row segments, a full-rate shape for plain stores (22.0 us). The residual rows are requested
//     before the MFMAs of their tile.
#include <stdlib.h>

#include "bf16_frag.h"
#include "lvae_common.h"

namespace lvae {

struct GateFwdArgs {
  const float* x;     // [M][64]
  const float* w;     // element (k, n) at w[k * w_sk + n * w_sn]
  int64_t w_sk, w_sn;
  const float* bias;  // [128] or null
  const float* res;   // [M][64] or null
  float* y;           // [M][128] or null
  float* out;         // [M][64]
  float* stats_out;   // [gridDim.x + 1][2][64] or null (last row: the pivot)
  const float* stats_pivot;
  int M, tiles, act;
  int x_bf16, y_bf16;  // storage of x and of ab (lvae_conv_desc.x_dtype / y_dtype); res and out are fp32
};

constexpr int GF_BM = 64, GF_LDA = 68;

constexpr int GF_LDO = 40;  // staging row stride (floats): 4 rows = 160 = 32 mod 64 banks, so the two lane halves of a write miss each other

// WT: the three outputs leave as 16-byte write-through stores (store_wt4), transposed from the accumulator layout through a 5 KB LDS
// strip that belongs to the wave alone (no workgroup barrier: a wave's LDS instructions execute in order); the residual rows are
// then fetched as float4 in the store layout. !WT: one dword per lane straight from the accumulators, plain stores.
// SPLIT: 0 = v_mfma_f32_32x32x2_f32. 1 (precision = LVAE_PREC_BF16) = x and W rounded to bf16 where they enter the matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate): 8 MFMAs per wave and tile instead of 64, on a unit that - unlike the fp32 MFMA - runs
// beside the vector ALU. 3 (LVAE_GATE_FWD_F32_SPLIT=1, for precision = LVAE_PREC_F32) = both operands split exactly into three bf16 pieces,
// the six piece products of order <= 2^-16 accumulated: fp32-equivalent (same parity tolerances), 6/16 of the fp32 matrix time and off
// the vector ALU's lanes. The fused backward kernel gains 0.34 ms per step from this form; this kernel is bound by its HBM traffic, not by
// its 7 us of fp32 MFMA, and measured 36.97 ms per step with it against 36.86 ms without, so SPLIT = 0 stays the fp32 default (the
// parity tests run both). Everything after the accumulators is the same fp32 code. The x
// tile is a [piece][pixel][channel] bf16 image (pitch 72) in the LDS buffers.
constexpr int GF_LDB = 72;
constexpr int GF_PLANE = GF_BM * GF_LDB;  // bf16 elements per piece
// S16 (SPLIT == 1, WT): x and ab are bf16-stored (residual-block internals under compute_dtype bf16) and move as 16-byte pieces: the x
// tile is two 16-byte loads per thread copied straight into the bf16 LDS image, and ab leaves as 8 channels per lane.
template <bool WT, int SPLIT, bool S16 = false>
__global__ __launch_bounds__(256, 2) void conv1x1_gate_fwd_kernel(GateFwdArgs a) {
  kernarg_warmup<(sizeof(GateFwdArgs) < 1024 ? sizeof(GateFwdArgs) : 1024)>();
  static_assert(!S16 || (WT && SPLIT == 1), "bf16 storage: write-through form, bf16 operands");
  constexpr bool BF16 = SPLIT > 0;
  constexpr int NPIECE = SPLIT > 0 ? SPLIT : 1;
  constexpr int A_FLOATS = SPLIT > 1 ? SPLIT * GF_PLANE / 2 : GF_BM * GF_LDA;  // floats per x buffer (SPLIT <= 1: the fp32 tile covers both)
  __shared__ __attribute__((aligned(16))) float As[2][A_FLOATS];
  __shared__ __attribute__((aligned(16))) float Os[WT ? 4 : 1][WT ? 32 * GF_LDO : 4];
  __shared__ __attribute__((aligned(16))) float red[2][2][64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int ch = wn * 32 + li;  // this lane's channel: column ch of the a half, 64 + ch of the b half
  const int M = a.M;

  // x tile: thread -> rows (t + 256 u) / 16, four channels; rows past M read row 0 and are masked in the epilogue
  f32x4 av[S16 ? 1 : 4];
  bf16x8 av8[S16 ? 2 : 1];
  auto load_a = [&](int tile) {
    if (S16) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = t + 256 * u, r = idx >> 3, k = (idx & 7) * 8;
        const int m = tile * GF_BM + r;
        av8[u] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(a.x) + (size_t)(m < M ? m : 0) * 64 + k);
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = t + 256 * u, r = idx >> 4, k = (idx & 15) * 4;
      const int m = tile * GF_BM + r;
      av[u] = load4_dt(a.x, (size_t)(m < M ? m : 0) * 64 + k, a.x_bf16 != 0);
    }
  };
  auto store_a = [&](float* dst) {
    if (S16) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = t + 256 * u, r = idx >> 3, k = (idx & 7) * 8;
        *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(dst) + r * GF_LDB + k) = av8[u];
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = t + 256 * u, r = idx >> 4, k = (idx & 15) * 4;
      if (BF16) {
        bf16x4 pl[NPIECE];
        split4<NPIECE>(av[u], pl);
#pragma unroll
        for (int q = 0; q < NPIECE; ++q) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(dst) + q * GF_PLANE + r * GF_LDB + k) = pl[q];
      } else {
        *reinterpret_cast<f32x4*>(dst + r * GF_LDA + k) = av[u];
      }
    }
  };

  int tile = blockIdx.x;
  load_a(tile);
  // B operands: k-step s = 4 g + j of group g multiplies A[.][8 g + 4 lh + j] (the four floats of this lane's ds_read_b128)
  // (BF16: k-step s of 16 multiplies A[.][16 s + 8 lh + 0..7], the eight bf16 of this lane's ds_read_b128)
  float breg[2][BF16 ? 1 : 32];
  bf16x8 bq[2][BF16 ? 4 : 1][NPIECE];
  if (BF16) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[j] = a.w[(int64_t)(16 * s + 8 * lh + j) * a.w_sk + (int64_t)(ni * 64 + ch) * a.w_sn];
        bf16x4 lo[NPIECE], hi[NPIECE];
        split4<NPIECE>(f32x4{wv[0], wv[1], wv[2], wv[3]}, lo);
        split4<NPIECE>(f32x4{wv[4], wv[5], wv[6], wv[7]}, hi);
#pragma unroll
        for (int q = 0; q < NPIECE; ++q)
          bq[ni][s][q] = bf16x8{lo[q][0], lo[q][1], lo[q][2], lo[q][3], hi[q][0], hi[q][1], hi[q][2], hi[q][3]};
      }
  } else {
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t ko = (int64_t)(8 * g + 4 * lh + j) * a.w_sk;
        breg[0][4 * g + j] = a.w[ko + (int64_t)ch * a.w_sn];
        breg[1][4 * g + j] = a.w[ko + (int64_t)(64 + ch) * a.w_sn];
      }
  }
  const float bias_a = a.bias ? a.bias[ch] : 0.f, bias_b = a.bias ? a.bias[64 + ch] : 0.f;
  const float piv = a.stats_out ? a.stats_pivot[ch] : 0.f;
  float st1 = 0.f, st2 = 0.f;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 sv1 = zero4, sv2 = zero4, piv4 = zero4;   // WT: statistics in the store layout (four channels per lane)
  const int rr = lane >> 3, c4 = (lane & 7) * 4;  // WT store layout: rows rr + 8 k of the wave's 32, channels wn*32 + c4 .. + 3
  if (WT && a.stats_out) piv4 = *reinterpret_cast<const f32x4*>(a.stats_pivot + wn * 32 + c4);
  float* os = Os[WT ? wave : 0];
  store_a(As[0]);
  __syncthreads();

  int cur = 0;
  for (; tile < a.tiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    const bool has_next = next < a.tiles;
    if (has_next) load_a(next);
    const int m0 = tile * GF_BM + wm * 32 + 4 * lh;
    float rv[WT ? 1 : 16];
    f32x4 rv4[WT ? 4 : 1];
    const int mw = tile * GF_BM + wm * 32 + rr;  // WT: first of this lane's four rows
    if (a.res) {
      if (WT) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int m = mw + 8 * k;
          rv4[k] = *reinterpret_cast<const f32x4*>(a.res + (size_t)(m < M ? m : 0) * 64 + wn * 32 + c4);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (r & 3) + 8 * (r >> 2);
          rv[r] = a.res[(size_t)(m < M ? m : 0) * 64 + ch];
        }
      }
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    if (BF16) {
      // piece products in ascending order of magnitude: (2,0) (0,2) (1,1) (1,0) (0,1) (0,0); one piece: the single product
      constexpr int NP = NPIECE == 1 ? 1 : 6;
      constexpr int PA[6] = {NPIECE - 1, 0, NPIECE > 1 ? 1 : 0, NPIECE > 1 ? 1 : 0, 0, 0};
      constexpr int PB[6] = {0, NPIECE - 1, NPIECE > 1 ? 1 : 0, 0, NPIECE > 1 ? 1 : 0, 0};
      const __bf16* arow = reinterpret_cast<const __bf16*>(As[cur]) + (wm * 32 + li) * GF_LDB + 8 * lh;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 af[NPIECE];
#pragma unroll
        for (int q = 0; q < NPIECE; ++q) af[q] = *reinterpret_cast<const bf16x8*>(arow + q * GF_PLANE + 16 * s);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[k]], bq[0][s][PB[k]], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[k]], bq[1][s][PB[k]], acc1, 0, 0, 0);
        }
      }
    } else {
      const float* arow = As[cur] + (wm * 32 + li) * GF_LDA + 4 * lh;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(arow + 8 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], breg[0][4 * g + j], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], breg[1][4 * g + j], acc1, 0, 0, 0);
        }
      }
    }
    if (WT) {
      // accumulator layout -> strip -> (row, four channels) per lane; three passes over the same strip: a half, b half, gate output
      const int wrow = 4 * lh;  // + (r & 3) + 8 (r >> 2)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc0[r] += bias_a;
        acc1[r] += bias_b;
      }
      const f32x16 ga = act_fwd16(acc0, a.act);   // (one uniform branch, not sixteen switch chains)
#pragma unroll
      for (int pass = 0; pass < 3; ++pass) {
        if (pass < 2 && a.y == nullptr) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = pass == 0 ? acc0[r] : (pass == 1 ? acc1[r] : ga[r] * sigmoidf_(acc1[r]));
          os[(wrow + (r & 3) + 8 * (r >> 2)) * GF_LDO + li] = v;
        }
        __builtin_amdgcn_wave_barrier();
        if (S16 && pass < 2) {  // ab as bf16: 8 channels per lane, rows (lane >> 2) + 16 k of the wave's 32
          const int r16 = lane >> 2, c8 = (lane & 3) * 8;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int row = r16 + 16 * k, m = tile * GF_BM + wm * 32 + row;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(os + row * GF_LDO + c8), v1 = *reinterpret_cast<const f32x4*>(os + row * GF_LDO + c8 + 4);
            if (m < M) {
              const bf16x8 o = to_bf16x8(v0, v1);
              asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1"
                           : : "v"(reinterpret_cast<__bf16*>(a.y) + (size_t)m * 128 + pass * 64 + wn * 32 + c8), "v"(o) : "memory");
            }
          }
          __builtin_amdgcn_wave_barrier();
          continue;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          f32x4 v = *reinterpret_cast<const f32x4*>(os + (rr + 8 * k) * GF_LDO + c4);
          const int m = mw + 8 * k;
          if (pass == 2) {
            if (a.res) v += rv4[k];
            if (m < M) {
              store_wt4(a.out + (size_t)m * 64 + wn * 32 + c4, v);
              const f32x4 dl = v - piv4;
              sv1 += dl;
              sv2 += dl * dl;
            }
          } else if (m < M) {
            store4_dt(a.y, (size_t)m * 128 + pass * 64 + wn * 32 + c4, v, a.y_bf16 != 0);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (r & 3) + 8 * (r >> 2);
      if (m < M) {
        const float va = acc0[r] + bias_a, vb = acc1[r] + bias_b;
        if (a.y) {  // (this store form is fp32 only: the launcher takes the WT form for bf16 storage)
          a.y[(size_t)m * 128 + ch] = va;
          a.y[(size_t)m * 128 + 64 + ch] = vb;
        }
        float o = act_fwd(va, a.act) * sigmoidf_(vb);
        if (a.res) o += rv[r];
        a.out[(size_t)m * 64 + ch] = o;
        const float dl = o - piv;
        st1 += dl;
        st2 += dl * dl;
      }
    }
    if (has_next) store_a(As[cur ^ 1]);
    __syncthreads();  // tile i+1 is in LDS; buffer `cur` is free again from the NEXT iteration's store on
    cur ^= 1;
  }

  if (a.stats_out) {  // lanes, then the two row waves, in a fixed order: one row of partials per workgroup
    if (WT) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sv1[j] += __shfl_xor(sv1[j], o, 64);
          sv2[j] += __shfl_xor(sv2[j], o, 64);
        }
      if (lane < 8) {
        *reinterpret_cast<f32x4*>(&red[0][wm][wn * 32 + c4]) = sv1;
        *reinterpret_cast<f32x4*>(&red[1][wm][wn * 32 + c4]) = sv2;
      }
    } else {
      st1 += __shfl_xor(st1, 32, 64);
      st2 += __shfl_xor(st2, 32, 64);
      if (lh == 0) {
        red[0][wm][ch] = st1;
        red[1][wm][ch] = st2;
      }
    }
    __syncthreads();
    if (t < 128) {
      const int c = t & 63, which = t >> 6;
      a.stats_out[((size_t)blockIdx.x * 2 + which) * 64 + c] = red[which][0][c] + red[which][1][c];
      // the pivot travels with the partials (row gridDim.x): a consumer that finalizes them in its own prologue (lvae_bn_fold) must not
      // depend on a buffer it updates itself
      if (blockIdx.x == 0 && which == 0) a.stats_out[((size_t)gridDim.x * 2) * 64 + c] = a.stats_pivot[c];
    }
  }
}

// workgroups the persistent kernel runs for this descriptor (== rows of BatchNorm partials it writes); 0: not this kernel
int conv1x1_gate_fwd_wgs(const lvae_conv_desc* d) {
  static const bool off = tune("LVAE_DISABLE_GATE_FWD_PERSISTENT", 0) != 0;  // A/B switch (tuning builds only)
  static const int max_wgs = (int)tune("LVAE_GATE_FWD_WGS", 512);
  static const int64_t min_m = tune("LVAE_GATE_FWD_MIN_M", 0);
  if (off || d == nullptr) return 0;
  if (d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0 || d->OH != d->H || d->OW != d->W || d->gather != LVAE_GATHER_CONV) return 0;
  if (d->C1 != 64 || d->C2 != 0 || d->x2 != nullptr || d->Cout != 128) return 0;
  if (d->in_scale != nullptr || d->in_fold != nullptr || d->out_scale != nullptr || d->out_act != LVAE_ACT_NONE) return 0;
  if (d->x == nullptr || (reinterpret_cast<uintptr_t>(d->x) & 15) != 0) return 0;
  const int64_t M = (int64_t)d->N * d->H * d->W;
  if (M < min_m || M >= ((int64_t)1 << 31) - 64) return 0;
  const int64_t tiles = (M + GF_BM - 1) / GF_BM;
  const int cap = max_wgs > 0 ? max_wgs : 512;
  return (int)(tiles < cap ? tiles : cap);
}

// -1000: not eligible (the caller uses conv1x1_kernel)
int conv1x1_gate_fwd_try(const lvae_conv_desc* d, const float* res, float* out, int act, hipStream_t s) {
  const int wgs = conv1x1_gate_fwd_wgs(d);
  if (wgs == 0 || out == nullptr) return -1000;
  GateFwdArgs a;
  a.x = d->x;
  a.w = d->w;
  a.w_sk = d->w_sk;
  a.w_sn = d->w_sn;
  a.bias = d->bias;
  a.res = res;
  a.y = d->y;
  a.out = out;
  a.stats_out = d->stats_out;
  a.stats_pivot = d->stats_pivot;
  a.M = (int)((int64_t)d->N * d->H * d->W);
  a.tiles = (a.M + GF_BM - 1) / GF_BM;
  a.act = act;
  a.x_bf16 = d->x_dtype == LVAE_DT_BF16;
  a.y_bf16 = d->y_dtype == LVAE_DT_BF16;
  static const bool wt = tune("LVAE_GATE_FWD_WT", 1) != 0;  // A/B switch (tuning builds only)
  const bool al = ((reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(d->y) |
                    reinterpret_cast<uintptr_t>(d->stats_pivot)) & 15) == 0;
  // fp32: on the fp32 MFMA by default (the kernel is HBM-bound: the six-product form measured the same); d->form = LVAE_FORM_SIX_PRODUCT asks for it
  const int split = d->precision == LVAE_PREC_BF16 ? 1 : (d->form == LVAE_FORM_SIX_PRODUCT ? 3 : 0);
  const dim3 grid(wgs), block(256);
  if ((a.x_bf16 || a.y_bf16) && !(split == 1 && al)) return -1000;  // bf16 storage: bf16-operand form, aligned buffers
  if ((wt || a.x_bf16 || a.y_bf16) && al) {
    if (split == 1 && a.x_bf16 && (a.y_bf16 || a.y == nullptr)) hipLaunchKernelGGL((conv1x1_gate_fwd_kernel<true, 1, true>), grid, block, 0, s, a);
    else if (split == 1) hipLaunchKernelGGL((conv1x1_gate_fwd_kernel<true, 1>), grid, block, 0, s, a);
    else if (split == 3) hipLaunchKernelGGL((conv1x1_gate_fwd_kernel<true, 3>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv1x1_gate_fwd_kernel<true, 0>), grid, block, 0, s, a);
  } else {
    if (split == 1) hipLaunchKernelGGL((conv1x1_gate_fwd_kernel<false, 1>), grid, block, 0, s, a);
    else if (split == 3) hipLaunchKernelGGL((conv1x1_gate_fwd_kernel<false, 3>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv1x1_gate_fwd_kernel<false, 0>), grid, block, 0, s, a);
  }
  LVAE_LAUNCH_CHECK("conv1x1_gate_fwd");
  return 0;
}

}  // namespace lvae
