// GateLayer2d backward for C = 64 as ONE persistent kernel: gate derivative, input gradient of the 1x1 convolution AND its weight /
// bias gradient (autograd of lib/nn.py:118-126 of the reference).
//
//   dab[m, c]     = dout[m, c] * sigmoid(b) * act'(a)            (a, b = the halves of the saved pre-activations ab [M, 128])
//   dab[m, 64+c]  = dout[m, c] * act(a) * sigmoid(b) * (1 - sigmoid(b))
//   dx[m, ci]     = (sum_co dab[m, co] * W[co, ci]) * drop[n(m), ci]
//   dW[co, ci]   += sum_m  y[m, ci] * dab[m, co]                 (y = the convolution input saved by the forward)     db[co] += sum_m dab[m, co]
//
// Before, this was two launches and three passes over memory: the fused gate-backward + dgrad kernel wrote dab (33.5 MB at
// 256x16x16) for the weight-gradient kernel, which read it back together with y. Here dab lives only in LDS: a workgroup loops
// over 64-pixel tiles (persistent, one per CU), forms dab once, uses it as the A operand of the dgrad GEMM and as the B operand of
// the weight-gradient GEMM, whose 64 x 128 accumulator stays in registers across all tiles of the workgroup. Eight waves: waves
// 0-3 run the dgrad GEMM of a tile while waves 4-7 run its weight-gradient GEMM (two waves per SIMD cover each other's LDS latency;
// both read the same dab tile); the raw operands of the next tile are prefetched into registers while the MFMAs run. HBM traffic per gated block drops
// from 150 MB to 84 MB and one launch disappears. Per-workgroup weight-gradient partials go to the usual split-K slabs
// ([workgroup][64][128] + [workgroup][128]) and are summed in a fixed order (wgrad_reduce_launch): deterministic.
//
// Where the 33.7 us at 256x16x16 go (phase-skip builds, round 2): 15.8 us are the 128 fp32 MFMAs per tile and SIMD (13.6 us at 2.4 GHz),
// 6.2 us the gate arithmetic, 2.3 us the dx stores, 1.6 us operand fetch that the prefetch does not hide, the rest launch, prologue and
// the slab write. Two restructurings were built and measured, and neither is kept:
//   * one barrier per tile (double-buffered dab / y tiles, W in registers, dx through wave-private LDS strips): 35.8 us against 35.2 us;
//   * twelve waves with the gate arithmetic on four producer waves (one per SIMD, beside a dgrad and a weight-gradient wave): 38.6 us.
//     Alone the producers need 1.8 us per tile and the MFMA waves 3.2 us; together a tile takes 5.9 us - more than their sum.
//     v_mfma_f32_32x32x2_f32 and the vector ALU do not overlap on this chip (the fp32 matrix rate equals the packed-fp32 vector rate:
//     the same lanes), so in an fp32 kernel VALU time adds to MFMA time whichever wave issues it. The bf16 MFMAs have their own unit.
#include "bf16_frag.h"
#include "lvae_common.h"

namespace lvae {

void wgrad_reduce_launch(const float* slab_w, const float* slab_b, int ksplit, int ntaps, int Cin, int Cout, int64_t stap, int64_t sk,
                         int64_t sn, float* dw, float* db, hipStream_t s);

struct GbfArgs {
  const float* dout;   // [M][64]
  const float* ab;     // [M][128]
  const float* y;      // [M][64]
  const float* w;      // element (k = co, n = ci) at w[k*w_sk + n*w_sn], w_sk == 1 (k-contiguous rows of 128)
  int64_t w_sn;
  const float* drop;   // [N][64] or null
  float* dx;           // [M][64]
  float* slab_w;       // [nwg][64 ci][128 co]
  float* slab_b;       // [nwg][128] or null
  int M, ohw, ntiles, act;
  int in_bf16, dx_bf16;  // storage of ab and y (lvae_conv_desc.x_dtype) and of dx (y_dtype); dout is fp32. bf16-operand kernel only.
  lvae_bn_apply ap;      // deferred BatchNorm-backward apply that produces dout (ap.parts == nullptr: dout is given)
};

constexpr int GB_LDA = 132;  // dab / weight row pitch (floats): conflict-free ds_read_b128 (as conv1x1.hip)
constexpr int GB_LDY = 68;   // y / dx staging row pitch

__global__ __launch_bounds__(512) void conv1x1_gate_bwd_fused_kernel(GbfArgs a) {
  kernarg_warmup<(sizeof(GbfArgs) < 1024 ? sizeof(GbfArgs) : 1024)>();
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Bs = smem;                       // [64 ci][132]: W[co][ci] as B[k = co][n = ci], k-contiguous
  float* As = Bs + 64 * GB_LDA;           // [64 px][132]: dab tile (A of the dgrad, B of the weight gradient); later dx staging
  float* Ys = As + 64 * GB_LDA;           // [64 px][68]: y tile (A^T of the weight gradient)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool wg_role = wave >= 4;  // waves 4-7: weight gradient; waves 0-3: dgrad
  const int wm = (wave & 3) >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- weights once per workgroup
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int idx = t + 512 * u, n = idx >> 5, k = (idx & 31) * 4;
    *reinterpret_cast<f32x4*>(Bs + n * GB_LDA + k) = *reinterpret_cast<const f32x4*>(a.w + (int64_t)n * a.w_sn + k);
  }

  // ---- raw operands of one tile: thread -> 2 items (row r = idx / 16, channels c = (idx % 16) * 4) of dout, a, b and y
  constexpr int IT = 2;
  f32x4 pg[IT], pa[IT], pb[IT], py[IT];
  const int c4 = (t & 15) * 4, r0 = t >> 4;
  auto prefetch = [&](int tile) {
    const int m0 = tile * 64;
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int m = m0 + r0 + 32 * u;
      const size_t mc = m < a.M ? (size_t)m : 0;  // clamped address; the values of rows past the end are zeroed below
      pg[u] = *reinterpret_cast<const f32x4*>(a.dout + mc * 64 + c4);
      pa[u] = *reinterpret_cast<const f32x4*>(a.ab + mc * 128 + c4);
      pb[u] = *reinterpret_cast<const f32x4*>(a.ab + mc * 128 + 64 + c4);
      py[u] = *reinterpret_cast<const f32x4*>(a.y + mc * 64 + c4);
    }
  };

  // weight-gradient accumulators of this wave: ci block (wave & 1), co blocks 2 * (wave >> 1) + {0, 1}
  f32x16 accw[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[j][r] = 0.f;
  f32x4 bs_lo = zero4, bs_hi = zero4;  // bias-gradient partials of this thread's channels (c4.. and 64 + c4..)

  int tile = blockIdx.x;
  if (tile < a.ntiles) prefetch(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    const int m0 = tile * 64;
    // ---- gate derivative from the prefetched registers -> LDS (dab tile, y tile)
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int r = r0 + 32 * u;
      f32x4 lo = zero4, hi = zero4, yv = zero4;
      if (m0 + r < a.M) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float sg = sigmoidf_(pb[u][j]);
          lo[j] = pg[u][j] * sg * act_grad(pa[u][j], a.act);
          hi[j] = pg[u][j] * act_fwd(pa[u][j], a.act) * sg * (1.f - sg);
        }
        yv = py[u];
      }
      bs_lo += lo;
      bs_hi += hi;
      *reinterpret_cast<f32x4*>(As + r * GB_LDA + c4) = lo;
      *reinterpret_cast<f32x4*>(As + r * GB_LDA + 64 + c4) = hi;
      *reinterpret_cast<f32x4*>(Ys + r * GB_LDY + c4) = yv;
    }
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) prefetch(tile + gridDim.x);  // in flight during the 128 MFMAs below

    f32x16 accx;
    if (!wg_role) {
      // ---- dgrad (waves 0-3): dx[32 px (wm)][32 ci (wn)] = dab[px][0:128] . W
#pragma unroll
      for (int r = 0; r < 16; ++r) accx[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 128; kk += 8) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(As + (wm * 32 + li) * GB_LDA + kk + 4 * lh);
        const f32x4 bf = *reinterpret_cast<const f32x4*>(Bs + (wn * 32 + li) * GB_LDA + kk + 4 * lh);
#pragma unroll
        for (int j = 0; j < 4; ++j) accx = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bf[j], accx, 0, 0, 0);
      }
    } else {
      // ---- weight gradient (waves 4-7): dW[32 ci (wn)][2 x 32 co (wm)] += y^T . dab over the 64 pixels of the tile (k = pixel)
#pragma unroll 8
      for (int p = 0; p < 64; p += 2) {
        const float ya = Ys[(p + lh) * GB_LDY + wn * 32 + li];
        const float b0 = As[(p + lh) * GB_LDA + (wm * 2) * 32 + li];
        const float b1 = As[(p + lh) * GB_LDA + (wm * 2 + 1) * 32 + li];
        accw[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ya, b0, accw[0], 0, 0, 0);
        accw[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ya, b1, accw[1], 0, 0, 0);
      }
    }
    __syncthreads();  // dab / y tiles are dead: the dab region becomes the dx staging tile [64][68]

    float* Os = As;
    if (!wg_role) {
#pragma unroll
      for (int r = 0; r < 16; ++r) Os[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * GB_LDY + wn * 32 + li] = accx[r];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int r = r0 + 32 * u, m = m0 + r;
      if (m < a.M) {
        f32x4 v = *reinterpret_cast<const f32x4*>(Os + r * GB_LDY + c4);
        if (a.drop) v = v * *reinterpret_cast<const f32x4*>(a.drop + (size_t)(m / a.ohw) * 64 + c4);
        store_wt4(a.dx + (size_t)m * 64 + c4, v);
      }
    }
    __syncthreads();  // staging tile is read: the next iteration overwrites it
  }

  // ---- this workgroup's weight / bias gradient partials -> slabs (128-byte row segments straight from the accumulators)
  float* sw = a.slab_w + (size_t)blockIdx.x * 64 * 128;
  if (wg_role) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sw[(size_t)(wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 128 + (wm * 2 + j) * 32 + li] = accw[j][r];
  }
  if (a.slab_b) {
    float* red = As;  // [32 row groups][128] (16 KB; every tile is done)
    *reinterpret_cast<f32x4*>(red + r0 * 128 + c4) = bs_lo;
    *reinterpret_cast<f32x4*>(red + r0 * 128 + 64 + c4) = bs_hi;
    __syncthreads();
    if (t < 128) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 32; ++g) v += red[g * 128 + t];
      a.slab_b[(size_t)blockIdx.x * 128 + t] = v;
    }
  }
}

// The same kernel on the bf16 matrix unit (v_mfma_f32_32x32x16_bf16, fp32 accumulate), which - unlike the fp32 MFMA - runs beside the
// vector ALU. The gate derivative itself, the bias gradient and everything that leaves the kernel stay fp32.
//   SPLIT = 1 (precision = LVAE_PREC_BF16): dab, y and W are rounded to bf16 where they enter the matrix cores.
//   SPLIT = 3 (precision = LVAE_PREC_F32, the default fp32 form): each operand is split exactly into three bf16 pieces and the six piece
//     products of order <= 2^-16 are accumulated (fp32-equivalent, the dropped terms are below 2^-24 of the product; same tolerances in
//     the parity tests as the fp32 MFMA). Six bf16 MFMAs cost 6/16 of one fp32 MFMA and no longer add to the gate arithmetic:
//     LVAE_GATE_BWD_F32_MFMA=1 goes back to conv1x1_gate_bwd_fused_kernel above.
// dab / y tiles are [piece][pixel][channel] bf16 images; the dgrad reads its A fragments row-wise (ds_read_b128, W fragments in
// registers of the dgrad waves), the weight gradient needs 8 consecutive PIXELS of one channel per lane for both operands and reads
// them with the transposing ds_read_b64_tr_b16 (bf16_frag.h), as conv3x3_wgrad_bf16_kernel does.
constexpr int GBB_LDA = 136;  // dab row pitch in bf16 (128 channels + 8: 272 bytes, 16-byte multiples for ds_read_b128)
constexpr int GBB_LDY = 72;   // y row pitch in bf16
// + the deferred apply's coefficient rows [6][64] and, in the six-product form, the dgrad's pre-split W fragments [8 k-steps][2 column halves][3][64 lanes][8]
// (48 KB: in registers they were 96 VGPRs of every wave of a kernel at the 256-register cap; with the deferred apply's extra operand rows it spilled)
constexpr size_t gbb_lds(int split) {
  return (size_t)split * (64 * GBB_LDA + 64 * GBB_LDY) * 2 + (size_t)64 * GB_LDY * 4 + 6 * 64 * 4 + (split == 3 ? (size_t)8 * 2 * 3 * 64 * 16 : 0);
}

// S16 (SPLIT == 1): ab, y and dx are bf16-stored (residual-block internals under compute_dtype bf16) and move as 16-byte pieces: a thread
// takes 8 channels of ONE pixel row per tile (5 loads of 16 bytes instead of 8, one 16-byte store instead of two of 8).
// AP (round 5): dout is not given but is the result of the BatchNorm-backward apply that ends the backward of the residual block that ran
// just before (its input is this block's output): dout = BN'(ap.dh; ap.x) + ap.add. The persistent workgroups reduce the producer's
// partial rows once (fixed order), then form dout while they stage a tile and store it to ap.out (this block's own last apply reads it as
// its `add`); workgroup 0 accumulates dgamma / dbeta. Saves the apply launch, its finalize launch and one 16.8 MB pass per gated block.
// ELU: both activation ids of the launch are LVAE_ACT_ELU (the model's default), known at compile time. With run-time ids every one of the
// staging loop's 16 per-element activation calls is a chain of scalar compares and taken branches (the emitted staging segment of the
// round-4 kernel held ~300 branch instructions; resblock_img.hip found the same in round 4).
template <int SPLIT, bool S16 = false, bool AP = false, bool ELU = false>
__global__ __launch_bounds__(512) void conv1x1_gate_bwd_fused_bf16_kernel(GbfArgs a) {
  kernarg_warmup<sizeof(GbfArgs)>();
#ifdef LVAE_GBF_DBG  // compile-time phase-skip mask of the profiling builds (tools/gbf_ab.sh); never defined in the product
  constexpr int dbg = LVAE_GBF_DBG;  // 1: no global loads, 2: no MFMAs, 4: no global stores, 8: no gate derivative arithmetic
#else
  constexpr int dbg = 0;
#endif
  const int gact = ELU ? LVAE_ACT_ELU : a.act, apact = ELU ? LVAE_ACT_ELU : a.ap.act;
  static_assert(SPLIT == 1 || !S16, "bf16 storage exists for the bf16-operand form only");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int A_PLANE = 64 * GBB_LDA, Y_PLANE = 64 * GBB_LDY;
  __bf16* As = reinterpret_cast<__bf16*>(smem_raw);                 // [SPLIT][64 px][136]: dab tile
  __bf16* Ys = As + SPLIT * A_PLANE;                                 // [SPLIT][64 px][72]: y tile
  float* Os = reinterpret_cast<float*>(Ys + SPLIT * Y_PLANE);        // [64 px][68]: dx staging (fp32)
  float* Cf = Os + 64 * GB_LDY;                                      // [6][64]: scale, shift, mean, rstd, mean(g), mean(g xhat) of the deferred apply
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool wg_role = wave >= 4;  // waves 4-7: weight gradient; waves 0-3: dgrad
  const int wm = (wave & 3) >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5, G = lane >> 4, i16 = lane & 15;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  if (AP) {
    // the producer's partial rows [rows][2][64], summed by 16 row groups (8 independent 16-byte loads in flight per thread), then in double
    const int q = t & 31, rg = t >> 5, rows = a.ap.rows;
    f32x4 acc = zero4;
    for (int r = rg; r < rows; r += 16 * 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int rr = r + 16 * u;
        v[u] = *reinterpret_cast<const f32x4*>(a.ap.parts + (size_t)(rr < rows ? rr : 0) * 128 + q * 4);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r + 16 * u < rows) acc += v[u];
    }
    *reinterpret_cast<f32x4*>(Os + rg * 128 + q * 4) = acc;   // [16][128] <= the staging tile
    if (t < 256) Cf[t] = a.ap.coef[t];
    __syncthreads();
    if (t < 64) {
      double sa = 0.0, sb = 0.0;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        sa += (double)Os[g * 128 + t];
        sb += (double)Os[g * 128 + 64 + t];
      }
      Cf[256 + t] = (float)(sa / (double)a.ap.M);
      Cf[320 + t] = (float)(sb / (double)a.ap.M);
      if (blockIdx.x == 0) {
        if (a.ap.dbeta) a.ap.dbeta[t] += (float)sa;
        if (a.ap.dgamma) a.ap.dgamma[t] += (float)sb;
      }
    }
    __syncthreads();
  }
  // dout[m][c .. c + 3] of the deferred apply from (dh, x, add) and the coefficient rows in LDS
  auto ap_value = [&](const f32x4 dh, const f32x4 xv, const f32x4 ad, int c) {
    const f32x4 sc = *reinterpret_cast<const f32x4*>(Cf + c), sh = *reinterpret_cast<const f32x4*>(Cf + 64 + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(Cf + 128 + c), rs = *reinterpret_cast<const f32x4*>(Cf + 192 + c);
    const f32x4 c1 = *reinterpret_cast<const f32x4*>(Cf + 256 + c), c2 = *reinterpret_cast<const f32x4*>(Cf + 320 + c);
    const f32x4 u = xv * sc + sh;
    f32x4 g;
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = dh[j] * act_grad(u[j], apact);
    return (g - c1 - (xv - mu) * rs * c2) * sc + ad;
  };

  // dgrad waves: W[co = 16 s + 8 lh + 0..7][ci = wn*32 + li] as the B fragment of k-step s, SPLIT pieces: in registers (bf16 operands) or,
  // in the six-product form, in LDS in fragment order (written once by waves 0 / 1, published by the first barrier below)
  constexpr bool WLDS = SPLIT == 3;
  bf16x8* Wf = reinterpret_cast<bf16x8*>(Cf + 6 * 64);   // [8][2][SPLIT][64]
  bf16x8 breg[WLDS ? 1 : 8][SPLIT];
  if (!wg_role && (!WLDS || wm == 0)) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float* wp = a.w + (int64_t)(wn * 32 + li) * a.w_sn + 16 * s + 8 * lh;
      bf16x4 lo[SPLIT], hi[SPLIT];
      split4<SPLIT>(*reinterpret_cast<const f32x4*>(wp), lo);
      split4<SPLIT>(*reinterpret_cast<const f32x4*>(wp + 4), hi);
#pragma unroll
      for (int q = 0; q < SPLIT; ++q) {
        const bf16x8 v = bf16x8{lo[q][0], lo[q][1], lo[q][2], lo[q][3], hi[q][0], hi[q][1], hi[q][2], hi[q][3]};
        if (WLDS) Wf[((s * 2 + wn) * SPLIT + q) * 64 + lane] = v;
        else breg[s][q] = v;
      }
    }
  }

  constexpr int IT = S16 ? 1 : 2;
  f32x4 pg[2], pa[IT], pb[IT], py[IT];
  f32x4 px[AP ? 2 : 1], pd[AP ? 2 : 1];   // deferred apply: x and add rows (pg then holds dh)
  bf16x8 qa, qb, qy;   // S16: ab[c8 .. c8 + 7], ab[64 + c8 ..], y[c8 ..] of row r8
  const int c4 = (t & 15) * 4, r0 = t >> 4;
  const int c8 = (t & 7) * 8, r8 = t >> 3;
  const bool ap_add = AP && a.ap.add != nullptr, ap_dh_bf = AP && a.ap.dh_bf16 != 0;
  // one of the thread's two rows of a tile (fp32-stored operands): requested as soon as the row before it in the same registers has been
  // consumed, so that the loads of tile t + 1 fly under the rest of tile t (the loop's barriers wait for LDS only)
  auto prefetch_row = [&](int tile, int u) {
    if (dbg & 1) {
      pg[u] = pa[u] = pb[u] = py[u] = zero4;
      if (AP) px[u] = pd[u] = zero4;
      return;
    }
    const int m = tile * 64 + r0 + 32 * u;
    const size_t mc = m < a.M ? (size_t)m : 0;  // clamped address; the values of rows past the end are zeroed below
    if (AP) {
      pg[u] = load4_dt(a.ap.dh, mc * 64 + c4, ap_dh_bf);
      px[u] = *reinterpret_cast<const f32x4*>(a.ap.x + mc * 64 + c4);
      pd[u] = ap_add ? *reinterpret_cast<const f32x4*>(a.ap.add + mc * 64 + c4) : zero4;
    } else {
      pg[u] = *reinterpret_cast<const f32x4*>(a.dout + mc * 64 + c4);
    }
    pa[u] = load4_dt(a.ab, mc * 128 + c4, a.in_bf16 != 0);
    pb[u] = load4_dt(a.ab, mc * 128 + 64 + c4, a.in_bf16 != 0);
    py[u] = load4_dt(a.y, mc * 64 + c4, a.in_bf16 != 0);
  };
  auto prefetch = [&](int tile) {
    const int m0 = tile * 64;
    if (dbg & 1) {
      pg[0] = pg[1] = zero4;
#pragma unroll
      for (int u = 0; u < IT; ++u) pa[u] = pb[u] = py[u] = zero4;
#pragma unroll
      for (int u = 0; u < (AP ? 2 : 1); ++u) px[u] = pd[u] = zero4;
      qa = qb = qy = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      return;
    }
    if (S16) {
      const int m = m0 + r8;
      const size_t mc = m < a.M ? (size_t)m : 0;
      if (AP) {
        pg[0] = load4_dt(a.ap.dh, mc * 64 + c8, ap_dh_bf);
        pg[1] = load4_dt(a.ap.dh, mc * 64 + c8 + 4, ap_dh_bf);
        px[0] = *reinterpret_cast<const f32x4*>(a.ap.x + mc * 64 + c8);
        px[1] = *reinterpret_cast<const f32x4*>(a.ap.x + mc * 64 + c8 + 4);
        pd[0] = ap_add ? *reinterpret_cast<const f32x4*>(a.ap.add + mc * 64 + c8) : zero4;
        pd[1] = ap_add ? *reinterpret_cast<const f32x4*>(a.ap.add + mc * 64 + c8 + 4) : zero4;
      } else {
      pg[0] = *reinterpret_cast<const f32x4*>(a.dout + mc * 64 + c8);
      pg[1] = *reinterpret_cast<const f32x4*>(a.dout + mc * 64 + c8 + 4);
      }
      qa = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(a.ab) + mc * 128 + c8);
      qb = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(a.ab) + mc * 128 + 64 + c8);
      qy = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(a.y) + mc * 64 + c8);
      return;
    }
#pragma unroll
    for (int u = 0; u < IT; ++u) prefetch_row(tile, u);
  };

  f32x16 accw[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[j][r] = 0.f;
  f32x4 bs_lo = zero4, bs_hi = zero4;  // bias-gradient partials (fp32 dab, not the rounded operand)
  f32x4 bs_lo8 = zero4, bs_hi8 = zero4;  // S16: channels c8 + 4 .. c8 + 7
  // transposed-read addresses of this lane (bf16_frag.h): pixel row 8 (G >> 1) + (i16 >> 2) of a k-step, channel 16 (G & 1) + 4 (i16 & 3) of a block
  const int trow = 8 * (G >> 1) + (i16 >> 2), tch = 16 * (G & 1) + 4 * (i16 & 3);
  // piece products in ascending order of magnitude: (2,0) (0,2) (1,1) (1,0) (0,1) (0,0); SPLIT = 1: the single product
  constexpr int NP = SPLIT == 1 ? 1 : 6;
  constexpr int PA[6] = {SPLIT - 1, 0, SPLIT > 1 ? 1 : 0, SPLIT > 1 ? 1 : 0, 0, 0};
  constexpr int PB[6] = {0, SPLIT - 1, SPLIT > 1 ? 1 : 0, 0, SPLIT > 1 ? 1 : 0, 0};

  int tile = blockIdx.x;
  if (tile < a.ntiles) prefetch(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    const int m0 = tile * 64;
    if (S16) {
      f32x4 lo[2] = {zero4, zero4}, hi[2] = {zero4, zero4};
      bf16x8 yv = {0, 0, 0, 0, 0, 0, 0, 0};
      if (m0 + r8 < a.M) {
        if (AP) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            pg[h] = ap_value(pg[h], px[h], pd[h], c8 + 4 * h);
            store_wt4(a.ap.out + (size_t)(m0 + r8) * 64 + c8 + 4 * h, pg[h]);
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float av = (float)qa[4 * h + j], bv = (float)qb[4 * h + j];
            const float sg = sigmoidf_(bv);
            lo[h][j] = pg[h][j] * sg * act_grad(av, gact);
            hi[h][j] = pg[h][j] * act_fwd(av, gact) * sg * (1.f - sg);
          }
        yv = qy;
      }
      bs_lo += lo[0];
      bs_lo8 += lo[1];
      bs_hi += hi[0];
      bs_hi8 += hi[1];
      *reinterpret_cast<bf16x8*>(As + r8 * GBB_LDA + c8) = to_bf16x8(lo[0], lo[1]);
      *reinterpret_cast<bf16x8*>(As + r8 * GBB_LDA + 64 + c8) = to_bf16x8(hi[0], hi[1]);
      *reinterpret_cast<bf16x8*>(Ys + r8 * GBB_LDY + c8) = yv;
    } else
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int r = r0 + 32 * u;
      f32x4 lo = zero4, hi = zero4, yv = zero4;
      if (m0 + r < a.M) {
        if (AP) {
          pg[u] = ap_value(pg[u], px[u], pd[u], c4);
          if (!(dbg & 4)) store_wt4(a.ap.out + (size_t)(m0 + r) * 64 + c4, pg[u]);
        }
        if (dbg & 8) {
          lo = pg[u] + pa[u];
          hi = pg[u] + pb[u];
        } else
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float sg = sigmoidf_(pb[u][j]);
          lo[j] = pg[u][j] * sg * act_grad(pa[u][j], gact);
          hi[j] = pg[u][j] * act_fwd(pa[u][j], gact) * sg * (1.f - sg);
        }
        yv = py[u];
      }
      bs_lo += lo;
      bs_hi += hi;
      if (tile + (int)gridDim.x < a.ntiles) prefetch_row(tile + gridDim.x, u);
      bf16x4 pl[SPLIT], ph[SPLIT], pyv[SPLIT];
      split4<SPLIT>(lo, pl);
      split4<SPLIT>(hi, ph);
      split4<SPLIT>(yv, pyv);
#pragma unroll
      for (int q = 0; q < SPLIT; ++q) {
        *reinterpret_cast<bf16x4*>(As + q * A_PLANE + r * GBB_LDA + c4) = pl[q];
        *reinterpret_cast<bf16x4*>(As + q * A_PLANE + r * GBB_LDA + 64 + c4) = ph[q];
        *reinterpret_cast<bf16x4*>(Ys + q * Y_PLANE + r * GBB_LDY + c4) = pyv[q];
      }
    }
    if (S16 && tile + (int)gridDim.x < a.ntiles) prefetch(tile + gridDim.x);
    lds_barrier();  // the dab / y tiles are published; the prefetched rows stay in flight

    if (!wg_role) {
      // ---- dgrad: dx[32 px (wm)][32 ci (wn)] = dab[px][0:128] . W, 8 k-steps of 16
      f32x16 accx;
#pragma unroll
      for (int r = 0; r < 16; ++r) accx[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        bf16x8 af[SPLIT];
#pragma unroll
        for (int q = 0; q < SPLIT; ++q)
          af[q] = *reinterpret_cast<const bf16x8*>(As + q * A_PLANE + (wm * 32 + li) * GBB_LDA + 16 * s + 8 * lh);
        if (WLDS) {
          bf16x8 bw[SPLIT];
#pragma unroll
          for (int q = 0; q < SPLIT; ++q) bw[q] = Wf[((s * 2 + wn) * SPLIT + q) * 64 + lane];
#pragma unroll
          for (int k = 0; k < NP; ++k) { if (dbg & 2) accx[k] += (float)af[PA[k]][0] * (float)bw[PB[k]][1]; else accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[k]], bw[PB[k]], accx, 0, 0, 0); }
        } else {
#pragma unroll
          for (int k = 0; k < NP; ++k) accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[k]], breg[s][PB[k]], accx, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) Os[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * GB_LDY + wn * 32 + li] = accx[r];
    } else {
      // ---- weight gradient: dW[32 ci (wn)][2 x 32 co (wm)] += y^T . dab, k = pixel: 4 k-steps of 16 pixels
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 af[SPLIT];
#pragma unroll
        for (int q = 0; q < SPLIT; ++q) {
          const __bf16* yp = Ys + q * Y_PLANE + (16 * s + trow) * GBB_LDY + wn * 32 + tch;
          af[q] = tr_frag(yp, yp + 4 * GBB_LDY);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bf16x8 bf[SPLIT];
#pragma unroll
          for (int q = 0; q < SPLIT; ++q) {
            const __bf16* dp = As + q * A_PLANE + (16 * s + trow) * GBB_LDA + (wm * 2 + j) * 32 + tch;
            bf[q] = tr_frag(dp, dp + 4 * GBB_LDA);
          }
#pragma unroll
          for (int k = 0; k < NP; ++k) { if (dbg & 2) accw[j][k] += (float)af[PA[k]][0] * (float)bf[PB[k]][1]; else accw[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[k]], bf[PB[k]], accw[j], 0, 0, 0); }
        }
      }
    }
    lds_barrier();  // dab / y tiles are dead (the next iteration overwrites them), the dx staging tile is complete
    if (S16) {
      const int m = m0 + r8;
      if (m < a.M) {
        f32x4 v0 = *reinterpret_cast<const f32x4*>(Os + r8 * GB_LDY + c8), v1 = *reinterpret_cast<const f32x4*>(Os + r8 * GB_LDY + c8 + 4);
        if (a.drop) {
          v0 = v0 * *reinterpret_cast<const f32x4*>(a.drop + (size_t)(m / a.ohw) * 64 + c8);
          v1 = v1 * *reinterpret_cast<const f32x4*>(a.drop + (size_t)(m / a.ohw) * 64 + c8 + 4);
        }
        const bf16x8 o = to_bf16x8(v0, v1);
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(reinterpret_cast<__bf16*>(a.dx) + (size_t)m * 64 + c8), "v"(o) : "memory");
      }
    } else
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int r = r0 + 32 * u, m = m0 + r;
      if (m < a.M) {
        f32x4 v = *reinterpret_cast<const f32x4*>(Os + r * GB_LDY + c4);
        if (a.drop) v = v * *reinterpret_cast<const f32x4*>(a.drop + (size_t)(m / a.ohw) * 64 + c4);
        if (!(dbg & 4) || v[0] == 12345.678f) store4_dt(a.dx, (size_t)m * 64 + c4, v, a.dx_bf16 != 0);
      }
    }
    // no barrier here: the next write to the staging tile comes after the next iteration's first barrier
  }

  float* sw = a.slab_w + (size_t)blockIdx.x * 64 * 128;
  if (wg_role) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sw[(size_t)(wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 128 + (wm * 2 + j) * 32 + li] = accw[j][r];
  }
  if (a.slab_b) {
    __syncthreads();  // the last tile's dx staging reads are done
    if (S16) {
      float* red = reinterpret_cast<float*>(smem_raw);  // [64 row groups][128] floats = 32 KB <= the 43 KB of the three tiles
      *reinterpret_cast<f32x4*>(red + r8 * 128 + c8) = bs_lo;
      *reinterpret_cast<f32x4*>(red + r8 * 128 + c8 + 4) = bs_lo8;
      *reinterpret_cast<f32x4*>(red + r8 * 128 + 64 + c8) = bs_hi;
      *reinterpret_cast<f32x4*>(red + r8 * 128 + 64 + c8 + 4) = bs_hi8;
      __syncthreads();
      if (t < 128) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 64; ++g) v += red[g * 128 + t];
        a.slab_b[(size_t)blockIdx.x * 128 + t] = v;
      }
      return;
    }
    float* red = Os;  // [32 row groups][128] floats = 16 KB <= the 17 KB staging tile
    *reinterpret_cast<f32x4*>(red + r0 * 128 + c4) = bs_lo;
    *reinterpret_cast<f32x4*>(red + r0 * 128 + 64 + c4) = bs_hi;
    __syncthreads();
    if (t < 128) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 32; ++g) v += red[g * 128 + t];
      a.slab_b[(size_t)blockIdx.x * 128 + t] = v;
    }
  }
}

static bool al16f(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// workgroups (= weight-gradient slabs) for M pixels: one per CU, fewer when there are fewer tiles
static int gbf_nwg(int64_t M) {
  const int64_t ntiles = (M + 63) / 64;
  return (int)(ntiles < 256 ? ntiles : 256);
}

// `d` describes the dgrad of the gate convolution exactly as for lvae_conv1x1_gate_bwd_f32 (C1 = 2C = 128, Cout = C = 64, weight
// strides of the transposed view, y = dx, out_scale = dropout mask). 0 when this kernel does not take the shape.
size_t conv1x1_gate_bwd_fused_workspace(const lvae_conv_desc* d) {
  static const bool off = tune("LVAE_DISABLE_GATE_FUSED", 0) != 0;  // A/B switch (tuning builds only)
  if (off || d == nullptr) return 0;
  if (d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0 || d->OH != d->H || d->OW != d->W) return 0;
  if (d->C1 != 128 || d->C2 != 0 || d->Cout != 64 || d->w_sk != 1 || d->w_sn % 4 != 0 || d->in_scale != nullptr) return 0;
  const int64_t M = (int64_t)d->N * d->H * d->W;
  static const int64_t min_m = tune("LVAE_GATE_FUSED_MIN_M", 256 * 64);
  if (M < min_m || M >= ((int64_t)1 << 31)) return 0;
  return (size_t)gbf_nwg(M) * (64 * 128 + 128) * sizeof(float);
}

int conv1x1_gate_bwd_fused(const lvae_conv_desc* d, const float* dout, const float* ab, const float* y, int act, float* dw,
                           int64_t dw_sk, int64_t dw_sn, float* db, void* workspace, const lvae_bn_apply* ap, hipStream_t s) {
  if (!al16f(dout) || !al16f(ab) || !al16f(y) || !al16f(d->w) || !al16f(d->y) || !al16f(d->out_scale) || !al16f(workspace)) return -1000;
  GbfArgs a;
  a.ap = lvae_bn_apply{};
  if (ap != nullptr && ap->parts != nullptr) {
    if (!al16f(ap->parts) || !al16f(ap->coef) || !al16f(ap->dh) || !al16f(ap->x) || !al16f(ap->add) || !al16f(ap->out)) return -1000;
    a.ap = *ap;
  }
  a.dout = dout;
  a.ab = ab;
  a.y = y;
  a.w = d->w;
  a.w_sn = d->w_sn;
  a.drop = d->out_scale;
  a.dx = d->y;
  a.M = d->N * d->H * d->W;
  a.ohw = d->H * d->W;
  a.ntiles = (a.M + 63) / 64;
  a.act = act;
  a.in_bf16 = d->x_dtype == LVAE_DT_BF16;
  a.dx_bf16 = d->y_dtype == LVAE_DT_BF16;
  if ((a.in_bf16 || a.dx_bf16) && d->precision != LVAE_PREC_BF16) return -1000;  // bf16 storage exists in the bf16-operand kernel only
  const int nwg = gbf_nwg(a.M);
  a.slab_w = static_cast<float*>(workspace);
  a.slab_b = db ? a.slab_w + (size_t)nwg * 64 * 128 : nullptr;
  constexpr size_t lds = (size_t)(2 * 64 * GB_LDA + 64 * GB_LDY) * sizeof(float);
  static std::atomic<bool> attr_set{false};  // idempotent attribute write; the flag itself is race-free
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_gate_bwd_fused_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("conv1x1_gate_bwd_fused: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  const bool f32_mfma = d->form == LVAE_FORM_F32_MFMA;  // default for fp32: the six-product form on the bf16 MFMA
  const bool with_ap = a.ap.parts != nullptr;
  if (with_ap && (f32_mfma && d->precision != LVAE_PREC_BF16)) {
    set_error("conv1x1_gate_bwd_fused: the deferred BatchNorm-backward apply exists in the bf16-matrix-pipe kernels only (not with LVAE_FORM_F32_MFMA)");
    return LVAE_EINVAL;
  }
  const bool elu = act == LVAE_ACT_ELU && (!with_ap || a.ap.act == LVAE_ACT_ELU);
#define GBF_LAUNCH(SP, S16_, AP_, LDS_)                                                                                                   \
  do {                                                                                                                                    \
    if (elu) hipLaunchKernelGGL((conv1x1_gate_bwd_fused_bf16_kernel<SP, S16_, AP_, true>), dim3(nwg), dim3(512), LDS_, s, a);            \
    else hipLaunchKernelGGL((conv1x1_gate_bwd_fused_bf16_kernel<SP, S16_, AP_, false>), dim3(nwg), dim3(512), LDS_, s, a);               \
  } while (0)
  if (d->precision == LVAE_PREC_BF16) {
    if (with_ap) {
      if (a.in_bf16 && a.dx_bf16) GBF_LAUNCH(1, true, true, gbb_lds(1));
      else GBF_LAUNCH(1, false, true, gbb_lds(1));
    } else if (a.in_bf16 && a.dx_bf16) GBF_LAUNCH(1, true, false, gbb_lds(1));
    else GBF_LAUNCH(1, false, false, gbb_lds(1));
  } else if (!f32_mfma) {
    static std::atomic<bool> attr3_set{false};
    if (!attr3_set) {
      hipError_t e = hipSuccess;
      const void* ks[4] = {(const void*)conv1x1_gate_bwd_fused_bf16_kernel<3, false, false, false>, (const void*)conv1x1_gate_bwd_fused_bf16_kernel<3, false, false, true>,
                           (const void*)conv1x1_gate_bwd_fused_bf16_kernel<3, false, true, false>, (const void*)conv1x1_gate_bwd_fused_bf16_kernel<3, false, true, true>};
      for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)gbb_lds(3));
      if (e != hipSuccess) {
        set_error("conv1x1_gate_bwd_fused: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return (int)e;
      }
      attr3_set = true;
    }
    if (with_ap) GBF_LAUNCH(3, false, true, gbb_lds(3));
    else GBF_LAUNCH(3, false, false, gbb_lds(3));
#undef GBF_LAUNCH
  } else {
    hipLaunchKernelGGL(conv1x1_gate_bwd_fused_kernel, dim3(nwg), dim3(512), lds, s, a);
  }
  LVAE_LAUNCH_CHECK("conv1x1_gate_bwd_fused");
  // slabs are [nwg][ci][co]: the gate convolution's weight element (ci, co) lives at dw[ci * dw_sk + co * dw_sn]
  wgrad_reduce_launch(a.slab_w, a.slab_b, nwg, 1, 64, 128, 0, dw_sk, dw_sn, dw, db, s);
  LVAE_LAUNCH_CHECK("conv1x1_gate_bwd_fused_reduce");
  return 0;
}

}  // namespace lvae

using namespace lvae;

extern "C" size_t lvae_conv1x1_gate_bwd_wgrad_workspace(const lvae_conv_desc* d) { return conv1x1_gate_bwd_fused_workspace(d); }

extern "C" int lvae_conv1x1_gate_bwd_wgrad_f32(const lvae_conv_desc* d, const float* dout, const float* ab, const float* y, int32_t act,
                                               float* dw, int64_t dw_sk, int64_t dw_sn, float* db, void* workspace,
                                               size_t workspace_bytes, const lvae_bn_apply* ap, void* stream) {
  const bool with_ap = ap != nullptr && ap->parts != nullptr;
  LVAE_REQUIRE(d && (dout || with_ap) && ab && y && d->y && dw && workspace, LVAE_EINVAL, "lvae_conv1x1_gate_bwd_wgrad_f32: null pointer");
  if (with_ap)
    LVAE_REQUIRE(ap->rows > 0 && ap->M == (int64_t)d->N * d->H * d->W && ap->coef && ap->dh && ap->x && ap->out, LVAE_EINVAL,
                 "lvae_conv1x1_gate_bwd_wgrad_f32: deferred apply needs parts / rows, M = N*H*W, the coefficient block, dh, x and out");
  const size_t need = conv1x1_gate_bwd_fused_workspace(d);
  LVAE_REQUIRE(need > 0, LVAE_EINVAL,
               "lvae_conv1x1_gate_bwd_wgrad_f32: unsupported shape (needs the gate of a 64-channel block: 1x1, 128 -> 64 dgrad view, at "
               "least 16384 pixels); use lvae_conv1x1_gate_bwd_f32 + lvae_conv2d_wgrad_f32");
  LVAE_REQUIRE(workspace_bytes >= need, LVAE_EWORKSPACE, "lvae_conv1x1_gate_bwd_wgrad_f32: workspace %zu < %zu", workspace_bytes, need);
  const int rc = conv1x1_gate_bwd_fused(d, dout, ab, y, act, dw, dw_sk, dw_sn, db, workspace, ap, (hipStream_t)stream);
  LVAE_REQUIRE(rc != -1000, LVAE_EALIGN, "lvae_conv1x1_gate_bwd_wgrad_f32: buffers must be 16-byte aligned");
  return rc;
}
