// Fused residual-block kernels for the low-resolution levels of the ladder (H*W <= 64: the 8x8, 4x4 and 2x2 levels, 11 of the 15
// stochastic layers of the CIFAR model), "whole-image tiles".
//
// At these sizes a workgroup's 64 (or 128) GEMM rows are WHOLE IMAGES (1-2 at 8x8, 4-8 at 4x4, 16-32 at 2x2), so a 3x3 convolution
// needs no pixel of another workgroup, and everything a residual block (lib/nn.py:78-99,118-126) does between two BatchNorm
// reductions can stay inside one launch with its intermediate tile in LDS:
//
//   forward   conv1:  [BN1 finalize] BN1+act -> conv1 -> bias, Dropout2d -> y1, BN2 partial sums            (PRO_AFFINE / EPI_PLAIN)
//             conv2:  [BN2 finalize] BN2+act -> conv2 -> bias, Dropout2d -> y2 -> 1x1 gate GEMM -> ab,
//                     act(a)*sigmoid(b) + x -> out, next block's BN1 partial sums                            (PRO_AFFINE / EPI_GATE)
//   backward  B1:     gate derivative (dout, ab) -> dab -> 1x1 dgrad GEMM, Dropout2d -> dy2 -> dgrad conv2
//                     -> dh2, BN2-backward partial sums                                                      (PRO_GATE_BWD / EPI_PLAIN)
//             B2:     BN2-backward finalize + apply (dh2, y1), Dropout2d -> dy1 -> dgrad conv1 -> dh1,
//                     BN1-backward partial sums                                                              (PRO_BN_APPLY / EPI_PLAIN)
//
// i.e. two launches per block and direction where the one-kernel-per-op form had three (forward) and five (backward: fused gate
// backward, dgrad, apply, dgrad, apply); y2 / dab / dy2 / dy1 are still written (the weight gradients read them) but never read back
// by this chain. The only cross-workgroup dependency inside a block is the 2 x 64-float BatchNorm reduction, which is where the
// launches are cut: a dependent kernel boundary costs ~1.5 us on this chip, a grid barrier 4-7 (MI355X_MICROARCH.md, price list).
//
// Arithmetic: conv3x3_bf16.hip's core — the patch of the tile (zero ring included) staged once in LDS as SPLIT bf16 planes, weights
// pre-split once per step in MFMA fragment order and streamed from L2 (no weights in LDS, no barrier in the reduction loop),
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation. SPLIT = 3 is the fp32-equivalent six-product form (same parity tolerances as
// every other fp32 form of the library), SPLIT = 1 bf16 operands (precision = LVAE_PREC_BF16). The 1x1 gate GEMMs run the same way
// with their weights split in registers.
#include <string.h>

#include "bf16_frag.h"
#include "lvae_common.h"

namespace lvae {

struct RbArgs {
  lvae_conv_desc d;
  lvae_bn_fold f;  // copy of *d.in_fold (f.parts == nullptr: none)
  lvae_rb_ext e;
  const __bf16* Wp;  // pre-split 3x3 weights in fragment order [tap][k-step 4][plane][n half 2][lane 64][8] (bf_weight_kernel)
  int NI, HW, halo_w, halo_h, halo_px, flip, nwg;
  uint32_t m_hw, m_w, m_per_img, m_halo_w;
};

constexpr int RB_LDK = 72;            // bf16 elements per patch row (64 channels + 8 pad = 144 bytes)
constexpr int RB_LDD = 136;           // bf16 elements per dab row (128 channels + 8 pad = 272 bytes)
constexpr int RB_LDO = 68;            // floats per staged output row (64 + 4)
constexpr int RB_LDG = 132;           // floats per staged gate pre-activation row (128 + 4)
constexpr int RB_SCR_BYTES = 9216;    // reduction scratch in front of the tile regions

// piece products of the six-product form in ascending order of magnitude: (2,0) (0,2) (1,1) (1,0) (0,1) (0,0)
template <int SPLIT>
__device__ __forceinline__ f32x16 mfma_pieces(const bf16x8 (&af)[SPLIT], const bf16x8 (&bf)[SPLIT], f32x16 acc) {
  if (SPLIT == 1) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[SPLIT - 1], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[SPLIT - 1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc, 0, 0, 0);
  }
  return acc;
}

// eight fp32 weights of one B fragment -> SPLIT bf16x8 pieces
template <int SPLIT>
__device__ __forceinline__ void split_frag(const float (&wv)[8], bf16x8 (&out)[SPLIT]) {
  bf16x4 lo[SPLIT], hi[SPLIT];
  split4<SPLIT>(f32x4{wv[0], wv[1], wv[2], wv[3]}, lo);
  split4<SPLIT>(f32x4{wv[4], wv[5], wv[6], wv[7]}, hi);
#pragma unroll
  for (int q = 0; q < SPLIT; ++q) out[q] = bf16x8{lo[q][0], lo[q][1], lo[q][2], lo[q][3], hi[q][0], hi[q][1], hi[q][2], hi[q][3]};
}

// In-kernel phase stamps of the profiling builds (-DLVAE_RB_DBG; tools/rb_stamps.sh): s_memtime per wave at the phase boundaries, written to
// a buffer of their own that nothing else reads. Never compiled into the product.
#ifdef LVAE_RB_DBG
__device__ unsigned long long g_rb_stamps[1024 * 4 * 12];
// the stamps stay in scalar registers until the end of the kernel: a store per stamp would sit in the wave's in-order memory counter and
// turn every later s_waitcnt vmcnt into a wait for that store's trip to memory (the first form of this instrument did exactly that)
#define RB_STAMP_DECL unsigned long long rb_ts[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define RB_STAMP(i) rb_ts[i] = __builtin_amdgcn_s_memtime()
#define RB_STAMP_FLUSH                                                                                             \
  do {                                                                                                             \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024)                                                              \
      for (int i_ = 0; i_ < 12; ++i_) g_rb_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 12 + i_] = rb_ts[i_];    \
  } while (0)
#else
#define RB_STAMP_DECL do {} while (0)
#define RB_STAMP(i) do {} while (0)
#define RB_STAMP_FLUSH do {} while (0)
#endif

// LDS-only workgroup barrier. __syncthreads() is a workgroup fence, for which hipcc waits for EVERY outstanding memory operation of the
// wave (s_waitcnt vmcnt(0)): global loads that were requested early on purpose, and the write-through stores of the epilogues, whose
// completion takes a trip to memory (measured with the phase stamps: 1.2-2 us per barrier behind such stores). All data that crosses
// waves inside these kernels goes through LDS, so only the LDS counter has to drain.
__device__ __forceinline__ void rb_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
// four fp32 values -> SPLIT bf16x4 pieces that sum to them exactly (SPLIT = 3) / their round-to-nearest bf16 (SPLIT = 1); pairwise, so
// that hipcc emits one v_cvt_pk_bf16_f32 per two values and stage (bf16_frag.h split8_3)
template <int SPLIT>
__device__ __forceinline__ void rb_split4(const f32x4 v, bf16x4 (&out)[SPLIT]) {
  u32x2v w[SPLIT];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    float a = v[2 * p], b = v[2 * p + 1];
#pragma unroll
    for (int q = 0; q < SPLIT; ++q) {
      const bf16x2 h = __builtin_convertvector(f32x2v{a, b}, bf16x2);
      const unsigned bits = __builtin_bit_cast(unsigned, h);
      w[q][p] = bits;
      if (q + 1 < SPLIT) {
        a -= __builtin_bit_cast(float, bits << 16);  // exact: the remainder of a round-to-nearest to 8 bits has <= 16 significant bits
        b -= __builtin_bit_cast(float, bits & 0xffff0000u);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < SPLIT; ++q) out[q] = __builtin_bit_cast(bf16x4, w[q]);
}

// activation derivative w.r.t. the pre-activation, four values, ONE wave-uniform branch for the common case
__device__ __forceinline__ f32x4 rb_act_grad4(f32x4 u, int act) {
  f32x4 r;
  if (act == LVAE_ACT_ELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = u[j] > 0.f ? 1.f : __expf(u[j]);
    return r;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = act_grad(u[j], act);
  return r;
}

// Sum of the partial rows [rows][2][64] a producer's statistics epilogue left, by all 256 threads with 16-byte loads, in a fixed order;
// the 8 row groups are combined in double by threads 0..63, which hand (channel, sum 1, sum 2) to `finish` between the two barriers.
// Two calls: rb_parts_issue requests the first 256 rows (32 loads per thread in flight) and must come BEFORE the kernel requests its big
// operand rows: the memory counter of a wave retires in order, so small L2-resident loads issued behind cold HBM loads can only be waited
// for together with them (in the step the sums used to arrive ~3 us late for exactly that reason: phase 0-1 of tools/rb_stamps_instep.py);
// rb_parts_finish consumes them (and loops over rows beyond 256, for producers with more workgroups).
constexpr int RB_PF = 32;
__device__ __forceinline__ void rb_parts_issue(const float* __restrict__ parts, int rows, f32x4 (&v)[RB_PF]) {
  const int t = threadIdx.x, q = t & 31, rg = t >> 5;   // q: 16-byte piece of a 128-float row; rg: row group (rows rg, rg + 8, ...)
#pragma unroll
  for (int u = 0; u < RB_PF; ++u) {
    const int rr = rg + 8 * u;
    v[u] = *reinterpret_cast<const f32x4*>(parts + (size_t)(rr < rows ? rr : 0) * 128 + q * 4);
  }
}
template <typename F>
__device__ __forceinline__ void rb_parts_finish(const float* __restrict__ parts, int rows, const f32x4 (&v0)[RB_PF], float* scr, F&& finish,
                                                unsigned long long* ts = nullptr) {
  const int t = threadIdx.x, q = t & 31, rg = t >> 5;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < RB_PF; ++u)
    if (rg + 8 * u < rows) acc += v0[u];
#ifdef LVAE_RB_DBG
  if (ts) { asm volatile("" :: "v"(acc[0])); ts[8] = __builtin_amdgcn_s_memtime(); }
#endif
  for (int r = rg + 8 * RB_PF; r < rows; r += 8 * 16) {
    f32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int rr = r + 8 * u;
      v[u] = *reinterpret_cast<const f32x4*>(parts + (size_t)(rr < rows ? rr : 0) * 128 + q * 4);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (r + 8 * u < rows) acc += v[u];
  }
  *reinterpret_cast<f32x4*>(scr + rg * 128 + q * 4) = acc;   // [8][128]
  rb_bar();
#ifdef LVAE_RB_DBG
  if (ts) ts[9] = __builtin_amdgcn_s_memtime();
#endif
  if (t < 64) {
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      sa += (double)scr[g * 128 + t];
      sb += (double)scr[g * 128 + 64 + t];
    }
    finish(t, sa, sb);
  }
  rb_bar();
}

// The tile is 64 pixels (whole images) x 64 channels per workgroup, 4 waves. In the 3x3 reduction loop wave w owns the w-th 16-channel
// block of EVERY tap (9 k-steps) for the whole 64 x 64 tile — four accumulators, 24 MFMAs per fragment set in the six-product form — so
// that no B fragment is fetched twice by a workgroup and a three-deep register ring covers the L2 latency (in a 2 (M) x 2 (N) wave
// layout a k-step is 6 MFMAs = 80 ns against a ~700 ns round trip: measured 22 us per launch whatever the level's size); the four
// partial tiles are summed through LDS in the epilogue. The small 1x1 gate GEMMs keep the 2 x 2 layout with their (pre-split) weights
// in registers. Everything an epilogue needs from memory (bias, Dropout2d mask rows, pivots, BatchNorm input rows, residual rows) is
// requested BEFORE the reduction loop.
// ELU: every activation id the launch uses is LVAE_ACT_ELU (the model's default), known at compile time. With a run-time id each of the
// kernel's ~40 four-value activation calls is a chain of scalar compares and TAKEN branches, and with one wave per SIMD a taken branch
// costs a refetch (~20-30 cycles): the staging segment held 172 branch instructions and ran at ~9 cycles per executed instruction.
// AP (PRO_GATE_BWD only): the launch's dout does not exist yet — it is the result of the BatchNorm-backward apply that ends the backward
// of the block that ran just before (that block's input IS this block's output), deferred into this prologue: dout = BN'(ap_dh; ap_x) +
// ap_add from the partial sums ap_parts, formed on the fly and stored to ap_out (this block's own final apply reads it as its `add`).
// One launch (7-8.6 us of a dependent chain) per residual block less.
template <int SPLIT, int MI, int PRO, int EPI, bool ELU, bool AP = false>
__global__ __launch_bounds__(256) void rb_conv_kernel(RbArgs a) {
  static_assert(MI == 1 || MI == 2, "32- or 64-pixel tiles (MI 32-row blocks)");
  static_assert(!AP || PRO == LVAE_RB_PRO_GATE_BWD, "the deferred apply feeds the gate-backward prologue");
  constexpr int BM = 32 * MI, LDK = RB_LDK, LDO = RB_LDO, NQ = BM / 16;
  constexpr int OS_BYTES = BM * LDO * 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  kernarg_warmup<sizeof(RbArgs)>();   // ~600 bytes = ten cache lines, read piecemeal by up to nine dependent s_load batches otherwise
  float* scr = reinterpret_cast<float*>(smem_raw);
  unsigned char* mainr = smem_raw + RB_SCR_BYTES;
  // the patch sits behind the dy2 staging tile when the prologue produces it through one (PRO_GATE_BWD)
  __bf16* As = reinterpret_cast<__bf16*>(mainr + (PRO == LVAE_RB_PRO_GATE_BWD ? OS_BYTES : 0));  // [SPLIT][halo_px][LDK]
  const int a_plane = a.halo_px * LDK;

  const lvae_conv_desc& d = a.d;
  const lvae_rb_ext& e = a.e;
  const int in_act = ELU ? LVAE_ACT_ELU : d.in_act, stats_act = ELU ? LVAE_ACT_ELU : d.stats_act;
  const int gate_act = ELU ? LVAE_ACT_ELU : e.act, bwd_act = ELU ? LVAE_ACT_ELU : e.bwd_act, ap_act = ELU ? LVAE_ACT_ELU : e.ap_act;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // the small 1x1 GEMMs: 64-pixel tiles use a 2 (M) x 2 (N) wave layout; 32-pixel tiles have one row block, so the gate backward's
  // dgrad GEMM (N = 64) runs on waves 0 / 1 and the gate forward's (N = 128) gives each wave ONE 32-column tile
  const int wm = MI == 2 ? wave >> 1 : 0, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const bool g1_on = MI == 2 || wave < 2;
  const int bid = blockIdx.x;
  const int n0 = bid * a.NI, HW = a.HW;
  const int nvalid = min(BM, (d.N - n0) * HW);  // pixel rows of this tile that exist
  const size_t row0 = (size_t)n0 * HW;          // global pixel row of tile row 0 (whole images: tile rows are consecutive pixels)
  const int c4 = (t & 15) * 4, p0 = t >> 4;     // staging / store map: thread -> rows p0 + 16 q, channels c4 .. c4 + 3
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const f32x4 one4 = {1.f, 1.f, 1.f, 1.f};

  // ---- 3x3 weights: this wave's B fragments of tap `tap` (k block = wave): both 32-channel halves, one 16-byte load per plane and half
  constexpr int RING = SPLIT == 1 ? 9 : 3;   // bf16 operands: the whole slice (72 registers); six-product form: three taps ahead
  bf16x8 bq[RING][2][SPLIT];
  const __bf16* wp_lane = a.Wp + (size_t)wave * SPLIT * 1024 + (size_t)lane * 8;
  auto load_b = [&](int tap, bf16x8 (&r)[2][SPLIT]) {
    const __bf16* src = wp_lane + (size_t)tap * 4 * SPLIT * 1024;
#pragma unroll
    for (int p = 0; p < SPLIT; ++p) {
      r[0][p] = *reinterpret_cast<const bf16x8*>(src + p * 1024);
      r[1][p] = *reinterpret_cast<const bf16x8*>(src + p * 1024 + 512);
    }
  };
  RB_STAMP_DECL;
#ifdef LVAE_RB_DBG_REPS
  // instrument only: the whole body twice in one launch, stamps of the LAST pass (is a phase slower the first time its code is fetched?)
  const int rb_reps = a.flip >= 0 ? LVAE_RB_DBG_REPS : 1;
#pragma unroll 1
  for (int rb_rep = 0; rb_rep < rb_reps; ++rb_rep) {
  rb_bar();
#endif
  RB_STAMP(0);
  // ---- L2 warm-up for the NEXT launch. Every convolution of a step has weights of its own, so the 221 KB a workgroup streams are cold in
  // its XCD's L2 when its kernel starts (in-step the launches measured ~4 us longer than back to back on one layer): the workgroups of an
  // XCD (round-robin placement: speed only) touch one word of every 128-byte line of the ranges the next launch will stream.
  // The touches are plain (volatile) loads whose values are consumed by an empty asm at the END of the prologue (pf_drain): the compiler
  // tracks them like any load, so their registers cannot be handed to another value while a touch is in flight. (Round 4 issued them from
  // inline asm with "=v" outputs that the compiler believed written at once; under the register pressure of a new variant of this kernel
  // hipcc split such a live range, the late-landing load overwrote the register's next owner — an address — and the step died with a
  // memory access fault: profiles/r05_faults/. The memory counter retires in order, so by the time the prologue has consumed its own
  // operand rows — requested BEHIND the touches — the touches have landed and the drain costs no wait.) Summing the values into a dummy
  // right away, the first form of round 4, made hipcc wait for each touch in turn (6.8 us of serialized HBM round trips).
  // Issued behind rb_parts_issue where a prologue has partial rows to reduce: those are L2-resident and needed first.
  constexpr int PFN = 6;   // touches per lane and range: 6 x 256 lanes x 128 B = 192 KB of a range per workgroup at most
  unsigned pf_dump[2][PFN];
  auto pf_issue = [&]() {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int per_xcd = (a.nwg + 7) >> 3, part = bid >> 3;
      const int lines = e.pf_ptr[k] != nullptr ? (int)((e.pf_bytes[k] + 127) >> 7) : 0, per = (lines + per_xcd - 1) / per_xcd;
      const int lo = part * per, hi = min(lo + per, lines);
      const char* base = static_cast<const char*>(e.pf_ptr[k]) + (size_t)(lo + t) * 128;   // ranges are far below 2 GB: 32-bit line arithmetic
#pragma unroll
      for (int i = 0; i < PFN; ++i) {
        pf_dump[k][i] = 0;
        if (lo + t + 256 * i < hi) pf_dump[k][i] = *reinterpret_cast<const unsigned*>(base + i * 256 * 128);
      }
    }
  };
  auto pf_drain = [&]() {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int i = 0; i < PFN; ++i) asm volatile("" ::"v"(pf_dump[k][i]));
  };

  // ---- this thread's four rows: patch position, image, global row (rows past the batch read row 0 and are masked)
  int hp[NQ], img_n[NQ];
  size_t grow[NQ];
  bool ok[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int p = p0 + 16 * q;
    const int img = fastdiv(p, a.m_hw), r = p - img * HW;
    const int y = fastdiv(r, a.m_w), x = r - y * d.W;
    hp[q] = ((img * a.halo_h + y + 1) * a.halo_w + x + 1) * LDK + c4;
    ok[q] = p < nvalid;
    img_n[q] = ok[q] ? n0 + img : n0;
    grow[q] = row0 + (ok[q] ? p : 0);
  }
  auto put_interior = [&](int q, f32x4 v) {
    bf16x4 pl[SPLIT];
    rb_split4<SPLIT>(v, pl);
#pragma unroll
    for (int k = 0; k < SPLIT; ++k) *reinterpret_cast<bf16x4*>(As + k * a_plane + hp[q]) = pl[k];
  };
  auto zero_ring = [&]() {
    const int per_img = a.halo_h * a.halo_w;
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int idx = t; idx < a.halo_px * 8; idx += 256) {
      const int px = idx >> 3, c8 = (idx & 7) * 8;
      const int img = fastdiv(px, a.m_per_img), r = px - img * per_img;
      const int hy = fastdiv(r, a.m_halo_w), hx = r - hy * a.halo_w;
      if (hy == 0 || hy == a.halo_h - 1 || hx == 0 || hx == a.halo_w - 1) {
#pragma unroll
        for (int k = 0; k < SPLIT; ++k) *reinterpret_cast<bf16x8*>(As + k * a_plane + px * LDK + c8) = z;
      }
    }
  };

  // =====================================================================================================================
  // prologue: the (transformed) input tile -> patch planes in LDS
  // =====================================================================================================================
  if (PRO == LVAE_RB_PRO_AFFINE) {
    // ---- BatchNorm coefficients of the input: given, or finalized here from the producer's partial sums (conv3x3_pos.hip's fold)
    const lvae_bn_fold& f = a.f;
    const bool has_tf = f.parts != nullptr || d.in_scale != nullptr;
    // the producer's partial rows (small, L2-resident) are requested FIRST, the raw rows of the interior (cold, from HBM) behind them: the
    // finalize then runs while the interior is still in flight
    f32x4 pv[RB_PF];
    if (f.parts != nullptr) rb_parts_issue(f.parts, f.rows, pv);
    pf_issue();
    const bool lead = t < 64 && f.parts != nullptr;
    const float pivot = lead ? f.parts[((size_t)f.rows * 2) * 64 + t] : 0.f;  // the producer's pivot, stored behind its partial rows
    const float gam = lead && f.gamma ? f.gamma[t] : 1.f, bet = lead && f.beta ? f.beta[t] : 0.f;
    const bool upd = lead && bid == 0 && f.running_mean != nullptr;
    const float rm0 = upd ? f.running_mean[t] : 0.f, rv0 = upd ? f.running_var[t] : 0.f;
    f32x4 xv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) xv[q] = *reinterpret_cast<const f32x4*>(d.x + grow[q] * 64 + c4);
    zero_ring();   // while the loads are in flight (the patch region is free from the start in this prologue)
    f32x4 sc = one4, sh = zero4;
    if (f.parts != nullptr) {
      rb_parts_finish(f.parts, f.rows, pv, scr, [&](int c, double sa, double sb) {
        const double M = (double)f.M, inv_m = 1.0 / M, dm = sa * inv_m;
        double m2 = sb - sa * dm;
        if (m2 < 0.0) m2 = 0.0;
        const double mean = (double)pivot + dm, var = m2 * inv_m;
        const float rstd = (float)(1.0 / sqrt(var + (double)f.eps));
        const float scl = gam * rstd, shf = bet - (float)mean * scl;
        scr[1024 + c] = scl;
        scr[1024 + 64 + c] = shf;
        if (bid == 0) {
          if (f.coef_out) {
            f.coef_out[c] = scl;
            f.coef_out[64 + c] = shf;
            f.coef_out[128 + c] = (float)mean;
            f.coef_out[192 + c] = rstd;
          }
          if (upd) {
            const double unbiased = f.M > 1 ? m2 / (M - 1.0) : var;
            f.running_mean[c] = (1.f - f.momentum) * rm0 + f.momentum * (float)mean;
            f.running_var[c] = (1.f - f.momentum) * rv0 + f.momentum * (float)unbiased;
          }
        }
      });
      sc = *reinterpret_cast<const f32x4*>(scr + 1024 + c4);
      sh = *reinterpret_cast<const f32x4*>(scr + 1024 + 64 + c4);
    } else if (d.in_scale != nullptr) {
      sc = *reinterpret_cast<const f32x4*>(d.in_scale + c4);
      sh = *reinterpret_cast<const f32x4*>(d.in_shift + c4);
    }
    RB_STAMP(1);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      f32x4 v = xv[q];
      if (has_tf) v = act_fwd4(v * sc + sh, in_act);
      put_interior(q, ok[q] ? v : zero4);
    }
    RB_STAMP(10);
    RB_STAMP(11);
  }

  if (PRO == LVAE_RB_PRO_BN_APPLY) {
    // ---- BatchNorm backward of the block's second (or first) BatchNorm on the way in: d.x = dh (gradient w.r.t. act(BN(x))),
    // e.bwd_x = x; the reduction over the batch was done by the producer's epilogue (partial rows), finalized here by every
    // workgroup in the same order (bitwise identical coefficients); workgroup 0 accumulates dgamma / dbeta
    // small L2-resident things first (partial rows, coefficients, accumulators), the cold operand rows behind them (in-order memory counter)
    f32x4 pv[RB_PF];
    rb_parts_issue(e.bwd_parts, e.bwd_rows, pv);
    pf_issue();
    const f32x4 sc = *reinterpret_cast<const f32x4*>(e.bwd_coef + c4), sh = *reinterpret_cast<const f32x4*>(e.bwd_coef + 64 + c4);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(e.bwd_coef + 128 + c4), rs = *reinterpret_cast<const f32x4*>(e.bwd_coef + 192 + c4);
    const bool acc_here = bid == 0 && t < 64;
    const float db0 = acc_here && e.dbeta ? e.dbeta[t] : 0.f, dg0 = acc_here && e.dgamma ? e.dgamma[t] : 0.f;
    f32x4 gv[NQ], xv[NQ], dm[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      gv[q] = *reinterpret_cast<const f32x4*>(d.x + grow[q] * 64 + c4);
      xv[q] = *reinterpret_cast<const f32x4*>(e.bwd_x + grow[q] * 64 + c4);
      dm[q] = e.pro_drop ? *reinterpret_cast<const f32x4*>(e.pro_drop + (size_t)img_n[q] * 64 + c4) : one4;
    }
    zero_ring();   // while the loads are in flight
    rb_parts_finish(e.bwd_parts, e.bwd_rows, pv, scr, [&](int c, double sa, double sb) {
      scr[1024 + c] = (float)(sa / (double)e.bwd_M);
      scr[1024 + 64 + c] = (float)(sb / (double)e.bwd_M);
      if (bid == 0) {
        if (e.dbeta) e.dbeta[c] = db0 + (float)sa;
        if (e.dgamma) e.dgamma[c] = dg0 + (float)sb;
      }
    }
#ifdef LVAE_RB_DBG
    , rb_ts
#endif
    );
    const f32x4 c1 = *reinterpret_cast<const f32x4*>(scr + 1024 + c4), c2 = *reinterpret_cast<const f32x4*>(scr + 1024 + 64 + c4);
    RB_STAMP(1);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const f32x4 g = gv[q] * rb_act_grad4(xv[q] * sc + sh, bwd_act);
      f32x4 v = (g - c1 - (xv[q] - mu) * rs * c2) * sc * dm[q];
      if (!ok[q]) v = zero4;
      if (ok[q] && e.xt_out) store_wt4(e.xt_out + grow[q] * 64 + c4, v);
      put_interior(q, v);
    }
  }

  if (PRO == LVAE_RB_PRO_GATE_BWD) {
    // ---- GateLayer2d backward on the way in: dab from (dout, ab) -> [SPLIT][BM][128] planes -> dy2 = (dab . Wg^T) * Dropout2d mask
    __bf16* Ds = reinterpret_cast<__bf16*>(mainr);
    constexpr int LDD = RB_LDD, d_plane = BM * LDD;
    f32x4 go[NQ], aa[NQ], bb[NQ], dm[NQ];
    if (AP) {
      // small L2-resident things first (partial rows, coefficients, accumulators), the cold operand rows behind them (in-order memory counter)
      f32x4 pv[RB_PF];
      rb_parts_issue(e.ap_parts, e.ap_rows, pv);
      pf_issue();
      const f32x4 sc = *reinterpret_cast<const f32x4*>(e.ap_coef + c4), sh = *reinterpret_cast<const f32x4*>(e.ap_coef + 64 + c4);
      const f32x4 mu = *reinterpret_cast<const f32x4*>(e.ap_coef + 128 + c4), rs = *reinterpret_cast<const f32x4*>(e.ap_coef + 192 + c4);
      const bool acc_here = bid == 0 && t < 64;
      const float db0 = acc_here && e.ap_dbeta ? e.ap_dbeta[t] : 0.f, dg0 = acc_here && e.ap_dgamma ? e.ap_dgamma[t] : 0.f;
      f32x4 xa[NQ], ad[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        go[q] = *reinterpret_cast<const f32x4*>(e.ap_dh + grow[q] * 64 + c4);
        xa[q] = *reinterpret_cast<const f32x4*>(e.ap_x + grow[q] * 64 + c4);
        ad[q] = *reinterpret_cast<const f32x4*>(e.ap_add + grow[q] * 64 + c4);   // (required: a conditional load costs a full s_waitcnt in the emitted code)
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        aa[q] = *reinterpret_cast<const f32x4*>(e.ab_in + grow[q] * 128 + c4);
        bb[q] = *reinterpret_cast<const f32x4*>(e.ab_in + grow[q] * 128 + 64 + c4);
      }
      rb_parts_finish(e.ap_parts, e.ap_rows, pv, scr, [&](int c, double sa, double sb) {
        scr[1024 + c] = (float)(sa / (double)e.ap_M);
        scr[1024 + 64 + c] = (float)(sb / (double)e.ap_M);
        if (bid == 0) {
          if (e.ap_dbeta) e.ap_dbeta[c] = db0 + (float)sa;
          if (e.ap_dgamma) e.ap_dgamma[c] = dg0 + (float)sb;
        }
      });
      const f32x4 c1 = *reinterpret_cast<const f32x4*>(scr + 1024 + c4), c2 = *reinterpret_cast<const f32x4*>(scr + 1024 + 64 + c4);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const f32x4 g = go[q] * rb_act_grad4(xa[q] * sc + sh, ap_act);
        const f32x4 v = (g - c1 - (xa[q] - mu) * rs * c2) * sc + ad[q];
        go[q] = v;
        if (ok[q]) store_wt4(e.ap_out + grow[q] * 64 + c4, v);
      }
    } else {
      pf_issue();
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        go[q] = *reinterpret_cast<const f32x4*>(e.dout + grow[q] * 64 + c4);
        aa[q] = *reinterpret_cast<const f32x4*>(e.ab_in + grow[q] * 128 + c4);
        bb[q] = *reinterpret_cast<const f32x4*>(e.ab_in + grow[q] * 128 + 64 + c4);
      }
    }
    // gate weights in the transposed use (pre-split planes [k-step 8][column tile 2][plane][lane][8]): this wave's 32 output columns
    bf16x8 gq[8][SPLIT];
    {
      const __bf16* gws = static_cast<const __bf16*>(e.gate_ws) + (size_t)wn * SPLIT * 512 + lane * 8;
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int k = 0; k < SPLIT; ++k) gq[s][k] = *reinterpret_cast<const bf16x8*>(gws + ((size_t)s * 2 * SPLIT + k) * 512);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) dm[q] = e.pro_drop ? *reinterpret_cast<const f32x4*>(e.pro_drop + (size_t)img_n[q] * 64 + c4) : one4;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int p = p0 + 16 * q;
      f32x4 da, db;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float s = sigmoidf_(bb[q][j]);
        const float gs = go[q][j] * s;
        da[j] = gs;
        db[j] = gs * (1.f - s);
      }
      da = da * rb_act_grad4(aa[q], gate_act);
      db = db * act_fwd4(aa[q], gate_act);
      if (!ok[q]) da = db = zero4;
      if (ok[q] && e.dab) {
        store_wt4(e.dab + grow[q] * 128 + c4, da);
        store_wt4(e.dab + grow[q] * 128 + 64 + c4, db);
      }
      bf16x4 pa[SPLIT], pb[SPLIT];
      rb_split4<SPLIT>(da, pa);
      rb_split4<SPLIT>(db, pb);
#pragma unroll
      for (int k = 0; k < SPLIT; ++k) {
        *reinterpret_cast<bf16x4*>(Ds + k * d_plane + p * LDD + c4) = pa[k];
        *reinterpret_cast<bf16x4*>(Ds + k * d_plane + p * LDD + 64 + c4) = pb[k];
      }
    }
    RB_STAMP(8);
    rb_bar();
    f32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
    if (g1_on) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        bf16x8 af[SPLIT];
#pragma unroll
        for (int k = 0; k < SPLIT; ++k) af[k] = *reinterpret_cast<const bf16x8*>(Ds + k * d_plane + (wm * 32 + li) * LDD + 16 * s + 8 * lh);
        acc1 = mfma_pieces<SPLIT>(af, gq[s], acc1);
      }
    }
    RB_STAMP(9);
    rb_bar();  // Ds is dead: its head becomes the dy2 staging tile
    float* Os1 = reinterpret_cast<float*>(mainr);
    if (g1_on) {
#pragma unroll
      for (int r = 0; r < 16; ++r) Os1[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDO + wn * 32 + li] = acc1[r];
    }
    rb_bar();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int p = p0 + 16 * q;
      f32x4 v = *reinterpret_cast<const f32x4*>(Os1 + p * LDO + c4) * dm[q];
      if (!ok[q]) v = zero4;
      if (ok[q] && e.xt_out) store_wt4(e.xt_out + grow[q] * 64 + c4, v);
      put_interior(q, v);
    }
    zero_ring();
  }

  pf_drain();   // the prologue has consumed operand rows requested behind the touches: they have landed (in-order counter), their registers are free
  // ---- everything the epilogue reads from memory is requested now; it arrives while the matrix cores work
  f32x4 ep_bias = zero4, ep_mask[NQ], ep_piv = zero4, ep_bsh = zero4, ep_bmu = zero4, ep_brs = zero4, ep_sx[NQ], ep_res[NQ], ep_ga = zero4, ep_gb = zero4;
  const bool plain_stats = EPI == LVAE_RB_EPI_PLAIN && d.stats_out != nullptr;
  const bool stats_bwd = plain_stats && d.stats_mode == LVAE_STATS_BN_BWD;
  {
    if (d.bias) ep_bias = *reinterpret_cast<const f32x4*>(d.bias + c4);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      ep_mask[q] = d.out_scale ? *reinterpret_cast<const f32x4*>(d.out_scale + (size_t)img_n[q] * 64 + c4) : one4;
      ep_sx[q] = zero4;
      ep_res[q] = zero4;
    }
    if (plain_stats) {
      ep_piv = *reinterpret_cast<const f32x4*>(d.stats_pivot + c4);
      if (stats_bwd) {
        ep_bsh = *reinterpret_cast<const f32x4*>(d.stats_pivot + 64 + c4);
        ep_bmu = *reinterpret_cast<const f32x4*>(d.stats_pivot + 128 + c4);
        ep_brs = *reinterpret_cast<const f32x4*>(d.stats_pivot + 192 + c4);
#pragma unroll
        for (int q = 0; q < NQ; ++q) ep_sx[q] = *reinterpret_cast<const f32x4*>(d.stats_x + grow[q] * 64 + c4);
      }
    }
    if (EPI == LVAE_RB_EPI_GATE) {
      if (e.out_stats) ep_piv = *reinterpret_cast<const f32x4*>(e.out_stats_pivot + c4);
      if (e.gate_bias) {
        ep_ga = *reinterpret_cast<const f32x4*>(e.gate_bias + c4);
        ep_gb = *reinterpret_cast<const f32x4*>(e.gate_bias + 64 + c4);
      }
      if (e.res) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) ep_res[q] = *reinterpret_cast<const f32x4*>(e.res + grow[q] * 64 + c4);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < RING; ++i) load_b(i, bq[i]);

  // ---- per-lane patch row of its A-fragment pixels (element offset inside a plane; tap (0, 0) = the pixel's upper-left neighbour);
  // this wave's 16-channel block of the reduction dimension
  int hbase[MI];
#pragma unroll
  for (int mb = 0; mb < MI; ++mb) {
    const int p = mb * 32 + li;
    const int img = fastdiv(p, a.m_hw), r = p - img * HW;
    const int ty = fastdiv(r, a.m_w), tx = r - ty * d.W;
    hbase[mb] = ((img * a.halo_h + ty) * a.halo_w + tx) * LDK + wave * 16 + 8 * lh;
  }
  f32x16 acc[MI][2];   // [row block][channel half]
#pragma unroll
  for (int mb = 0; mb < MI; ++mb)
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nh][r] = 0.f;

  RB_STAMP(2);
  rb_bar();
  RB_STAMP(3);

  // =====================================================================================================================
  // 9 k-steps per wave (its channel block of every tap), no barrier: B RING taps ahead from L2, A one tap ahead from LDS
  // =====================================================================================================================
  {
    bf16x8 af[2][MI][SPLIT];
    auto load_a = [&](int tap, bf16x8 (&fa)[MI][SPLIT]) {
      const int kh = tap / 3, kw = tap - kh * 3;
      const int dh = a.flip ? 2 - kh : kh, dw = a.flip ? 2 - kw : kw;
      const int off = (dh * a.halo_w + dw) * LDK;
#pragma unroll
      for (int p = 0; p < SPLIT; ++p)
#pragma unroll
        for (int mb = 0; mb < MI; ++mb) fa[mb][p] = *reinterpret_cast<const bf16x8*>(As + p * a_plane + hbase[mb] + off);
    };
    load_a(0, af[0]);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int cur = tap & 1;
      if (tap + 1 < 9) load_a(tap + 1, af[cur ^ 1]);
#pragma unroll
      for (int mb = 0; mb < MI; ++mb)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) acc[mb][nh] = mfma_pieces<SPLIT>(af[cur][mb], bq[tap % RING][nh], acc[mb][nh]);
      if (tap + RING < 9) load_b(tap + RING, bq[tap % RING]);
    }
  }

  RB_STAMP(4);
  // gate forward: this wave's pre-split gate weights (planes [k-step 4][column tile 4][plane][lane][8]; a-half tile wn, b-half tile 2 + wn)
  bf16x8 gqa[EPI == LVAE_RB_EPI_GATE ? 4 : 1][SPLIT], gqb[EPI == LVAE_RB_EPI_GATE ? 4 : 1][SPLIT];
  if (EPI == LVAE_RB_EPI_GATE) {
    const __bf16* gws = static_cast<const __bf16*>(e.gate_ws) + lane * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int k = 0; k < SPLIT; ++k) {
        // 64-pixel tiles: a-half tile wn and b-half tile 2 + wn; 32-pixel tiles: the ONE column tile `wave` (0, 1: a halves; 2, 3: b halves)
        gqa[s][k] = *reinterpret_cast<const bf16x8*>(gws + ((size_t)(s * 4 + (MI == 2 ? wn : wave)) * SPLIT + k) * 512);
        if (MI == 2) gqb[s][k] = *reinterpret_cast<const bf16x8*>(gws + ((size_t)(s * 4 + 2 + wn) * SPLIT + k) * 512);
        else gqb[s][k] = gqa[s][k];
      }
  }

  rb_bar();  // every wave is done with the patch: the region becomes the output staging tiles

  // =====================================================================================================================
  // epilogue: the four waves' partial tiles [wave][64][68] -> summed by the reader -> 16-byte row stores
  // =====================================================================================================================
  float* Os = reinterpret_cast<float*>(mainr);
  constexpr int OSW = BM * LDO;
#pragma unroll
  for (int mb = 0; mb < MI; ++mb)
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Os[wave * OSW + (mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDO + nh * 32 + li] = acc[mb][nh][r];
  rb_bar();
  RB_STAMP(5);
  auto os_sum = [&](int p) {
    const float* o = Os + p * LDO + c4;
    return (*reinterpret_cast<const f32x4*>(o) + *reinterpret_cast<const f32x4*>(o + OSW)) +
           (*reinterpret_cast<const f32x4*>(o + 2 * OSW) + *reinterpret_cast<const f32x4*>(o + 3 * OSW));
  };

  f32x4 st1 = zero4, st2 = zero4;
  const float* stats_pivot = nullptr;
  float* stats_out = nullptr;
  bool stats_fwd = true;

  if (EPI == LVAE_RB_EPI_PLAIN) {
    stats_out = d.stats_out;
    stats_pivot = d.stats_pivot;
    stats_fwd = !stats_bwd;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int p = p0 + 16 * q;
      const f32x4 v = (os_sum(p) + ep_bias) * ep_mask[q];
      if (ok[q]) {
        store_wt4(d.y + grow[q] * 64 + c4, v);
        if (stats_out) {
          if (stats_fwd) {
            const f32x4 dl = v - ep_piv;
            st1 += dl;
            st2 += dl * dl;
          } else {
            const f32x4 g = v * rb_act_grad4(ep_sx[q] * ep_piv + ep_bsh, stats_act);
            st1 += g;
            st2 += g * (ep_sx[q] - ep_bmu) * ep_brs;
          }
        }
      }
    }
  }

  if (EPI == LVAE_RB_EPI_GATE) {
    // ---- y2 = (conv + bias) * Dropout2d mask: stored for the backward, and split into the A planes of the gate GEMM
    __bf16* Gs = reinterpret_cast<__bf16*>(mainr + 4 * OS_BYTES);  // [SPLIT][BM][LDK], behind the four partial tiles
    constexpr int g_plane = BM * LDK;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int p = p0 + 16 * q;
      f32x4 v = (os_sum(p) + ep_bias) * ep_mask[q];
      if (!ok[q]) v = zero4;
      if (ok[q]) store_wt4(d.y + grow[q] * 64 + c4, v);
      bf16x4 pl[SPLIT];
      rb_split4<SPLIT>(v, pl);
#pragma unroll
      for (int k = 0; k < SPLIT; ++k) *reinterpret_cast<bf16x4*>(Gs + k * g_plane + p * LDK + c4) = pl[k];
    }
    rb_bar();
    f32x16 acca, accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) acca[r] = accb[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 af[SPLIT];
#pragma unroll
      for (int k = 0; k < SPLIT; ++k) af[k] = *reinterpret_cast<const bf16x8*>(Gs + k * g_plane + (wm * 32 + li) * LDK + 16 * s + 8 * lh);
      acca = mfma_pieces<SPLIT>(af, gqa[s], acca);
      if (MI == 2) accb = mfma_pieces<SPLIT>(af, gqb[s], accb);
    }
    rb_bar();  // the partial tiles and Gs are dead: the region becomes the pre-activation tile [BM][132]
    constexpr int LDG = RB_LDG;
    float* Qs = reinterpret_cast<float*>(mainr);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (MI == 2) {
        Qs[row * LDG + wn * 32 + li] = acca[r];
        Qs[row * LDG + 64 + wn * 32 + li] = accb[r];
      } else {
        Qs[row * LDG + wave * 32 + li] = acca[r];
      }
    }
    rb_bar();
    stats_out = e.out_stats;
    stats_pivot = e.out_stats_pivot;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int p = p0 + 16 * q;
      if (ok[q]) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(Qs + p * LDG + c4) + ep_ga;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(Qs + p * LDG + 64 + c4) + ep_gb;
        if (e.ab) {
          store_wt4(e.ab + grow[q] * 128 + c4, av);
          store_wt4(e.ab + grow[q] * 128 + 64 + c4, bv);
        }
        f32x4 o = act_fwd4(av, gate_act);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] *= sigmoidf_(bv[j]);
        o += ep_res[q];
        store_wt4(e.out + grow[q] * 64 + c4, o);
        if (stats_out) {
          const f32x4 dl = o - ep_piv;
          st1 += dl;
          st2 += dl * dl;
        }
      }
    }
  }

  RB_STAMP(6);
  if (stats_out) {  // 16 row groups x 64 channels -> one row of partials per workgroup (fixed order); the pivot travels behind the rows
    rb_bar();
    float* red = reinterpret_cast<float*>(mainr);
    *reinterpret_cast<f32x4*>(red + (t >> 4) * 64 + c4) = st1;
    *reinterpret_cast<f32x4*>(red + 1024 + (t >> 4) * 64 + c4) = st2;
    rb_bar();
    if (t < 128) {
      const int c = t & 63, which = t >> 6;
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) v += red[which * 1024 + r * 64 + c];
      stats_out[((size_t)bid * 2 + which) * 64 + c] = v;
      if (bid == 0 && which == 0 && stats_fwd) stats_out[((size_t)a.nwg * 2) * 64 + c] = stats_pivot[c];
    }
  }
  RB_STAMP(7);
#ifdef LVAE_RB_DBG_REPS
  }
#endif
  RB_STAMP_FLUSH;
}

// ---------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------
static bool al16r(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

size_t conv3x3_bf16_workspace(const lvae_conv_desc* d, int split);
void conv3x3_bf16_prep_entry(const lvae_conv_desc* d, int split, void* entry);
int conv3x3_bf16_prepare_single(const lvae_conv_desc* d, int split, hipStream_t s);
size_t resblock_gate_ws_bytes(const lvae_conv_desc* d, int planes);
void resblock_gate_prep_entry(const lvae_conv_desc* d, int planes, void* entry);
int resblock_gate_prepare_single(const lvae_conv_desc* d, int planes, hipStream_t s);
int conv3x3_wino2_gate_rows(const lvae_conv_desc* d);
int conv3x3_wino2_gate_try(const lvae_conv_desc* d, const lvae_rb_ext* gate, hipStream_t s);

// the 1x1 gate convolution in the direction it is used: 64 -> 128 (forward) or 128 -> 64 (backward)
static bool rb_gate_ok(const lvae_conv_desc* g) {
  return g != nullptr && g->w != nullptr && ((g->C1 == 64 && g->Cout == 128) || (g->C1 == 128 && g->Cout == 64));
}

static int rb_split(const lvae_conv_desc* d) { return d->precision == LVAE_PREC_BF16 ? 1 : 3; }

// 32-row blocks per wave: 1 (64-pixel tiles) unless the level has so many pixels that 64-pixel tiles would need more than one round of
// workgroups (tuning builds can force either)
static int rb_mi(const lvae_conv_desc* d) {
  // 64-pixel tiles (two 32-row blocks), or — round 5 — 32-pixel tiles where 64-pixel tiles would leave at least half of the 256 CUs
  // without a workgroup (the 4x4 and 2x2 levels at batch 256: 64 / 16 workgroups): a workgroup's reduction loop is MFMA-bound
  // (216 MFMAs per wave at one wave per SIMD), so half the tile on twice the CUs is half the loop
  static const int force = (int)tune("LVAE_RB_MI", 0);   // A/B switch (tuning builds only): 1 | 2
  if (force == 1 || force == 2) return force;
  const int HW = d->H * d->W;
  if (HW <= 16 && 32 % HW == 0 && ((int64_t)d->N * HW + 63) / 64 <= 128) return 1;
  return 2;
}

static size_t rb_lds_bytes(int split, int mi, int halo_px, int pro, int epi) {
  const size_t BM = 32 * mi, patch = (size_t)split * halo_px * RB_LDK * 2, os = BM * RB_LDO * 4;
  size_t m = patch > 4 * os ? patch : 4 * os;   // the epilogue's four partial tiles alias the patch
  if (pro == LVAE_RB_PRO_GATE_BWD) {
    const size_t ds = (size_t)split * BM * RB_LDD * 2;
    if (os + patch > m) m = os + patch;
    if (ds > m) m = ds;
  }
  if (epi == LVAE_RB_EPI_GATE) {
    const size_t g = 4 * os + (size_t)split * BM * RB_LDK * 2, q = BM * RB_LDG * 4;
    if (g > m) m = g;
    if (q > m) m = q;
  }
  if (m < 2048 * 4) m = 2048 * 4;  // statistics reduction
  return RB_SCR_BYTES + m;
}

static bool rb_plan(const lvae_conv_desc* d, RbArgs& a, int& mi) {
  static const bool off = tune("LVAE_DISABLE_RB", 0) != 0;  // A/B switch (tuning builds only)
  if (off || d == nullptr) return false;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->x2 != nullptr || d->OH != d->H || d->OW != d->W) return false;
  if (d->C1 != 64 || d->C2 != 0 || d->Cout != 64 || d->out_act != LVAE_ACT_NONE) return false;
  const int HW = d->H * d->W;
  if (HW > 64 || 64 % HW != 0 || d->N < 1) return false;
  if (d->x_dtype != LVAE_DT_F32 || d->y_dtype != LVAE_DT_F32 || d->stats_x_dtype != LVAE_DT_F32) return false;
  if ((int64_t)d->N * HW * 128 >= ((int64_t)1 << 31)) return false;
  mi = rb_mi(d);
  const int BM = 32 * mi;
  a.d = *d;
  a.d.in_fold = nullptr;
  a.f = lvae_bn_fold{};
  a.HW = HW;
  a.NI = BM / HW;
  a.halo_h = d->H + 2;
  a.halo_w = d->W + 2;
  a.halo_px = a.NI * a.halo_h * a.halo_w;
  a.flip = d->gather == LVAE_GATHER_TRANSPOSED ? 1 : 0;
  a.nwg = (d->N + a.NI - 1) / a.NI;
  a.m_hw = fastdiv_magic(HW);
  a.m_w = fastdiv_magic(d->W);
  a.m_per_img = fastdiv_magic(a.halo_h * a.halo_w);
  a.m_halo_w = fastdiv_magic(a.halo_w);
  return rb_lds_bytes(rb_split(d), mi, a.halo_px, LVAE_RB_PRO_GATE_BWD, LVAE_RB_EPI_GATE) <= 160 * 1024;
}

template <int SPLIT, int MI, int PRO, int EPI, bool ELU, bool AP = false>
static int rb_launch(const RbArgs& a, hipStream_t s) {
  auto kern = rb_conv_kernel<SPLIT, MI, PRO, EPI, ELU, AP>;
  static std::atomic<bool> attr_set{false};  // idempotent attribute write; the flag itself is race-free
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      set_error("resblock_conv: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.nwg), dim3(256), rb_lds_bytes(SPLIT, MI, a.halo_px, PRO, EPI), s, a);
  LVAE_LAUNCH_CHECK("resblock_conv");
  return 0;
}

// every activation id this launch will look at is ELU (ids of features the launch does not use are ignored)
static bool rb_all_elu(const RbArgs& a) {
  const lvae_conv_desc& d = a.d;
  const lvae_rb_ext& e = a.e;
  const auto elu = [](int act) { return act == LVAE_ACT_ELU; };
  if (e.prologue == LVAE_RB_PRO_AFFINE && (d.in_scale != nullptr || a.f.parts != nullptr) && !elu(d.in_act)) return false;
  if (e.prologue == LVAE_RB_PRO_BN_APPLY && !elu(e.bwd_act)) return false;
  if (e.prologue == LVAE_RB_PRO_GATE_BWD && e.ap_parts != nullptr && !elu(e.ap_act)) return false;
  if ((e.prologue == LVAE_RB_PRO_GATE_BWD || e.epilogue == LVAE_RB_EPI_GATE) && !elu(e.act)) return false;
  if (d.stats_out != nullptr && d.stats_mode == LVAE_STATS_BN_BWD && !elu(d.stats_act)) return false;
  return true;
}

template <int SPLIT, int MI, bool ELU>
static int rb_dispatch2(const RbArgs& a, hipStream_t s) {
  const int pro = a.e.prologue, epi = a.e.epilogue;
  if (pro == LVAE_RB_PRO_AFFINE) return epi == LVAE_RB_EPI_GATE ? rb_launch<SPLIT, MI, LVAE_RB_PRO_AFFINE, LVAE_RB_EPI_GATE, ELU>(a, s)
                                                                  : rb_launch<SPLIT, MI, LVAE_RB_PRO_AFFINE, LVAE_RB_EPI_PLAIN, ELU>(a, s);
  if (pro == LVAE_RB_PRO_BN_APPLY) return rb_launch<SPLIT, MI, LVAE_RB_PRO_BN_APPLY, LVAE_RB_EPI_PLAIN, ELU>(a, s);
  if (a.e.ap_parts != nullptr) return rb_launch<SPLIT, MI, LVAE_RB_PRO_GATE_BWD, LVAE_RB_EPI_PLAIN, ELU, true>(a, s);
  return rb_launch<SPLIT, MI, LVAE_RB_PRO_GATE_BWD, LVAE_RB_EPI_PLAIN, ELU>(a, s);
}

template <int SPLIT, int MI>
static int rb_dispatch(const RbArgs& a, hipStream_t s) {
  return rb_all_elu(a) ? rb_dispatch2<SPLIT, MI, true>(a, s) : rb_dispatch2<SPLIT, MI, false>(a, s);
}

}  // namespace lvae

using namespace lvae;

extern "C" size_t lvae_resblock_gate_workspace(const lvae_conv_desc* g) { return rb_gate_ok(g) ? resblock_gate_ws_bytes(g, rb_split(g)) : 0; }

extern "C" int lvae_resblock_gate_prepare_entry(const lvae_conv_desc* g, void* entry) {
  LVAE_REQUIRE(rb_gate_ok(g) && entry, LVAE_EINVAL, "lvae_resblock_gate_prepare_entry: needs a 64 -> 128 or 128 -> 64 1x1 descriptor and an entry buffer");
  LVAE_REQUIRE(g->workspace && (size_t)g->workspace_bytes >= resblock_gate_ws_bytes(g, rb_split(g)) && al16r(g->workspace), LVAE_EINVAL,
               "lvae_resblock_gate_prepare_entry: no scratch for the pre-split gate weights");
  resblock_gate_prep_entry(g, rb_split(g), entry);
  return 0;
}

// rows of out_stats (= workgroups) of an LVAE_RB_EPI_GATE launch for d: the whole-image kernels' rows, or — fp32, 64 -> 64 channels, the
// 256-pixel six-product Winograd kernel's shapes (16x16 and 32x32 levels at batch 256) with ITS workspace attached — that kernel's rows
extern "C" int32_t lvae_resblock_conv_gate_rows(const lvae_conv_desc* d) {
  RbArgs a;
  int mi;
  if (rb_plan(d, a, mi)) return a.nwg;
  return d != nullptr && d->precision == LVAE_PREC_F32 ? conv3x3_wino2_gate_rows(d) : 0;
}

extern "C" int32_t lvae_resblock_conv_rows(const lvae_conv_desc* d) {
  RbArgs a;
  int mi;
  return rb_plan(d, a, mi) ? a.nwg : 0;
}

extern "C" size_t lvae_resblock_conv_workspace(const lvae_conv_desc* d) {
  RbArgs a;
  int mi;
  return rb_plan(d, a, mi) ? conv3x3_bf16_workspace(d, rb_split(d)) : 0;
}

extern "C" int lvae_resblock_conv_prepare_entry(const lvae_conv_desc* d, void* entry) {
  LVAE_REQUIRE(d && entry, LVAE_EINVAL, "lvae_resblock_conv_prepare_entry: null pointer");
  RbArgs a;
  int mi;
  LVAE_REQUIRE(rb_plan(d, a, mi), LVAE_EINVAL, "lvae_resblock_conv_prepare_entry: shape not supported (lvae_resblock_conv_rows(d) == 0)");
  LVAE_REQUIRE(d->workspace && (size_t)d->workspace_bytes >= conv3x3_bf16_workspace(d, rb_split(d)) && al16r(d->workspace), LVAE_EINVAL,
               "lvae_resblock_conv_prepare_entry: no scratch for the pre-split weights");
  conv3x3_bf16_prep_entry(d, rb_split(d), entry);
  return 0;
}

extern "C" int lvae_resblock_conv_f32(const lvae_conv_desc* d, const lvae_rb_ext* ext, void* stream) {
  LVAE_REQUIRE(d != nullptr && d->w != nullptr && d->y != nullptr, LVAE_EINVAL, "lvae_resblock_conv_f32: null descriptor / w / y");
  RbArgs a;
  int mi = 1;
  if (!rb_plan(d, a, mi)) {
    // larger levels: only conv + gate (forward), through the 256-pixel six-product Winograd kernel with the gate behind it
    LVAE_REQUIRE(ext != nullptr && ext->prologue == LVAE_RB_PRO_AFFINE && ext->epilogue == LVAE_RB_EPI_GATE && d->precision == LVAE_PREC_F32 &&
                     conv3x3_wino2_gate_rows(d) > 0,
                 LVAE_EINVAL,
                 "lvae_resblock_conv_f32: shape not supported (whole-image kernels: 3x3 / stride 1 / pad 1, 64 -> 64 channels, H*W a divisor of 64, "
                 "fp32 tensors; conv + gate also for lvae_resblock_conv_gate_rows(d) > 0)");
    const lvae_rb_ext& e = *ext;
    LVAE_REQUIRE(e.out && al16r(e.out) && al16r(e.ab) && al16r(e.res) && al16r(e.gate_bias) && al16r(e.out_stats) && al16r(e.out_stats_pivot) &&
                     (e.out_stats == nullptr || e.out_stats_pivot != nullptr) && d->stats_out == nullptr && d->x != nullptr && d->y != nullptr,
                 LVAE_EINVAL, "lvae_resblock_conv_f32: gate epilogue needs x, y, out (16-byte aligned tensors); statistics of y are not available with it");
    lvae_conv_desc g = lvae_conv_desc{};
    g.w = e.gate_w; g.w_sk = e.gate_w_sk; g.w_sn = e.gate_w_sn; g.precision = d->precision; g.C1 = 64; g.Cout = 128;
    g.workspace = e.gate_ws; g.workspace_bytes = e.gate_ws_bytes;
    LVAE_REQUIRE(e.gate_ws != nullptr && al16r(e.gate_ws) && (size_t)e.gate_ws_bytes >= resblock_gate_ws_bytes(&g, 3) && (e.gate_ws_ready || e.gate_w != nullptr),
                 LVAE_EWORKSPACE, "lvae_resblock_conv_f32: gate_ws missing or smaller than lvae_resblock_gate_workspace(), or not ready and no gate_w");
    if (!e.gate_ws_ready) {
      const int rc = resblock_gate_prepare_single(&g, 3, (hipStream_t)stream);
      if (rc) return rc;
    }
    const int rc = conv3x3_wino2_gate_try(d, ext, (hipStream_t)stream);
    LVAE_REQUIRE(rc != -1000, LVAE_EINVAL, "lvae_resblock_conv_f32: the Winograd kernel did not take the descriptor");
    return rc;
  }
  const int split = rb_split(d);
  a.e = lvae_rb_ext{};
  if (ext != nullptr) a.e = *ext;
  const lvae_rb_ext& e = a.e;
  const int pro = e.prologue, epi = e.epilogue;
  LVAE_REQUIRE(pro >= LVAE_RB_PRO_AFFINE && pro <= LVAE_RB_PRO_GATE_BWD && (epi == LVAE_RB_EPI_PLAIN || epi == LVAE_RB_EPI_GATE), LVAE_EINVAL,
               "lvae_resblock_conv_f32: bad prologue / epilogue id");
  LVAE_REQUIRE(epi == LVAE_RB_EPI_PLAIN || pro == LVAE_RB_PRO_AFFINE, LVAE_EINVAL, "lvae_resblock_conv_f32: the gate epilogue goes with the forward prologue");
  LVAE_REQUIRE(d->workspace != nullptr && (size_t)d->workspace_bytes >= conv3x3_bf16_workspace(d, split) && al16r(d->workspace), LVAE_EWORKSPACE,
               "lvae_resblock_conv_f32: workspace (pre-split weights) missing or smaller than lvae_resblock_conv_workspace(d)");
  LVAE_REQUIRE(al16r(d->y) && al16r(d->bias) && al16r(d->out_scale) && al16r(d->in_scale) && al16r(d->in_shift) && al16r(d->stats_pivot) &&
                   al16r(d->stats_x) && al16r(d->stats_out),
               LVAE_EALIGN, "lvae_resblock_conv_f32: pointers must be 16-byte aligned");
  LVAE_REQUIRE((d->in_scale == nullptr) || (d->in_shift != nullptr), LVAE_EINVAL, "lvae_resblock_conv_f32: in_scale without in_shift");
  LVAE_REQUIRE(d->stats_out == nullptr || (d->stats_pivot != nullptr && (d->stats_mode == LVAE_STATS_BN_FWD ||
                                                                           (d->stats_mode == LVAE_STATS_BN_BWD && d->stats_x != nullptr))),
               LVAE_EINVAL, "lvae_resblock_conv_f32: bad stats_pivot / stats_mode / stats_x");
  if (pro == LVAE_RB_PRO_AFFINE) {
    LVAE_REQUIRE(d->x != nullptr && al16r(d->x), LVAE_EINVAL, "lvae_resblock_conv_f32: null or unaligned x");
    if (d->in_fold != nullptr) {
      a.f = *d->in_fold;
      LVAE_REQUIRE(a.f.parts != nullptr && al16r(a.f.parts) && a.f.rows > 0 && a.f.M > 0 && d->in_scale == nullptr, LVAE_EINVAL,
                   "lvae_resblock_conv_f32: bad in_fold (parts / rows / M, or in_scale given too)");
    }
  } else {
    LVAE_REQUIRE(d->in_fold == nullptr && d->in_scale == nullptr, LVAE_EINVAL, "lvae_resblock_conv_f32: input transform with a backward prologue");
    LVAE_REQUIRE(al16r(e.pro_drop) && al16r(e.xt_out), LVAE_EALIGN, "lvae_resblock_conv_f32: pro_drop / xt_out must be 16-byte aligned");
  }
  if (pro == LVAE_RB_PRO_BN_APPLY) {
    LVAE_REQUIRE(d->x && e.bwd_x && e.bwd_parts && e.bwd_coef && e.bwd_rows > 0 && e.bwd_M > 0 && al16r(d->x) && al16r(e.bwd_x) &&
                     al16r(e.bwd_parts) && al16r(e.bwd_coef),
                 LVAE_EINVAL, "lvae_resblock_conv_f32: BatchNorm-apply prologue needs x (= dh), bwd_x, bwd_parts / rows / M and the coefficient block, 16-byte aligned");
  }
  if (pro == LVAE_RB_PRO_GATE_BWD) {
    LVAE_REQUIRE((e.dout || e.ap_parts) && e.ab_in && al16r(e.dout) && al16r(e.ab_in) && al16r(e.dab), LVAE_EINVAL,
                 "lvae_resblock_conv_f32: gate-backward prologue needs dout (or the deferred apply that produces it) and ab_in (16-byte aligned tensors)");
    if (e.ap_parts != nullptr) {
      LVAE_REQUIRE(e.ap_rows > 0 && e.ap_M > 0 && e.ap_coef && e.ap_dh && e.ap_x && e.ap_add && e.ap_out && al16r(e.ap_parts) && al16r(e.ap_coef) &&
                       al16r(e.ap_dh) && al16r(e.ap_x) && al16r(e.ap_add) && al16r(e.ap_out),
                   LVAE_EINVAL, "lvae_resblock_conv_f32: deferred apply needs ap_parts / ap_rows / ap_M, the coefficient block, ap_dh, ap_x, ap_add and ap_out, 16-byte aligned");
    }
  } else {
    LVAE_REQUIRE(e.ap_parts == nullptr, LVAE_EINVAL, "lvae_resblock_conv_f32: the deferred apply goes with the gate-backward prologue");
  }
  if (pro == LVAE_RB_PRO_GATE_BWD || epi == LVAE_RB_EPI_GATE) {
    lvae_conv_desc g = lvae_conv_desc{};   // the gate convolution in the direction this launch uses it
    g.w = e.gate_w; g.w_sk = e.gate_w_sk; g.w_sn = e.gate_w_sn; g.precision = d->precision;
    g.C1 = epi == LVAE_RB_EPI_GATE ? 64 : 128; g.Cout = epi == LVAE_RB_EPI_GATE ? 128 : 64;
    g.workspace = e.gate_ws; g.workspace_bytes = e.gate_ws_bytes;
    LVAE_REQUIRE(e.gate_ws != nullptr && al16r(e.gate_ws) && (size_t)e.gate_ws_bytes >= resblock_gate_ws_bytes(&g, split) &&
                     (e.gate_ws_ready || e.gate_w != nullptr),
                 LVAE_EWORKSPACE, "lvae_resblock_conv_f32: gate_ws missing or smaller than lvae_resblock_gate_workspace(), or not ready and no gate_w");
    if (!e.gate_ws_ready) {
      const int rc = resblock_gate_prepare_single(&g, split, (hipStream_t)stream);
      if (rc) return rc;
    }
  }
  if (epi == LVAE_RB_EPI_GATE) {
    LVAE_REQUIRE(e.out && al16r(e.out) && al16r(e.ab) && al16r(e.res) && al16r(e.gate_bias) && al16r(e.out_stats) &&
                     al16r(e.out_stats_pivot) && (e.out_stats == nullptr || e.out_stats_pivot != nullptr) && d->stats_out == nullptr,
                 LVAE_EINVAL, "lvae_resblock_conv_f32: gate epilogue needs out (16-byte aligned tensors); statistics of y are not available with it");
  }
  if (!d->workspace_ready) {
    const int rc = conv3x3_bf16_prepare_single(d, split, (hipStream_t)stream);
    if (rc) return rc;
  }
  a.Wp = static_cast<const __bf16*>(d->workspace);
  hipStream_t s = (hipStream_t)stream;
  if (mi == 1) return split == 1 ? rb_dispatch<1, 1>(a, s) : rb_dispatch<3, 1>(a, s);
  return split == 1 ? rb_dispatch<1, 2>(a, s) : rb_dispatch<3, 2>(a, s);
}

#ifdef LVAE_RB_DBG
extern "C" int lvae_debug_rb_stamps(void* host_out, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lvae::g_rb_stamps), bytes < sizeof(lvae::g_rb_stamps) ? bytes : sizeof(lvae::g_rb_stamps));
}
#endif
