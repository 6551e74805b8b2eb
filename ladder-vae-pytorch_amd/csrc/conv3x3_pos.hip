// 3x3 / stride 1 / pad 1 convolution (forward and dgrad) for the low-resolution levels (H*W <= 16: the 4x4 and 2x2 levels of
// the ladder, 7 of the 15 stochastic layers of the CIFAR model), "position-major":
//
// the GEMM rows of a workgroup are 32 IMAGES at ONE output position instead of 32 pixels of a few images. Two things follow:
//  * taps that fall outside the image are the same for all 32 rows, so they are skipped instead of multiplied with
//    materialised zeros (at 2x2 only 4 of 9 taps exist for every position, at 4x4 6.25 on average: 2.25x / 1.44x fewer MFMAs
//    than the halo-tile kernel spends);
//  * a level has (N/32) x H*W x (Cout/32) independent tiles — 256 workgroups at 4x4, 64 at 2x2 for N = 256 — where the halo-tile
//    kernel had 128 / 32 workgroups of 3.8 us of fp32 MFMA each (64 FLOP/clk/SIMD is all a CU has).
// These launches are latency bound, so the kernel is single shot: every global load (the <= 9 input rows sets, the <= 9
// weight taps) is issued before the first use, ONE barrier, then the 4 waves split the reduction channels (each 8 MFMAs per
// tap) and are summed through LDS in the epilogue, which is the halo kernel's (bias, Dropout2d scale, activation, BatchNorm
// statistics of the output or BatchNorm-backward sums).
//
// Optional folded BatchNorm finalize (d->in_fold): the BatchNorm coefficients of the INPUT are computed in the prologue from
// the partial sums the producing kernel's epilogue left (lvae_bn_finalize_parts_f32 as a prologue: 64 loads per thread while the
// operand loads are in flight, instead of a 5 us launch in a dependent chain); workgroup 0 publishes (scale, shift, mean, rstd)
// for the backward and updates the running statistics.
#include "lvae_common.h"

namespace lvae {

struct PosArgs {
  lvae_conv_desc d;
  lvae_bn_fold f;  // copy of *d.in_fold (f.parts == nullptr: none)
  int P, Cin, flip, n_groups, ntn;
  uint32_t m_w;  // fastdiv magic of W
};

constexpr int kPosNotEligible = -1000;

template <int CIN_T, bool B_KCONTIG>
__global__ __launch_bounds__(256) void conv3x3_pos_kernel(PosArgs a) {
  kernarg_warmup<(sizeof(PosArgs) < 1024 ? sizeof(PosArgs) : 1024)>();
  constexpr int LDA = CIN_T + 4;             // A row pitch (floats): conflict-free ds_read_b128 (as conv3x3_halo.hip)
  constexpr int CIN4 = CIN_T / 4;            // float4 per A row
  constexpr int RPP = 256 / CIN4;            // A rows per pass of the 256 threads (16 at 64 channels, 32 at 32)
  constexpr int APT = 32 / RPP;              // passes per tap (2 / 1)
  constexpr int ASZ = 32 * LDA;              // floats per A tap slot
  constexpr int BSZ = B_KCONTIG ? 32 * LDA : CIN_T * 32;  // floats per B tap slot ([n][k] padded | [k][n])
  constexpr int BPT = CIN_T * 32 / 4 / 256;  // float4 of a weight tap per thread (2 / 1)
  constexpr int KW_ = CIN_T / 4;             // reduction channels per wave (16 / 8)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float s_fin[4 * 2 * 64 + 2 * 64];  // finalize scratch [4 row groups][2][64] + scale[64] + shift[64]
  const lvae_conv_desc& d = a.d;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;

  const int bid = blockIdx.x;
  const int tile_n = bid % a.ntn, rest = bid / a.ntn;
  const int pos = rest % a.P, ig = rest / a.P;
  const int oy = fastdiv(pos, a.m_w), ox = pos - oy * d.W;
  const int n0 = ig * 32, co0 = tile_n * 32;
  const int Cin = a.Cin;
  const int nrows = min(32, d.N - n0);  // images of this group that exist

  // ---- valid taps (wave uniform): weight tap index `tap`, input position (iy, ix); slot = running index among the valid ones
  // ---- issue ALL global loads: A rows of every valid tap, weight tiles of every valid tap
  f32x4 av[9][APT], bv[9][BPT];
  const int c4 = (t % CIN4) * 4, r0 = t / CIN4;
  const bool c_ok = c4 < Cin;
  unsigned valid = 0;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int kh = tap / 3, kw = tap - kh * 3;
    const int dh = a.flip ? 2 - kh : kh, dw = a.flip ? 2 - kw : kw;
    const int iy = oy + dh - 1, ix = ox + dw - 1;
    if ((unsigned)iy >= (unsigned)d.H || (unsigned)ix >= (unsigned)d.W) continue;  // uniform
    valid |= 1u << tap;
#pragma unroll
    for (int p = 0; p < APT; ++p) {
      const int r = r0 + p * RPP;
      const bool ok = (r < nrows) & c_ok;
      const size_t off = ok ? ((size_t)((n0 + r) * d.H + iy) * d.W + ix) * Cin + c4 : 0;
      av[tap][p] = *reinterpret_cast<const f32x4*>(d.x + off);
    }
    const float* wt = d.w + (int64_t)tap * d.w_stap;
#pragma unroll
    for (int p = 0; p < BPT; ++p) {
      const int idx = t + 256 * p;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (B_KCONTIG) {
        const int n = idx / CIN4, k = (idx - n * CIN4) * 4;
        if (co0 + n < d.Cout && k < Cin) v = *reinterpret_cast<const f32x4*>(wt + (int64_t)(co0 + n) * d.w_sn + k);
      } else {
        const int k = idx >> 3, n = (idx & 7) * 4;
        if (k < Cin && co0 + n < d.Cout) v = *reinterpret_cast<const f32x4*>(wt + (int64_t)k * d.w_sk + co0 + n);
      }
      bv[tap][p] = v;
    }
  }

  // ---- BatchNorm coefficients of the input: given (in_scale / in_shift), or finalized here from the producer's partial sums
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  const lvae_bn_fold& f = a.f;
  const bool has_tf = f.parts != nullptr || d.in_scale != nullptr;
  if (f.parts != nullptr) {
    constexpr int G = 4;  // row groups: thread -> (channel t & 63, rows g, g + 4, ...); C <= 64 on this path
    const int C = Cin;
    const int c = t & 63, g = t >> 6;
    const int rows = f.rows;
    float s1 = 0.f, s2 = 0.f;
    if (c < C)
      for (int r = g; r < rows; r += 8 * G) {  // 16 independent loads per round trip (a rolled loop pays one L2 latency per row)
        float pa[8], pb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int rr = r + u * G;
          const size_t o = ((size_t)(rr < rows ? rr : 0) * 2) * C + c;
          pa[u] = f.parts[o];
          pb[u] = f.parts[o + C];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool ok = r + u * G < rows;
          s1 += ok ? pa[u] : 0.f;
          s2 += ok ? pb[u] : 0.f;
        }
      }
    s_fin[(g * 2) * 64 + c] = s1;
    s_fin[(g * 2 + 1) * 64 + c] = s2;
    __syncthreads();
    if (t < C) {
      double sa = 0.0, sb = 0.0;
      for (int q = 0; q < G; ++q) {
        sa += (double)s_fin[(q * 2) * 64 + t];
        sb += (double)s_fin[(q * 2 + 1) * 64 + t];
      }
      const float pivot = f.parts[((size_t)rows * 2) * C + t];  // the producer's pivot, stored behind its partial rows
      const double M = (double)f.M, inv_m = 1.0 / M, dm = sa * inv_m;
      double m2 = sb - sa * dm;
      if (m2 < 0.0) m2 = 0.0;
      const double mean = (double)pivot + dm, var = m2 * inv_m;
      const float rstd = (float)(1.0 / sqrt(var + (double)f.eps));
      const float gam = f.gamma ? f.gamma[t] : 1.f, bet = f.beta ? f.beta[t] : 0.f;
      const float scl = gam * rstd, shf = bet - (float)mean * scl;
      s_fin[512 + t] = scl;
      s_fin[512 + 64 + t] = shf;
      if (bid == 0) {
        if (f.coef_out) {
          f.coef_out[t] = scl;
          f.coef_out[C + t] = shf;
          f.coef_out[2 * C + t] = (float)mean;
          f.coef_out[3 * C + t] = rstd;
        }
        if (f.running_mean) {
          const double unbiased = f.M > 1 ? m2 / (M - 1.0) : var;
          f.running_mean[t] = (1.f - f.momentum) * f.running_mean[t] + f.momentum * (float)mean;
          f.running_var[t] = (1.f - f.momentum) * f.running_var[t] + f.momentum * (float)unbiased;
        }
      }
    }
    __syncthreads();
    if (c_ok) {
      sc = *reinterpret_cast<const f32x4*>(s_fin + 512 + c4);
      sh = *reinterpret_cast<const f32x4*>(s_fin + 512 + 64 + c4);
    }
  } else if (d.in_scale != nullptr && c_ok) {
    sc = *reinterpret_cast<const f32x4*>(d.in_scale + c4);
    sh = *reinterpret_cast<const f32x4*>(d.in_shift + c4);
  }

  // ---- registers -> LDS (input transform applied once per element; rows of images that do not exist are zero)
  const int ntaps = __popc(valid);
  float* As = smem;
  float* Bs = smem + (size_t)ntaps * ASZ;
  {
    int slot = 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (!((valid >> tap) & 1u)) continue;
#pragma unroll
      for (int p = 0; p < APT; ++p) {
        const int r = r0 + p * RPP;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((r < nrows) & c_ok) {
          v = av[tap][p];
          if (has_tf) v = act_fwd4(v * sc + sh, d.in_act);
        }
        *reinterpret_cast<f32x4*>(As + slot * ASZ + r * LDA + c4) = v;
      }
#pragma unroll
      for (int p = 0; p < BPT; ++p) {
        const int idx = t + 256 * p;
        if (B_KCONTIG) {
          const int n = idx / CIN4, k = (idx - n * CIN4) * 4;
          *reinterpret_cast<f32x4*>(Bs + slot * BSZ + n * LDA + k) = bv[tap][p];
        } else {
          const int k = idx >> 3, n = (idx & 7) * 4;
          *reinterpret_cast<f32x4*>(Bs + slot * BSZ + k * 32 + n) = bv[tap][p];
        }
      }
      ++slot;
    }
  }
  __syncthreads();

  // ---- MFMAs: wave w reduces channels [w*KW_, (w+1)*KW_) of every tap
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int kb = wave * KW_;
  for (int slot = 0; slot < ntaps; ++slot) {
    const float* Ab = As + slot * ASZ + li * LDA + kb + 4 * lh;
    const float* Bb = Bs + slot * BSZ;
#pragma unroll
    for (int q = 0; q < KW_ / 8; ++q) {
      const f32x4 af = *reinterpret_cast<const f32x4*>(Ab + q * 8);
      f32x4 bf;
      if (B_KCONTIG) {
        bf = *reinterpret_cast<const f32x4*>(Bb + li * LDA + kb + q * 8 + 4 * lh);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = Bb[(kb + q * 8 + 4 * lh + j) * 32 + li];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bf[j], acc, 0, 0, 0);
    }
  }
  __syncthreads();  // operands are dead: LDS becomes the 4 partial output tiles [wave][32 rows][36]

  constexpr int LDO = 36;
  float* Os = smem;
#pragma unroll
  for (int r = 0; r < 16; ++r) Os[wave * 32 * LDO + ((r & 3) + 8 * (r >> 2) + 4 * lh) * LDO + li] = acc[r];
  __syncthreads();

  // ---- epilogue: thread -> (row = image, 4 channels); 128-byte row segments
  const int row = t >> 3, oc4 = (t & 7) * 4, col = co0 + oc4;
  const bool live = row < nrows && col < d.Cout;
  f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = st1, piv = st1;
  if (d.stats_out && col < d.Cout) piv = *reinterpret_cast<const f32x4*>(d.stats_pivot + col);
  if (live) {
    f32x4 v = *reinterpret_cast<const f32x4*>(Os + row * LDO + oc4);
#pragma unroll
    for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(Os + w * 32 * LDO + row * LDO + oc4);
    if (d.bias) v += *reinterpret_cast<const f32x4*>(d.bias + col);
    const int n = n0 + row;
    if (d.out_scale) v = v * *reinterpret_cast<const f32x4*>(d.out_scale + (size_t)n * d.Cout + col);
    v = act_fwd4(v, d.out_act);
    const size_t o = ((size_t)(n * d.H + oy) * d.W + ox) * d.Cout + col;
    *reinterpret_cast<f32x4*>(d.y + o) = v;
    if (d.stats_out) {
      if (d.stats_mode == LVAE_STATS_BN_BWD) {  // piv = scale; shift, mean, rstd follow in the [4][Cout] block
        const f32x4 bsh = *reinterpret_cast<const f32x4*>(d.stats_pivot + d.Cout + col);
        const f32x4 bmu = *reinterpret_cast<const f32x4*>(d.stats_pivot + 2 * d.Cout + col);
        const f32x4 brs = *reinterpret_cast<const f32x4*>(d.stats_pivot + 3 * d.Cout + col);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(d.stats_x + o);
        const f32x4 ag = act_grad4(xv * piv + bsh, d.stats_act);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float gj = v[j] * ag[j];
          st1[j] = gj;
          st2[j] = gj * (xv[j] - bmu[j]) * brs[j];
        }
      } else {
        const f32x4 dl = v - piv;
        st1 = dl;
        st2 = dl * dl;
      }
    }
  }
  if (d.stats_out) {  // 32 rows x 32 channels -> one row of partials per (image group, position), summed in a fixed order
    __syncthreads();
    float* red = smem;
    *reinterpret_cast<f32x4*>(red + row * 32 + oc4) = st1;
    *reinterpret_cast<f32x4*>(red + 1024 + row * 32 + oc4) = st2;
    __syncthreads();
    if (t < 64) {
      const int c = t & 31, which = t >> 5;
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 32; ++r) v += red[which * 1024 + r * 32 + c];
      const int tm = ig * a.P + pos;
      if (co0 + c < d.Cout) {
        d.stats_out[((size_t)tm * 2 + which) * d.Cout + co0 + c] = v;
        // the pivot travels with the partials (row index = number of partial rows): a consumer that finalizes them in its own
        // prologue must not read it from a buffer that consumer also updates (the running mean)
        if (tm == 0 && which == 0 && d.stats_mode == LVAE_STATS_BN_FWD)
          d.stats_out[((size_t)a.n_groups * a.P * 2) * d.Cout + co0 + c] = d.stats_pivot[co0 + c];
      }
    }
  }
}

static bool al16q(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// eligibility: 3x3 / stride 1 / pad 1 on images of at most 16 pixels, 32 or 64 reduction channels
static bool pos_select(const lvae_conv_desc* d, bool& ncontig) {
  static const bool off = tune("LVAE_DISABLE_POS", 0) != 0;  // A/B switch (tuning builds only)
  if (off) return false;
  const int Cin = d->C1;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->x2 != nullptr || d->OH != d->H || d->OW != d->W) return false;
  if (d->H * d->W > 16) return false;
  if (Cin > 64 || Cin % 4 != 0 || d->Cout % 4 != 0 || d->w_stap % 4 != 0) return false;
  if (!al16q(d->x) || !al16q(d->w) || !al16q(d->y) || !al16q(d->bias) || !al16q(d->out_scale) || !al16q(d->in_scale) ||
      !al16q(d->in_shift) || !al16q(d->stats_pivot) || !al16q(d->stats_x))
    return false;
  const bool kc = d->w_sk == 1 && d->w_sn % 4 == 0, nc = d->w_sn == 1 && d->w_sk % 4 == 0;
  if (!kc && !nc) return false;
  ncontig = nc;
  return true;
}

bool conv3x3_pos_eligible(const lvae_conv_desc* d) {
  bool nc;
  return pos_select(d, nc);
}

int conv3x3_pos_stats_rows(const lvae_conv_desc* d) {
  bool nc;
  if (!pos_select(d, nc)) return 0;
  return ((d->N + 31) / 32) * d->H * d->W;
}

template <int CIN_T, bool KC>
static int launch_pos(const PosArgs& a, hipStream_t s) {
  auto kern = conv3x3_pos_kernel<CIN_T, KC>;
  static std::atomic<bool> attr_set{false};  // idempotent attribute write; the flag itself is race-free
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 157 * 1024);
    if (e != hipSuccess) {
      set_error("conv3x3_pos: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  // taps a position can have at most: 9 unless the image is narrower / lower than 3
  const int th = a.d.H >= 3 ? 3 : a.d.H, tw = a.d.W >= 3 ? 3 : a.d.W;
  constexpr size_t asz = (size_t)32 * (CIN_T + 4), bsz = KC ? (size_t)32 * (CIN_T + 4) : (size_t)CIN_T * 32;
  size_t lds = (size_t)th * tw * (asz + bsz) * sizeof(float);
  const size_t lds_out = (size_t)4 * 32 * 36 * sizeof(float);
  if (lds < lds_out) lds = lds_out;
  hipLaunchKernelGGL(kern, dim3(a.n_groups * a.P * a.ntn), dim3(256), lds, s, a);
  LVAE_LAUNCH_CHECK("conv3x3_pos");
  return 0;
}

// returns kPosNotEligible when the descriptor does not fit this kernel
int conv3x3_pos_try(const lvae_conv_desc* d, hipStream_t s) {
  bool ncontig = false;
  if (!pos_select(d, ncontig)) return kPosNotEligible;
  PosArgs a;
  a.d = *d;
  a.d.in_fold = nullptr;
  a.f = lvae_bn_fold{};
  if (d->in_fold != nullptr) a.f = *d->in_fold;
  a.P = d->H * d->W;
  a.Cin = d->C1;
  a.flip = d->gather == LVAE_GATHER_TRANSPOSED ? 1 : 0;
  a.n_groups = (d->N + 31) / 32;
  a.ntn = (d->Cout + 31) / 32;
  a.m_w = fastdiv_magic(d->W);
  const bool kc = !ncontig;
  if (d->C1 <= 32) return kc ? launch_pos<32, true>(a, s) : launch_pos<32, false>(a, s);
  return kc ? launch_pos<64, true>(a, s) : launch_pos<64, false>(a, s);
}

}  // namespace lvae
