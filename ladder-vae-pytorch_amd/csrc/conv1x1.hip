// Pointwise (1x1, stride 1) convolution for K = Cin <= 128, N = Cout <= 128: GateLayer2d's 64->128 conv, MergeLayer's
// 128->64 conv (channel concat read from two tensors) and their dgrads.
//
// With K <= 128 there is nothing to pipeline over: the generic K-staged kernel spends its time in 2-4 dependent
// load -> barrier -> MFMA round trips (measured 32 us for a layer whose MFMA floor is 7 us). Here a workgroup stages its
// whole A tile (BM pixels x K) and the whole weight matrix in ONE batch of loads, runs all MFMAs, and writes the tile
// through LDS with 16-byte row stores. Optional fused epilogue for the gate (lib/nn.py:121-126 + the residual add of
// lib/nn.py:99): with N = 2C, out = act(y[:, :C]) * sigmoid(y[:, C:]) + res, written next to (or instead of) y.
//
// 4 waves as 2(M) x 2(N); wave wn owns columns {ni*64 + wn*32 + lane} (ni = 0,1), so for the gate the a- and b-halves of
// a channel sit in the same lane.
#include <stdlib.h>

#include "lvae_common.h"

namespace lvae {

struct PwArgs {
  lvae_conv_desc d;
  int M, K, ohw;
  const float* gate_res;  // [M][N/2] or null
  float* gate_out;        // [M][N/2] or null (null: plain convolution)
  int gate_act;
  // gate backward fused as the A operand (K = 2C): A[m][k] = dab[m][k] computed from dout [M][C] and ab [M][2C]; also written
  // to gb_dab (the weight gradient of the gate convolution reads it)
  const float* gb_dout;
  const float* gb_ab;
  float* gb_dab;
  int gb_act;
  // two output tensors (plain epilogue only): columns [0, split) go to d.y with row pitch `split`, columns [split, N) to y2 with row pitch
  // N - split — the dgrad of a convolution whose input was a channel concat of two tensors (MergeLayer, models/lvae_layers.py:347-359) in
  // ONE launch instead of one per half (round 5). y2 == nullptr: one tensor, row pitch N.
  float* y2;
  int split;
};

template <int BM, int KT, int NT, bool B_KCONTIG>
__global__ __launch_bounds__(256) void conv1x1_kernel(PwArgs a) {
  kernarg_warmup<(sizeof(PwArgs) < 1024 ? sizeof(PwArgs) : 1024)>();
  constexpr int LDA = KT + 4;
  constexpr int K4 = KT / 4;
  constexpr int WMT = BM / 2, MI = WMT / 32, NI = NT / 64;
  constexpr int LDO = NT + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;             // [BM][LDA]
  float* Bs = smem + BM * LDA;  // k-contig: [NT][LDA] ; n-contig: [KT][NT]
  const lvae_conv_desc& d = a.d;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * BM;
  const int K = a.K, N = d.Cout;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- one batch: A tile (two-pointer concat) and the weight matrix; clamped addresses, no exec-mask regions
  constexpr int APT = BM * K4 / 256;  // float4 of A per thread
  constexpr int BPT = NT * K4 / 256;  // float4 of B per thread (both layouts hold KT*NT floats)
  f32x4 av[APT], bv[BPT];
  const bool gate_bwd = a.gb_dout != nullptr;
  // gate forward: the residual rows this thread adds in the epilogue, fetched NOW (the epilogue would otherwise pay a second HBM
  // round trip at the end of every workgroup's dependent chain); same (row, channel) map as the epilogue loop
  constexpr int RPT = NT >= 128 ? BM * (NT / 8) / 256 : 1;  // float4 of the residual per thread (C = NT / 2 channels)
  f32x4 resv[RPT];
  const bool res_pre = a.gate_out != nullptr && a.gate_res != nullptr && NT >= 128 && N == NT;
  if (res_pre) {
    const int c4n = N >> 3;
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
      const int idx = t + 256 * u, r = idx / c4n, c = (idx - r * c4n) * 4;
      const int m = m0 + r;
      resv[u] = *reinterpret_cast<const f32x4*>(a.gate_res + (size_t)(m < a.M ? m : 0) * (N >> 1) + c);
    }
  }
  if (!gate_bwd) {
#pragma unroll
    for (int u = 0; u < APT; ++u) {
      const int idx = t + 256 * u, r = idx / K4, k = (idx - r * K4) * 4;
      const int m = m0 + r;
      const bool ok = (m < a.M) & (k < K);
      const bool first = k < d.C1;
      const float* src = first ? d.x : d.x2;
      const size_t off = ok ? (size_t)m * (first ? d.C1 : d.C2) + (first ? k : k - d.C1) : 0;
      const f32x4 v = *reinterpret_cast<const f32x4*>((ok ? src : d.x) + off);
      av[u] = ok ? v : zero4;
    }
  }
#pragma unroll
  for (int u = 0; u < BPT; ++u) {
    const int idx = t + 256 * u;
    if (B_KCONTIG) {
      const int n = idx / K4, k = (idx - n * K4) * 4;
      const bool ok = (n < N) & (k < K);
      const f32x4 v = *reinterpret_cast<const f32x4*>(d.w + (ok ? (int64_t)n * d.w_sn + k : 0));
      bv[u] = ok ? v : zero4;
    } else {
      const int k = idx / (NT / 4), n = (idx - k * (NT / 4)) * 4;
      const bool ok = (k < K) & (n < N);
      const f32x4 v = *reinterpret_cast<const f32x4*>(d.w + (ok ? (int64_t)k * d.w_sk + n : 0));
      bv[u] = ok ? v : zero4;
    }
  }
  if (!gate_bwd) {
#pragma unroll
    for (int u = 0; u < APT; ++u) {
      const int idx = t + 256 * u, r = idx / K4, k = (idx - r * K4) * 4;
      f32x4 v = av[u];
      if (d.in_scale && m0 + r < a.M && k < K) {
        v = act_fwd4(v * *reinterpret_cast<const f32x4*>(d.in_scale + k) + *reinterpret_cast<const f32x4*>(d.in_shift + k), d.in_act);
      }
      *reinterpret_cast<f32x4*>(As + r * LDA + k) = v;
    }
  } else {
    // dab[:, c] = dout * sigmoid(b) * act'(a) ; dab[:, C + c] = dout * act(a) * sigmoid(b) * (1 - sigmoid(b))   (lib/nn.py:121-126)
    const int C = K >> 1;
    for (int idx = t; idx < BM * (KT / 8); idx += 256) {
      const int r = idx / (KT / 8), c = (idx - r * (KT / 8)) * 4;
      const int m = m0 + r;
      if (c >= C) continue;
      f32x4 lo = zero4, hi = zero4;
      if (m < a.M) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(a.gb_dout + (size_t)m * C + c);
        const f32x4 av4 = *reinterpret_cast<const f32x4*>(a.gb_ab + (size_t)m * K + c);
        const f32x4 bv4 = *reinterpret_cast<const f32x4*>(a.gb_ab + (size_t)m * K + C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float sg = sigmoidf_(bv4[j]);
          lo[j] = g[j] * sg * act_grad(av4[j], a.gb_act);
          hi[j] = g[j] * act_fwd(av4[j], a.gb_act) * sg * (1.f - sg);
        }
        if (a.gb_dab) {
          *reinterpret_cast<f32x4*>(a.gb_dab + (size_t)m * K + c) = lo;
          *reinterpret_cast<f32x4*>(a.gb_dab + (size_t)m * K + C + c) = hi;
        }
      }
      *reinterpret_cast<f32x4*>(As + r * LDA + c) = lo;
      *reinterpret_cast<f32x4*>(As + r * LDA + C + c) = hi;
    }
    // columns [2C, KT) of the tile (K < KT) must read as zero
    if (K < KT)
      for (int idx = t; idx < BM * ((KT - K) / 4); idx += 256) {
        const int r = idx / ((KT - K) / 4), k = K + (idx - r * ((KT - K) / 4)) * 4;
        *reinterpret_cast<f32x4*>(As + r * LDA + k) = zero4;
      }
  }
#pragma unroll
  for (int u = 0; u < BPT; ++u) {
    const int idx = t + 256 * u;
    if (B_KCONTIG) {
      const int n = idx / K4, k = (idx - n * K4) * 4;
      *reinterpret_cast<f32x4*>(Bs + n * LDA + k) = bv[u];
    } else {
      const int k = idx / (NT / 4), n = (idx - k * (NT / 4)) * 4;
      *reinterpret_cast<f32x4*>(Bs + k * NT + n) = bv[u];
    }
  }
  __syncthreads();

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

#pragma unroll
  for (int kk = 0; kk < KT; kk += 8) {
    f32x4 af[MI], bf[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const f32x4*>(As + (wm * WMT + mi * 32 + li) * LDA + kk + 4 * lh);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = ni * 64 + wn * 32 + li;
      if (B_KCONTIG) {
        bf[ni] = *reinterpret_cast<const f32x4*>(Bs + col * LDA + kk + 4 * lh);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[ni][j] = Bs[(kk + 4 * lh + j) * NT + col];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
  }
  __syncthreads();  // operands are dead: reuse LDS as the output staging tile [BM][LDO]

  float* Os = smem;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Os[(wm * WMT + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDO + ni * 64 + wn * 32 + li] = acc[mi][ni][r];
  __syncthreads();

  if (a.gate_out != nullptr) {
    // gate epilogue: C = N/2 channels; thread -> (pixel, 4 channels)
    const int C = N >> 1, c4n = C >> 2;  // float4 per output pixel
    // BatchNorm partials of `out` (d.stats_out): 256 % c4n == 0, so a thread keeps the same channel group for all its rows
    const bool stats = d.stats_out != nullptr && (256 % c4n) == 0;
    f32x4 st1 = zero4, st2 = zero4, piv = zero4;
    if (stats) piv = *reinterpret_cast<const f32x4*>(d.stats_pivot + (t % c4n) * 4);
    int it = 0;
    for (int idx = t; idx < BM * c4n; idx += 256, ++it) {
      const int r = idx / c4n, c = (idx - r * c4n) * 4;
      const int m = m0 + r;
      if (m >= a.M) continue;
      f32x4 va = *reinterpret_cast<const f32x4*>(Os + r * LDO + c);
      f32x4 vb = *reinterpret_cast<const f32x4*>(Os + r * LDO + C + c);
      if (d.bias) {
        va += *reinterpret_cast<const f32x4*>(d.bias + c);
        vb += *reinterpret_cast<const f32x4*>(d.bias + C + c);
      }
      if (d.y) {
        store_wt4(d.y + (size_t)m * N + c, va);
        store_wt4(d.y + (size_t)m * N + C + c, vb);
      }
      f32x4 o = act_fwd4(va, a.gate_act);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] *= sigmoidf_(vb[j]);
      if (res_pre) {
        f32x4 rv = resv[0];
#pragma unroll
        for (int u = 1; u < RPT; ++u)
          if (it == u) rv = resv[u];
        o += rv;
      } else if (a.gate_res) {
        o += *reinterpret_cast<const f32x4*>(a.gate_res + (size_t)m * C + c);
      }
      store_wt4(a.gate_out + (size_t)m * C + c, o);
      const f32x4 dl = o - piv;
      st1 += dl;
      st2 += dl * dl;
    }
    if (stats) {  // 256 / c4n row groups x C channels -> one row of partials per workgroup (fixed order)
      __syncthreads();  // the staging tile is dead
      const int G = 256 / c4n;
      float* red = smem;
      *reinterpret_cast<f32x4*>(red + (t / c4n) * C + (t % c4n) * 4) = st1;
      *reinterpret_cast<f32x4*>(red + G * C + (t / c4n) * C + (t % c4n) * 4) = st2;
      __syncthreads();
      if (t < 2 * C) {
        const int c = t % C, which = t / C;
        float v = 0.f;
        for (int r = 0; r < G; ++r) v += red[which * G * C + r * C + c];
        d.stats_out[((size_t)blockIdx.x * 2 + which) * C + c] = v;
        // the pivot travels with the partials (row gridDim.x): a consumer that finalizes them in its own prologue (lvae_bn_fold) must
        // not depend on a buffer it updates itself
        if (blockIdx.x == 0 && which == 0) d.stats_out[((size_t)gridDim.x * 2) * C + c] = d.stats_pivot[c];
      }
    }
  } else {
    const int n4 = N >> 2;
    for (int idx = t; idx < BM * n4; idx += 256) {
      const int r = idx / n4, c = (idx - r * n4) * 4;
      const int m = m0 + r;
      if (m >= a.M) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(Os + r * LDO + c);
      if (d.bias) v += *reinterpret_cast<const f32x4*>(d.bias + c);
      if (d.out_scale) v = v * *reinterpret_cast<const f32x4*>(d.out_scale + (size_t)(m / a.ohw) * N + c);
      v = act_fwd4(v, d.out_act);
      if (a.y2 == nullptr) store_wt4(d.y + (size_t)m * N + c, v);
      else if (c < a.split) store_wt4(d.y + (size_t)m * a.split + c, v);
      else store_wt4(a.y2 + (size_t)m * (N - a.split) + (c - a.split), v);
    }
  }
}

static bool al16p(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int BM, int KT, int NT, bool KC>
static int launch_pw(const PwArgs& a, hipStream_t s) {
  auto kern = conv1x1_kernel<BM, KT, NT, KC>;
  constexpr size_t lds_in = (size_t)(BM * (KT + 4) + (KC ? NT * (KT + 4) : KT * NT)) * sizeof(float);
  constexpr size_t lds_out = (size_t)BM * (NT + 4) * sizeof(float);
  constexpr size_t lds = lds_in > lds_out ? lds_in : lds_out;
  static std::atomic<bool> attr_set{false};  // idempotent attribute write; the flag itself is race-free
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("conv1x1: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((a.M + BM - 1) / BM), dim3(256), lds, s, a);
  LVAE_LAUNCH_CHECK("conv1x1");
  return 0;
}

template <int KT, int NT, bool KC>
static int pick_bm(const PwArgs& a, hipStream_t s) {
  static const int force = (int)tune("LVAE_PW_BM", 0);
  if (force == 64) return launch_pw<64, KT, NT, KC>(a, s);
  if (force == 128) return launch_pw<128, KT, NT, KC>(a, s);
  return launch_pw<64, KT, NT, KC>(a, s);  // measured: 64-pixel tiles (2-3 workgroups per CU overlap load / MFMA / store) beat 128
}

// -1000: not eligible (the caller uses the generic kernel)
static int conv1x1_try_all(const lvae_conv_desc* d, const float* gate_res, float* gate_out, int gate_act, const float* gb_dout,
                           const float* gb_ab, float* gb_dab, int gb_act, hipStream_t s, float* y2, int split) {
  const int K = d->C1 + d->C2, N = d->Cout;
  if (d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0 || d->OH != d->H || d->OW != d->W) return -1000;
  if (K > 128 || N > 128 || d->C1 % 4 || d->C2 % 4 || N % 4 || (gate_out && N % 8)) return -1000;
  if (gb_dout && (K % 8 || d->C2 != 0 || d->in_scale != nullptr || !al16p(gb_dout) || !al16p(gb_ab) || !al16p(gb_dab))) return -1000;
  if (!al16p(d->x) || !al16p(d->x2) || !al16p(d->w) || !al16p(d->y) || !al16p(d->bias) || !al16p(d->in_scale) ||
      !al16p(d->in_shift) || !al16p(d->out_scale) || !al16p(gate_res) || !al16p(gate_out))
    return -1000;
  const bool kc = d->w_sk == 1 && d->w_sn % 4 == 0 && K % 4 == 0;
  const bool nc = d->w_sn == 1 && d->w_sk % 4 == 0;
  if (!kc && !nc) return -1000;
  if (gate_out == nullptr && d->y == nullptr) return -1000;
  PwArgs a;
  a.d = *d;
  a.M = d->N * d->H * d->W;
  a.K = K;
  a.ohw = d->H * d->W;
  a.gate_res = gate_res;
  a.gate_out = gate_out;
  a.gate_act = gate_act;
  a.gb_dout = gb_dout;
  a.gb_ab = gb_ab;
  a.gb_dab = gb_dab;
  a.gb_act = gb_act;
  a.y2 = y2;
  a.split = split;
  if (y2 != nullptr && (gate_out != nullptr || d->stats_out != nullptr || split <= 0 || split >= N || split % 4 != 0 || !al16p(y2))) return -1000;
  const bool k64 = K <= 64, n64 = N <= 64;
  if (nc) {
    if (k64) return n64 ? pick_bm<64, 64, false>(a, s) : pick_bm<64, 128, false>(a, s);
    return n64 ? pick_bm<128, 64, false>(a, s) : pick_bm<128, 128, false>(a, s);
  }
  if (k64) return n64 ? pick_bm<64, 64, true>(a, s) : pick_bm<64, 128, true>(a, s);
  return n64 ? pick_bm<128, 64, true>(a, s) : pick_bm<128, 128, true>(a, s);
}

int conv1x1_try_ex(const lvae_conv_desc* d, const float* gate_res, float* gate_out, int gate_act, const float* gb_dout,
                   const float* gb_ab, float* gb_dab, int gb_act, hipStream_t s) {
  return conv1x1_try_all(d, gate_res, gate_out, gate_act, gb_dout, gb_ab, gb_dab, gb_act, s, nullptr, 0);
}

int conv1x1_try(const lvae_conv_desc* d, const float* gate_res, float* gate_out, int gate_act, hipStream_t s) {
  return conv1x1_try_all(d, gate_res, gate_out, gate_act, nullptr, nullptr, nullptr, 0, s, nullptr, 0);
}

}  // namespace lvae

namespace lvae {
int conv_desc_check(const lvae_conv_desc* d, const char* who);
}
using namespace lvae;

extern "C" int lvae_conv1x1_dgrad_cat_f32(const lvae_conv_desc* d, float* dx2, int32_t split, void* stream) {
  int rc = conv_desc_check(d, "lvae_conv1x1_dgrad_cat_f32");
  if (rc) return rc;
  LVAE_REQUIRE(d->y != nullptr && dx2 != nullptr && split > 0 && split < d->Cout && split % 4 == 0 && d->stats_out == nullptr && d->x2 == nullptr,
               LVAE_EINVAL, "lvae_conv1x1_dgrad_cat_f32: needs y, dx2, 0 < split < Cout (a multiple of 4), no statistics epilogue, one input tensor");
  LVAE_REQUIRE(d->x_dtype == LVAE_DT_F32 && d->y_dtype == LVAE_DT_F32, LVAE_EINVAL, "lvae_conv1x1_dgrad_cat_f32: fp32 tensors only");
  rc = conv1x1_try_all(d, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, (hipStream_t)stream, dx2, split);
  LVAE_REQUIRE(rc != -1000, LVAE_EINVAL,
               "lvae_conv1x1_dgrad_cat_f32: shape not supported (1x1, stride 1, at most 128 reduction and 128 output channels, multiples of 4, "
               "16-byte aligned tensors, unit weight stride along one axis): use two lvae_conv2d_f32 launches");
  return rc;
}
