// Weight gradient of the 1x1 gate convolutions (64 -> 64/128 channels, no input transform): a plain GEMM
//   dW[ci][co] = sum_pixels x[pixel][ci] * dy[pixel][co]
// whose operands are already in MFMA layout in memory: lane (li, lh) of v_mfma_f32_32x32x2_f32 needs A[ci = li][k = lh]
// = x[pixel + lh][ci] and B[k = lh][co = li] = dy[pixel + lh][co], i.e. 32 consecutive floats of an NHWC row per lane half.
// So there is no LDS staging at all: every wave streams its own pixel range with 128-byte row segments straight from
// global memory (2 + Cout/32 loads per 2*Cout/32 MFMAs), keeps the whole 64 x Cout block in accumulators, and the four
// waves of a workgroup are summed through LDS at the end. The tile kernel this replaces stages both operands through LDS
// with four loader waves per CU and was bound by their memory-level parallelism (36 us for 50 MB at 256x16x16).
// Partials go to the usual split-K slabs [workgroup][ci][co] (+ [workgroup][co] for the bias), reduced in fixed order.
#include <stdlib.h>

#include "lvae_common.h"

namespace lvae {

struct W1x1Args {
  const float* x;    // [M][64]
  const float* dy;   // [M][Cout]
  float* slab_w;     // [nwg][64][Cout]
  float* slab_b;     // [nwg][Cout] or nullptr
  int M, Cout, ppw;  // ppw: pixels per wave (even)
};

template <int NB>  // 32-wide output-channel blocks: Cout = 32 * NB
__global__ __launch_bounds__(256, 1) void conv1x1_wgrad_kernel(W1x1Args a) {
  kernarg_warmup<(sizeof(W1x1Args) < 1024 ? sizeof(W1x1Args) : 1024)>();
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [4 waves][2][NB][16][64] + [4][NB][32]
  constexpr int PER_WAVE = 2 * NB * 16 * 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int p_begin = (blockIdx.x * 4 + wave) * a.ppw;
  const int p_end = min(a.M, p_begin + a.ppw);
  const int Cout = a.Cout;

  f32x16 acc[2][NB];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[h][j][r] = 0.f;
  float bsum[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) bsum[j] = 0.f;

  const float* xp = a.x + li;
  const float* yp = a.dy + li;
  // batches of 4 k-steps (8 pixels), fetched two batches (64 MFMAs) ahead through a ring of three register sets: with one
  // wave per SIMD nothing else hides the HBM round trip
  struct Batch {
    float xa[4][2];
    float b[4][NB];
  };
  auto load_batch = [&](Batch& q, int p) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int px = p + 2 * u + lh;
      const bool live = px < p_end;
      const int pc = live ? px : p_begin;  // clamped address, value zeroed
      const float m = live ? 1.f : 0.f;
      q.xa[u][0] = m * xp[(size_t)pc * 64];
      q.xa[u][1] = m * xp[(size_t)pc * 64 + 32];
#pragma unroll
      for (int j = 0; j < NB; ++j) q.b[u][j] = m * yp[(size_t)pc * Cout + 32 * j];
    }
  };
  auto mfma_batch = [&](const Batch& q) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        bsum[j] += q.b[u][j];
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(q.xa[u][0], q.b[u][j], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(q.xa[u][1], q.b[u][j], acc[1][j], 0, 0, 0);
      }
  };
  Batch ring[3];
  load_batch(ring[0], p_begin);
  load_batch(ring[1], p_begin + 8);
  for (int p = p_begin; p < p_end; p += 24) {
    load_batch(ring[2], p + 16);
    __builtin_amdgcn_sched_barrier(0);
    mfma_batch(ring[0]);
    __builtin_amdgcn_sched_barrier(0);
    if (p + 8 < p_end) {
      load_batch(ring[0], p + 24);
      __builtin_amdgcn_sched_barrier(0);
      mfma_batch(ring[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (p + 16 < p_end) {
      load_batch(ring[1], p + 32);
      __builtin_amdgcn_sched_barrier(0);
      mfma_batch(ring[2]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- sum the four waves through LDS; every thread then reduces and stores a quarter of the block
  float* mine = smem + wave * PER_WAVE;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[((h * NB + j) * 16 + r) * 64 + lane] = acc[h][j][r];
  float* bs = smem + 4 * PER_WAVE;  // [4][NB][32]
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const float v = bsum[j] + __shfl_xor(bsum[j], 32, 64);
    if (lh == 0) bs[(wave * NB + j) * 32 + li] = v;
  }
  __syncthreads();
  float* slab = a.slab_w + (size_t)blockIdx.x * 64 * Cout;
  for (int e = t; e < PER_WAVE; e += 256) {
    const float v = (smem[e] + smem[PER_WAVE + e]) + (smem[2 * PER_WAVE + e] + smem[3 * PER_WAVE + e]);
    const int ln = e & 63, r = (e >> 6) & 15, hj = e >> 10, j = hj % NB, h = hj / NB;
    const int ci = 32 * h + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), co = 32 * j + (ln & 31);
    slab[(size_t)ci * Cout + co] = v;
  }
  if (a.slab_b && t < NB * 32) {
    const float v = (bs[t] + bs[NB * 32 + t]) + (bs[2 * NB * 32 + t] + bs[3 * NB * 32 + t]);
    a.slab_b[(size_t)blockIdx.x * Cout + t] = v;
  }
}

void wgrad_reduce_launch(const float* slab_w, const float* slab_b, int ksplit, int ntaps, int Cin, int Cout, int64_t stap,
                         int64_t sk, int64_t sn, float* dw, float* db, hipStream_t s);

static bool w1x1_plan(const lvae_conv_desc* d, int& nwg, int& ppw) {
  static const bool off = tune("LVAE_DISABLE_W1X1", 0) != 0;  // A/B switch (tuning builds only)
  if (off) return false;
  if (d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0 || d->gather != LVAE_GATHER_CONV) return false;
  if (d->C1 != 64 || d->C2 != 0 || d->x2 != nullptr || d->in_scale != nullptr) return false;
  if (d->Cout != 64 && d->Cout != 128) return false;
  if (d->OH != d->H || d->OW != d->W) return false;
  const int64_t M = (int64_t)d->N * d->H * d->W;
  if (M < 32768 || M * 128 >= ((int64_t)1 << 31)) return false;  // smaller layers go out in groups through the tile kernel
  nwg = 256;
  ppw = (int)((M + nwg * 4 - 1) / (nwg * 4));
  ppw = (ppw + 1) & ~1;
  nwg = (int)((M + 4 * ppw - 1) / (4 * ppw));
  return true;
}

size_t conv1x1_wgrad_workspace(const lvae_conv_desc* d) {
  int nwg, ppw;
  if (!w1x1_plan(d, nwg, ppw)) return 0;
  return (size_t)nwg * ((size_t)64 * d->Cout + d->Cout) * sizeof(float);
}

// returns -1000 when not eligible
int conv1x1_wgrad_try(const lvae_conv_desc* d, const float* dy, float* dw, float* db, void* workspace, hipStream_t s) {
  int nwg, ppw;
  if (!w1x1_plan(d, nwg, ppw)) return -1000;
  W1x1Args a;
  a.x = d->x;
  a.dy = dy;
  a.M = d->N * d->H * d->W;
  a.Cout = d->Cout;
  a.ppw = ppw;
  a.slab_w = static_cast<float*>(workspace);
  a.slab_b = db ? a.slab_w + (size_t)nwg * 64 * d->Cout : nullptr;
  const int nb = d->Cout / 32;
  const size_t lds = ((size_t)4 * 2 * nb * 16 * 64 + 4 * nb * 32) * sizeof(float);
  static std::atomic<bool> attr_set{false};  // idempotent attribute write; the flag itself is race-free
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_wgrad_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_wgrad_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      set_error("conv1x1_wgrad: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  if (nb == 4) hipLaunchKernelGGL(conv1x1_wgrad_kernel<4>, dim3(nwg), dim3(256), lds, s, a);
  else hipLaunchKernelGGL(conv1x1_wgrad_kernel<2>, dim3(nwg), dim3(256), lds, s, a);
  LVAE_LAUNCH_CHECK("conv1x1_wgrad");
  wgrad_reduce_launch(a.slab_w, a.slab_b, nwg, 1, 64, d->Cout, d->w_stap, d->w_sk, d->w_sn, dw, db, s);
  LVAE_LAUNCH_CHECK("conv2d_wgrad_reduce");
  return 0;
}

}  // namespace lvae
