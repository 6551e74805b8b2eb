// Shared device helpers for liblvae_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include "../../include/lvae_hip.h"

namespace lvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

void set_error(const char* fmt, ...);

// Phase-skip switches of the profiling builds (they make a kernel skip stages, i.e. produce wrong results): compiled out of the
// product library, which reads NO environment variable at all (see tune() below and lvae_conv_desc.form).
inline int debug_phase_switch(const char* name) {
#ifdef LVAE_PHASE_DEBUG
  const char* v = getenv(name);
  return v ? atoi(v) : 0;
#else
  (void)name;
  return 0;
#endif
}

// Thresholds between kernel variants and A/B switches. In the product library they are COMPILE-TIME constants: which kernel runs depends
// only on the descriptor (its `form` field included), never on hidden process state. A -DLVAE_TUNING_ENV build (profiling tools only:
// `make EXTRA=-DLVAE_TUNING_ENV`) reads LVAE_<name> from the environment once, on first use, so that thresholds can be swept without
// rebuilding.
inline int64_t tune(const char* name, int64_t dflt) {
#ifdef LVAE_TUNING_ENV
  const char* v = getenv(name);
  return v ? atoll(v) : dflt;
#else
  (void)name;
  return dflt;
#endif
}

#define LVAE_REQUIRE(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      lvae::set_error(__VA_ARGS__);    \
      return (code);                   \
    }                                  \
  } while (0)

#define LVAE_LAUNCH_CHECK(name)                                        \
  do {                                                                 \
    hipError_t e__ = hipGetLastError();                                \
    if (e__ != hipSuccess) {                                           \
      lvae::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return (int)e__;                                                 \
    }                                                                  \
  } while (0)

// 16-byte store of a large output tensor as a write-through (sc1) store. Plain stores leave the lines dirty in the XCD's L2 until the
// end of the kernel, whose release then writes them back in one go, after the last wave (about bytes / 6 TB/s: ~3 us behind the
// 16.8 MB of a 256x16x16x64 tensor; MI355X_MICROARCH.md, "boundary" and "publish-large"). Written through, the bytes drain
// while the kernel is still computing. The next kernel reads them from memory / Infinity Cache either way (the L2s are not coherent
// across XCDs and are invalidated at its start), so dropping the line costs nothing. 16-byte stores only: narrower sc1 stores are one
// fabric write each. -DLVAE_WT_STORES=0 restores plain stores (A/B builds).
#ifndef LVAE_WT_STORES
#define LVAE_WT_STORES 1
#endif
__device__ __forceinline__ void store_wt4(float* p, f32x4 v) {
#if LVAE_WT_STORES
  // s_nop: a VALU write to the data registers of a store wider than 8 bytes needs two wait states after it on gfx940+ (LLVM GCNHazardRecognizer, VMEM store hazard); the compiler's hazard
  // recognizer inserts that for its own stores but cannot see into inline asm (found the hard way: corrupted lanes)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
#else
  *reinterpret_cast<f32x4*>(p) = v;
#endif
}

// LDS-only workgroup barrier. __syncthreads() is a workgroup FENCE, for which hipcc drains every outstanding memory operation of the wave
// (s_waitcnt vmcnt(0)): global loads that were requested early on purpose, and write-through stores, whose acknowledgement takes a trip
// to memory (1-2 us behind an epilogue's stores, measured with in-kernel stamps in resblock_img.hip). Where everything that crosses waves
// goes through LDS, only the LDS counter has to drain.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Kernel-argument warm-up. A launch's argument block is private to it (cold in the scalar cache and in L2), and hipcc reads a large block
// piecemeal, where a field is first used: several dependent s_load -> s_waitcnt pairs spread over the prologue, each a possible miss to
// memory. One scalar load per 64-byte line up front turns them into ONE round trip; the results are never used. -DLVAE_KERNARG_WARM=0
// compiles it out (A/B builds, tools/kernarg_ab.sh).
#ifndef LVAE_KERNARG_WARM
#define LVAE_KERNARG_WARM 1
#endif
template <int BYTES>
__device__ __forceinline__ void kernarg_warmup() {
#if LVAE_KERNARG_WARM
  const char* kp = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr int NL = (BYTES + 63) / 64;
  unsigned ka[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) asm volatile("s_load_dword %0, %1, %2" : "=s"(ka[i]) : "s"(kp), "n"(i * 64) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < NL; ++i) asm volatile("" ::"s"(ka[i]));
#endif
}

constexpr float kSeluAlpha = 1.6732632423543772848170429916717f;
constexpr float kSeluScale = 1.0507009873554804934193349852946f;

__device__ __forceinline__ float act_fwd(float x, int act) {
  switch (act) {
    case LVAE_ACT_ELU: return x > 0.f ? x : __expf(x) - 1.f;  // |abs err| < 1.2e-7: ELU output is O(1)
    case LVAE_ACT_RELU: return x > 0.f ? x : 0.f;
    case LVAE_ACT_LEAKYRELU: return x > 0.f ? x : 0.01f * x;
    case LVAE_ACT_SELU: return kSeluScale * (x > 0.f ? x : kSeluAlpha * (__expf(x) - 1.f));
    default: return x;
  }
}

// 4-wide activation with ONE wave-uniform branch for the two common cases (none, ELU) instead of a switch per element
__device__ __forceinline__ f32x4 act_fwd4(f32x4 v, int act) {
  if (act == LVAE_ACT_NONE) return v;
  if (act == LVAE_ACT_ELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : __expf(v[j]) - 1.f;
    return v;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = act_fwd(v[j], act);
  return v;
}

// one accumulator tile (16 values per lane) through the activation, again with one wave-uniform branch for the common cases
__device__ __forceinline__ f32x16 act_fwd16(f32x16 v, int act) {
  if (act == LVAE_ACT_NONE) return v;
  if (act == LVAE_ACT_ELU) {
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = v[j] > 0.f ? v[j] : __expf(v[j]) - 1.f;
    return v;
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = act_fwd(v[j], act);
  return v;
}

// derivative w.r.t. the pre-activation x
__device__ __forceinline__ float act_grad(float x, int act) {
  switch (act) {
    case LVAE_ACT_ELU: return x > 0.f ? 1.f : __expf(x);
    case LVAE_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case LVAE_ACT_LEAKYRELU: return x > 0.f ? 1.f : 0.01f;
    case LVAE_ACT_SELU: return kSeluScale * (x > 0.f ? 1.f : kSeluAlpha * __expf(x));
    default: return 1.f;
  }
}

// 4-wide derivative with ONE wave-uniform branch for the common case: with a run-time id every scalar act_grad() call is a chain of scalar
// compares and taken branches in the emitted code (the statistics epilogues of the dgrad kernels call it per element)
__device__ __forceinline__ f32x4 act_grad4(f32x4 u, int act) {
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

// derivative expressed from the activation OUTPUT y
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
  switch (act) {
    case LVAE_ACT_ELU: return y > 0.f ? 1.f : y + 1.f;
    case LVAE_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case LVAE_ACT_LEAKYRELU: return y > 0.f ? 1.f : 0.01f;
    case LVAE_ACT_SELU: return y > 0.f ? kSeluScale : y + kSeluScale * kSeluAlpha;
    default: return 1.f;
  }
}

// v_exp_f32 + v_rcp_f32 (1 ulp each): |abs err| < 2e-7 on a value in (0, 1); the IEEE expf + division it replaces cost ~25
// instructions per element in the gate kernels
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// numerically stable softplus, threshold 20 like torch
__device__ __forceinline__ float softplusf_(float x) { return x > 20.f ? x : log1pf(expf(x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over a 256-thread block; result valid in every thread. `red` = 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// n / d for a run-time constant d via a host-computed magic number: exact for n, d < 65536
// (M = ceil(2^32 / d): M*d - 2^32 < d, so the error term n*(M*d - 2^32) / 2^32 stays below 1).
inline uint32_t fastdiv_magic(int d) { return d <= 1 ? 0u : (uint32_t)((((uint64_t)1 << 32) + d - 1) / (uint64_t)d); }
__device__ __forceinline__ int fastdiv(int n, uint32_t magic) { return magic ? (int)__umulhi((uint32_t)n, magic) : n; }

inline int grid_for(int64_t work_items, int per_block, int cap = 256 * 8) {
  int64_t g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

}  // namespace lvae
