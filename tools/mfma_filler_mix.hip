// What do the fillers of an MFMA gap cost by KIND? One wave per SIMD (256 workgroups x 256 threads, one per CU), a loop of 12
// v_mfma_f32_32x32x16_bf16 with one block of vector work spread over the 12 gaps by sched_group_barrier (FILL instructions per gap).
// MODE 0: 48 v_add_f32; 1: 48 v_fma_f32 in four dependent chains; 2: the exact three-piece bf16 split of 8 values (44: v_cvt_pk_bf16_f32,
// v_lshlrev, v_and, v_sub) + 8 fold-back; 3: MODE 2 whose pieces ARE the A operand of the next 12 MFMAs (as in conv3x3_wino2_kernel);
// 4: 8 ds_read_b128 + 24 v_fma (the row / column combination).   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 (&af)[3]) {
  u32x4 w[3];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float a = x[2 * p], b = x[2 * p + 1];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
      const unsigned bits = __builtin_bit_cast(unsigned, h);
      w[q][p] = bits;
      if (q < 2) { a -= __builtin_bit_cast(float, bits << 16); b -= __builtin_bit_cast(float, bits & 0xffff0000u); }
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) af[q] = __builtin_bit_cast(bf16x8, w[q]);
}

template <int MODE, int FILL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(float* out, long long* stamps, int iters) {
  extern __shared__ float lds[];
  bf16x8 af[3], b;
  for (int i = 0; i < 8; ++i) { b[i] = (__bf16)(i * 0.5f); for (int q = 0; q < 3; ++q) af[q][i] = (__bf16)(threadIdx.x * 0.001f + i + q); }
  f32x16 acc[2];
  for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.37f + i;
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = i * 0.001f;
  __syncthreads();
  const float* lp = lds + (threadIdx.x & 63) * 68;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 an[3] = {af[0], af[1], af[2]};
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = x[i] + 1.0009765625f;
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 12; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);
    } else if (MODE == 2 || MODE == 3) {
      split8(x, an);
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = x[i] * 0.999f + 1.5f;
    } else {
      f32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4*>(lp + 4 * i + 64 * (it & 7));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ta = __builtin_fmaf(-1.f, v[1][e], v[0][e]), tb = __builtin_fmaf(-1.f, v[3][e], v[2][e]);
        const float tc = __builtin_fmaf(-1.f, v[5][e], v[4][e]), td = __builtin_fmaf(-1.f, v[7][e], v[6][e]);
        x[e] += ta - tb;
        x[4 + e] += tc - td;
      }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[c % 3], b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, af[(c + 1) % 3], acc[1], 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (MODE == 4 && g < 2) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, FILL, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 3) { af[0] = an[0]; af[1] = an[1]; af[2] = an[2]; }
    else if (MODE == 2) { x[0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, an[0])[0] ^ __builtin_bit_cast(u32x4, an[1])[1] ^ __builtin_bit_cast(u32x4, an[2])[2]); }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s + acc[0][threadIdx.x & 15] + acc[1][3] == 12345.f) out[0] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) stamps[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE, int FILL>
void run(const char* what, float* out, long long* st) {
  const int iters = 400;
  (void)hipFuncSetAttribute((const void*)k<MODE, FILL>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  k<MODE, FILL><<<256, 256, 100 * 1024>>>(out, st, iters);
  (void)hipDeviceSynchronize();
  long long h[4]; (void)hipMemcpy(h, st, 32, hipMemcpyDeviceToHost);
  printf("%-44s %d fillers/gap: %6.1f cycles per MFMA (wave 0; 32 = the matrix pipe alone)\n", what, FILL, (double)h[0] / (12.0 * iters));
}

int main() {
  float* out; long long* st;
  (void)hipMalloc(&out, 64); (void)hipMalloc(&st, 64);
  run<0, 4>("48 v_add_f32", out, st);
  run<1, 4>("48 v_fma_f32, four dependent chains", out, st);
  run<2, 5>("three-piece split of 8 values (52)", out, st);
  run<2, 7>("three-piece split of 8 values (52)", out, st);
  run<3, 5>("split, pieces feed the next MFMAs", out, st);
  run<3, 7>("split, pieces feed the next MFMAs", out, st);
  run<4, 3>("8 ds_read_b128 + 24 v_fma + 8 v_add", out, st);
  return 0;
}
