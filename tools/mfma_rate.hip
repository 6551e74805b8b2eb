// How fast does one SIMD issue v_mfma_f32_32x32x16_bf16, and what does s_memtime tick at?  hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate
// Each wave runs NM MFMAs as chains of CH dependent instructions on NACC accumulators; 256 workgroups of WAVES waves (one per CU, LDS-padded).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, long long* stamps, int nm) {
  extern __shared__ float pad[];
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  long long t0 = __builtin_readcyclecounter();
  long long w0 = wall_clock64();
  for (int it = 0; it < nm / (6 * NACC); ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  long long w1 = wall_clock64();
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) s += acc[j][threadIdx.x & 15];
  if (s == 12345.f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = w1 - w0; }
  if (pad[0] == 1.f) out[1] = 1.f;
}

template <int NACC>
void run(int waves, int nm, float* out, long long* st) {
  hipFuncSetAttribute((const void*)k<NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<256, waves * 64, 100 * 1024>>>(out, st, nm);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) k<NACC><<<256, waves * 64, 100 * 1024>>>(out, st, nm);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
  const double us = ms * 1e3 / 20;
  const double per_simd = (double)nm * waves / 4;
  printf("waves %d  chains of 6 on %d accumulators  %d MFMAs/wave: %.1f us/launch, %.2f ns per MFMA per SIMD; s_memtime %lld ticks = %.1f per MFMA per SIMD; wall_clock64 %lld ticks (100 MHz => %.1f us) => s_memtime at %.2f GHz\n",
         waves, NACC, nm, us, us * 1e3 / per_simd, h[0], (double)h[0] / per_simd, h[1], h[1] / 100.0, (double)h[0] / (h[1] * 10.0));
}

int main() {
  float* out; long long* st;
  hipMalloc(&out, 64); hipMalloc(&st, 64);
  const int nm = 6 * 8 * 400;
  run<2>(8, nm, out, st);
  run<8>(8, nm, out, st);
  run<2>(4, nm, out, st);
  run<8>(4, nm, out, st);
  run<1>(4, nm, out, st);
  return 0;
}
