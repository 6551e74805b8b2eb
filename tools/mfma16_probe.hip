// Lane layout probe of v_mfma_f32_16x16x32_bf16 on the device (as tools/tr_probe.hip did for ds_read_b64_tr_b16).
// Assumed: A[i][k]: lane = 16 * (k / 8) + i, element k % 8;  B[k][j]: lane = 16 * (k / 8) + j, element k % 8;  D[i][j]: lane = 16 * (i / 4) + j, register i % 4.
// hipcc --offload-arch=gfx950 -O2 tools/mfma16_probe.hip -o /tmp/mfma16_probe && /tmp/mfma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* A, const float* B, float* D) {   // A [16][32], B [32][16], D [16][16] row-major
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * (l / 16) + e;
    a[e] = (__bf16)A[(l % 16) * 32 + k];
    b[e] = (__bf16)B[k * 16 + (l % 16)];
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + (l % 16)] = c[r];
}
int main() {
  float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 24) - 128) / 16.f; };   // exactly representable in bf16
  for (float& v : hA) v = rnd();
  for (float& v : hB) v = rnd();
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      float acc = 0.f;
      for (int k = 0; k < 32; ++k) acc += hA[i * 32 + k] * hB[k * 16 + j];
      ref[i * 16 + j] = acc;
    }
  float *dA, *dB, *dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
  hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
  printf("v_mfma_f32_16x16x32_bf16 with the assumed lane layout: %d of 256 outputs differ from the host product\n", bad);
  return bad != 0;
}
