// Probe of ds_read_b64_tr_b16 semantics on gfx950: each 16-lane group reads a 4-row x 16-column block of 16-bit elements;
// lane 4q+p supplies the address of row q, columns 4p..4p+3; lane i receives column i of the 4 rows (row q in element q).
// build+run: hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* y) {
  __shared__ short lds[64 * 64];   // [row][col], value = row * 100 + col
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)((i / 64) * 100 + (i % 64));
  __syncthreads();
  const int l = threadIdx.x, G = l >> 4, i = l & 15;
  // group G reads rows 4G..4G+3, columns 16..31
  const short* p = lds + (4 * G + (i >> 2)) * 64 + 16 + 4 * (i & 3);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; ++j) y[l * 4 + j] = v[j];
}
int main() {
  short* d;
  hipMalloc(&d, 256 * sizeof(short));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int q = 0; q < 4; ++q) {
      const int want = (4 * (l >> 4) + q) * 100 + 16 + (l & 15);
      if (h[l * 4 + q] != want) ++bad;
    }
  printf("lane 0: %d %d %d %d | lane 5: %d %d %d %d | lane 17: %d %d %d %d | mismatches vs (row 4G+q, col 16+i): %d\n", h[0], h[1], h[2], h[3],
         h[20], h[21], h[22], h[23], h[68], h[69], h[70], h[71], bad);
  return bad != 0;
}
