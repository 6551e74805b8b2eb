// bf16 MFMA operand helpers shared by the bf16-precision kernels (gfx950 only).
#pragma once
#include "lvae_common.h"

namespace lvae {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// Two transposed 4 x 16 blocks of a [row][channel] bf16 image in LDS -> the 8 consecutive rows (the MFMA's k) of this lane's channel:
// ds_read_b64_tr_b16 hands a 4-row x 16-channel block to 16 lanes channel-major (the hardware transpose; semantics probed on the device
// with tools/tr_probe.hip). For k-step s (16 rows) and a 32-channel block at channel cb, lane (G = lane >> 4, i16 = lane & 15) passes
//   p0 = image + (16 s + 8 (G >> 1) + (i16 >> 2)) * pitch + cb + 16 (G & 1) + 4 (i16 & 3),   p1 = p0 + 4 * pitch
// and receives rows 16 s + 8 (lane >> 5) + 0..7 of channel cb + (lane & 31): the A (or B) fragment of v_mfma_f32_32x32x16_bf16.
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* p0, const __bf16* p1) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x4 to_bf16x4(const f32x4 v) { return bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]}; }

__device__ __forceinline__ bf16x8 to_bf16x8(const f32x4 lo, const f32x4 hi) {
  return bf16x8{(__bf16)lo[0], (__bf16)lo[1], (__bf16)lo[2], (__bf16)lo[3], (__bf16)hi[0], (__bf16)hi[1], (__bf16)hi[2], (__bf16)hi[3]};
}

// Four consecutive channels of an activation row stored as fp32 or bf16 (`bf`: wave-uniform, lvae_conv_desc.*_dtype); `off` in elements.
__device__ __forceinline__ f32x4 load4_dt(const float* base, size_t off, bool bf) {
  if (bf) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(base) + off);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
  return *reinterpret_cast<const f32x4*>(base + off);
}
// fp32: 16-byte write-through store (store_wt4); bf16: round to nearest even, one plain 8-byte store (a narrower sc1 store would be one
// fabric write each, MI355X_MICROARCH.md): 16 lanes cover the 128 bytes of a 64-channel row
__device__ __forceinline__ void store4_dt(float* base, size_t off, f32x4 v, bool bf) {
  if (bf) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(base) + off) = to_bf16x4(v);
  else store_wt4(base + off, v);
}

// v = out[0] + out[1] + ... exactly (SPLIT = 3: all 24 significant bits; SPLIT = 1: the round-to-nearest bf16 value)
template <int SPLIT>
__device__ __forceinline__ void split4(const f32x4 v, bf16x4 (&out)[SPLIT]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float r = v[j];
#pragma unroll
    for (int p = 0; p < SPLIT; ++p) {
      const __bf16 b = (__bf16)r;
      out[p][j] = b;
      if (p + 1 < SPLIT) r -= (float)b;  // exact: the remainder of a round-to-nearest to 8 bits has at most 16 significant bits
    }
  }
}

// Exact three-piece split of eight fp32 values straight into the three bf16x8 MFMA operand pieces, written pairwise so that hipcc
// emits ONE v_cvt_pk_bf16_f32 per two values and stage (left to itself it converts many values one at a time and converts again to
// pack: 6.5-7.5 vector instructions per value in the emitted code of the Winograd kernels; this form is 5.5): per pair and stage
// cvt_pk, two widenings (shift / mask of the packed word), two subtractions; the packed words ARE the operand pieces.
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split8_3(const f32x4 lo, const f32x4 hi, bf16x8 (&af)[3]) {
  u32x4 w[3];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float a = p < 2 ? lo[2 * p] : hi[2 * p - 4], b = p < 2 ? lo[2 * p + 1] : hi[2 * p - 3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const bf16x2 h = __builtin_convertvector(f32x2v{a, b}, bf16x2);
      const unsigned bits = __builtin_bit_cast(unsigned, h);
      w[q][p] = bits;
      if (q < 2) {
        a -= __builtin_bit_cast(float, bits << 16);          // exact: the remainder of a round-to-nearest to 8 bits has <= 16 significant bits
        b -= __builtin_bit_cast(float, bits & 0xffff0000u);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) af[q] = __builtin_bit_cast(bf16x8, w[q]);
}

// One entry of the batched weight pre-transform table (lvae_conv2d_prepare_weights): the same 64 bytes as the Winograd entry of
// conv3x3_wino.hip. kind: 1 | 3 = planes of a 3x3 weight for conv3x3_bf16.hip / resblock_img.hip; 33 | 35 = 32 + planes of a 1x1 gate weight
// for resblock_img.hip; 0 | 16 = Winograd (conv3x3_wino.hip)
struct BfPrepEntry {
  const float* w;
  __bf16* U;
  int64_t stap, sk, sn;
  int32_t K, N, Npad, flip;
  int32_t Kpad, kind;
};
static_assert(sizeof(BfPrepEntry) == 64, "entry layout is part of the C ABI (lvae_conv2d_prepare_entry)");

}  // namespace lvae
