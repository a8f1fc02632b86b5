// tsx_pack.hpp -- reduced-precision packing helpers shared by the preconditioner kernel families
// (tsx_kernels_pc.hpp: one lane per column; tsx_kernels_pcs.hpp: segmented scan over the levels)
#pragma once
#include "tsx_dev.hpp"

typedef _Float16 tsx_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 tsx_h4 __attribute__((ext_vector_type(4)));
constexpr int TSX_P16_GROUPS = 8;
constexpr float TSX_FP8_SCALE = 64.0f;

// four fp8 e4m3 bytes of one word -> floats (still scaled by TSX_FP8_SCALE)
__device__ __forceinline__ void tsx_fp8x4(unsigned w, float (&o)[4]) {
  const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false);
  const auto hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
  o[0] = lo[0];
  o[1] = lo[1];
  o[2] = hi[0];
  o[3] = hi[1];
}
__device__ __forceinline__ unsigned tsx_to_fp8x4(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a * TSX_FP8_SCALE, b * TSX_FP8_SCALE, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c * TSX_FP8_SCALE, d * TSX_FP8_SCALE, w, true);
  return (unsigned)w;
}
__device__ __forceinline__ unsigned tsx_to_h2(float a, float b) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  h2 v;
  v[0] = (_Float16)a;
  v[1] = (_Float16)b;
  return __builtin_bit_cast(unsigned, v);
}

__device__ __forceinline__ unsigned short tsx_to_bf16(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even
  return (unsigned short)(u >> 16);
}
