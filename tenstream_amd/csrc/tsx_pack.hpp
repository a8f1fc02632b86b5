// tsx_pack.hpp -- reduced-precision packing helpers shared by the preconditioner kernel families
// (tsx_kernels_pc.hpp: one lane per column; tsx_kernels_pcs.hpp: segmented scan over the levels)
#pragma once
#include "tsx_dev.hpp"

// tiny dense helpers for the H x H blocks of the two-stream recurrences (H = 1: 3_10, H = 4: 8_16)
template <int H>
struct TsxSm {  // tiny dense helpers, fully unrolled
  static __device__ __forceinline__ void matvec(const double (&M)[H][H], const double (&v)[H], double (&o)[H]) {
#pragma unroll
    for (int a = 0; a < H; ++a) {
      double t = 0.0;
#pragma unroll
      for (int b = 0; b < H; ++b) t += M[a][b] * v[b];
      o[a] = t;
    }
  }
  static __device__ __forceinline__ void matmul(const double (&X)[H][H], const double (&Y)[H][H], double (&O)[H][H]) {
#pragma unroll
    for (int a = 0; a < H; ++a)
#pragma unroll
      for (int b = 0; b < H; ++b) {
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < H; ++c) t += X[a][c] * Y[c][b];
        O[a][b] = t;
      }
  }
  // O = (I - X)^-1 by Gauss-Jordan without pivoting (I - Rdu*A is strictly diagonally dominant: entries of
  // Rdu*A are products of energy-conserving transfer coefficients, row sums < 1)
  static __device__ __forceinline__ void inv_i_minus(const double (&X)[H][H], double (&O)[H][H]) {
    double W[H][H];
#pragma unroll
    for (int a = 0; a < H; ++a)
#pragma unroll
      for (int b = 0; b < H; ++b) {
        W[a][b] = (a == b ? 1.0 : 0.0) - X[a][b];
        O[a][b] = (a == b ? 1.0 : 0.0);
      }
#pragma unroll
    for (int c = 0; c < H; ++c) {
      const double piv = 1.0 / W[c][c];
#pragma unroll
      for (int b = 0; b < H; ++b) {
        W[c][b] *= piv;
        O[c][b] *= piv;
      }
#pragma unroll
      for (int a = 0; a < H; ++a) {
        if (a == c) continue;
        const double f = W[a][c];
#pragma unroll
        for (int b = 0; b < H; ++b) {
          W[a][b] -= f * W[c][b];
          O[a][b] -= f * O[c][b];
        }
      }
    }
  }
};

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
