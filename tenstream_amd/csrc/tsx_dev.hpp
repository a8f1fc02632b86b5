// tsx_dev.hpp -- device helpers shared by all kernel headers; hand-written HIP kernels for gfx950 (MI355X): 64-wide wavefronts, HBM-bound.
//
// All kernels are bandwidth-bound streaming kernels (arithmetic intensity ~0.36 flop/B, no MFMA):
// lanes run along x (i), the fastest index of every plane, so each wave instruction touches one
// contiguous 256/512-byte span per plane.  Reductions are wavefront-reduced (__shfl_down over 64
// lanes), one LDS hop per block, then per-block partials that a single-block scalar stage sums in a
// fixed order (deterministic; no float atomics).
#pragma once
#include "tsx_internal.hpp"

#define TSX_BLOCK 256

// stream direction tables (src/pprts.F90:339-343 for 3_10, :416-419 for 8_16): both solvers use
// is_inward = [F,T,F,T,...] for top and side streams, so parity of the index decides.
__host__ __device__ constexpr bool tsx_inward(int q) { return (q & 1) != 0; }

__device__ __forceinline__ double tsx_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// block-wide sum of NS values; thread 0 of the block writes partials[s * TSX_MAX_PARTIAL_BLOCKS + blockIdx.x]
template <int NS>
__device__ __forceinline__ void tsx_block_reduce_store(double (&v)[NS], double *__restrict__ partials) {
  __shared__ double sm[NS][TSX_BLOCK / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    double r = tsx_wave_sum(v[s]);
    if (lane == 0) sm[s][wv] = r;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      double r = 0;
#pragma unroll
      for (int q = 0; q < TSX_BLOCK / 64; ++q) r += sm[s][q];
      partials[(size_t)s * TSX_MAX_PARTIAL_BLOCKS + blockIdx.x] = r;
    }
  }
}

// XCD-aware chunk assignment: blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2),
// so give each XCD one contiguous eighth of the window of chunks in flight: the +-1 row (xm) and
// +-1 lane neighbours a cell reads are then served by that XCD's own L2.
__device__ __forceinline__ long long tsx_swizzle(long long b, long long nb) {
  return (nb & 7) == 0 ? (b & 7) * (nb >> 3) + (b >> 3) : b;
}

// ------------------------------------------------------------------------------------------------
// y = (I - T) x.  Restates op_mat_mult_ediff (src/pprts_shell.F90:413-519) in dst-owned form: the thread of
// cell (k,i,j) gathers the cell's D source streams and writes the D streams leaving it.  Surface row uses the
// assembled semantics (src/pprts.F90:5755-5794).
// FUSE bit0 (1): partial slot0 += w.y   (BiCGStab (rhat, v) and (s, t))
// FUSE bit1 (2): partial slot1 += x.y   (x = the operator's input at the same index)
// FUSE bit2 (4): partial slot2 += y.y   (BiCGStab (t, t))
// XT / WT: storage type of the input x and of w.  With a preconditioner the input is the preconditioned direction
// (p-hat, s-hat) which -- like the shadow residual rhat -- may be held in fp32: flexible BiCGStab (KSPFBCGS,
// src/pprts.F90:4342) allows any direction as long as the same stored vector feeds both A*dir and x += a*dir.
// Each thread owns CPT consecutive cells along x so that every plane is read with
// 8/16-byte (fp32 coefficients) and 16/32-byte (fp64 vectors) loads per lane; the +-x neighbours inside
// the group come from registers.  Requires xm % CPT == 0.
template <int CPT> struct TsxVec;
template <> struct TsxVec<1> {
  static __device__ __forceinline__ void ld(const double *p, double *o) { o[0] = p[0]; }
  static __device__ __forceinline__ void ld(const float *p, double *o) { o[0] = (double)p[0]; }
  static __device__ __forceinline__ void st(double *p, const double *v) { p[0] = v[0]; }
  static __device__ __forceinline__ void st(float *p, const double *v) { p[0] = (float)v[0]; }
};
template <> struct TsxVec<2> {
  static __device__ __forceinline__ void ld(const double *p, double *o) {
    const double2 v = *reinterpret_cast<const double2 *>(p);
    o[0] = v.x; o[1] = v.y;
  }
  static __device__ __forceinline__ void ld(const float *p, double *o) {
    const float2 v = *reinterpret_cast<const float2 *>(p);
    o[0] = (double)v.x; o[1] = (double)v.y;
  }
  static __device__ __forceinline__ void st(double *p, const double *v) {
    double2 o; o.x = v[0]; o.y = v[1];
    *reinterpret_cast<double2 *>(p) = o;
  }
  static __device__ __forceinline__ void st(float *p, const double *v) {
    *reinterpret_cast<float2 *>(p) = make_float2((float)v[0], (float)v[1]);
  }
};
template <> struct TsxVec<4> {
  static __device__ __forceinline__ void ld(const double *p, double *o) {
    const double2 a = reinterpret_cast<const double2 *>(p)[0], b = reinterpret_cast<const double2 *>(p)[1];
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
  }
  static __device__ __forceinline__ void ld(const float *p, double *o) {
    const float4 v = *reinterpret_cast<const float4 *>(p);
    o[0] = (double)v.x; o[1] = (double)v.y; o[2] = (double)v.z; o[3] = (double)v.w;
  }
  static __device__ __forceinline__ void st(double *p, const double *v) {
    double2 a, b; a.x = v[0]; a.y = v[1]; b.x = v[2]; b.y = v[3];
    reinterpret_cast<double2 *>(p)[0] = a;
    reinterpret_cast<double2 *>(p)[1] = b;
  }
  static __device__ __forceinline__ void st(float *p, const double *v) {
    *reinterpret_cast<float4 *>(p) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
};

// raw (unconverted) CPT-wide register images of a plane element group: lets loads be issued long before use
template <typename T, int CPT> struct TsxRaw;
template <> struct TsxRaw<float, 1> {
  typedef float type;
  static __device__ __forceinline__ type ld(const float *p) { return *p; }
  static __device__ __forceinline__ void cvt(type v, double *o) { o[0] = (double)v; }
};
template <> struct TsxRaw<float, 2> {
  typedef float2 type;
  static __device__ __forceinline__ type ld(const float *p) { return *reinterpret_cast<const float2 *>(p); }
  static __device__ __forceinline__ void cvt(type v, double *o) { o[0] = (double)v.x; o[1] = (double)v.y; }
};
template <> struct TsxRaw<double, 1> {
  typedef double type;
  static __device__ __forceinline__ type ld(const double *p) { return *p; }
  static __device__ __forceinline__ void cvt(type v, double *o) { o[0] = v; }
};
template <> struct TsxRaw<double, 2> {
  typedef double2 type;
  static __device__ __forceinline__ type ld(const double *p) { return *reinterpret_cast<const double2 *>(p); }
  static __device__ __forceinline__ void cvt(type v, double *o) { o[0] = v.x; o[1] = v.y; }
};

// ---- colour-split order of the red-black preconditioner's private arrays: within a row of columns the xm/2 columns of
// colour (i + j) & 1 == 0 first, then colour 1, so that a pass over one colour streams contiguous memory
__host__ __device__ __forceinline__ long long tsx_split_col(int i, int j, int xm) {
  return (long long)j * xm + (long long)((i + j) & 1) * (xm >> 1) + (i >> 1);
}
// position of element idx of an N-vector (D planes of Nc cells + D tail rows of ncol columns) in that order
__device__ __forceinline__ long long tsx_split_pos(long long idx, const TsxGeo &g) {
  const long long body = (long long)g.D * g.Nc;
  if (idx < body) {
    const long long d = idx / g.Nc, c = idx - d * g.Nc;
    const int i = (int)(c % g.xm);
    const long long t = c / g.xm;
    const int j = (int)(t % g.ym);
    return d * g.Nc + (t / g.ym) * g.ncol + tsx_split_col(i, j, g.xm);
  }
  const long long t = idx - body, d = t / g.ncol;
  const int col = (int)(t - d * g.ncol);
  return body + d * g.ncol + tsx_split_col(col % g.xm, col / g.xm, g.xm);
}

// ---- 64-bit hash of a transport block (shared storage of identical blocks, tsx_dedup.hip): fed coefficient by coefficient,
// by tsx_k_dd_hash from the stored planes or by the kernel that produces the block while it still has it in registers
constexpr unsigned long long TSX_DD_EMPTY = 0ull;
constexpr unsigned long long TSX_DD_H1D = 0x1d1d1d1d1d1d1d1dull;  // all cells of 1-D layers (their blocks are never read)
constexpr unsigned long long TSX_DD_SEED = 0x243f6a8885a308d3ull;
__device__ __forceinline__ unsigned long long tsx_mix64(unsigned long long h, unsigned long long v) {
  h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 33;
  return h;
}
__device__ __forceinline__ unsigned long long tsx_dd_hash_step(unsigned long long h, int q, float coeff) {
  return tsx_mix64(h, (unsigned long long)__float_as_uint(coeff) + ((unsigned long long)q << 32));
}
__device__ __forceinline__ unsigned long long tsx_dd_hash_final(unsigned long long h) {
  return (h == TSX_DD_EMPTY || h == TSX_DD_H1D) ? h ^ 0x5555555555555555ull : h;
}
