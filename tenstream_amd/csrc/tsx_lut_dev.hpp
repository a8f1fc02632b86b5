// tsx_lut_dev.hpp -- device side of the coefficient lookup that more than one translation unit needs (templates and inline
// functions only): the table descriptor, bisection search, N-linear weights with lattice snapping, the interpolation of one
// diffuse block, and the kernel that interpolates only the distinct coordinate tuples (tsx_dedup.hip "coordinates first").
#pragma once
#include "tsx_dev.hpp"

// ------------------------------------------------------------------------------------------------
// Coefficient lookup on the device: get_coeff -> LUT_get_diff2diff -> interp_vec_bilinear_iterative
// (src/pprts_base.F90:1517-1542, src/optprop_LUT.F90:1560-1596, src/interpolation.F90:317-360, snapping :546-556,
//  bisection src/search.fypp:177-228).  real32 arithmetic in the reference's operation order (explicit
// __f*_rn intrinsics keep the compiler from contracting mul+add), so the planes are bit-identical to what
// alloc_coeff_diff2diff stores.  One thread per cell; the table (34 MB for 3_10) lives in L2/Infinity Cache,
// the D*D outputs are written plane-wise (coalesced along x).
struct TsxLutDev {
  int ndim;
  int nvec;
  int n[8];
  int axis_off[8];      // offset of each axis in `axes`
  long long offs[8];    // ndarray_offsets (src/helper_functions.fypp:2431-2437)
  const float *axes;
  const float *table;   // (nvec, nentries) column-major == src/mmap.F90 payload
};

__device__ __forceinline__ float tsx_search_sorted_bisection(const float *__restrict__ arr1, int n, float val) {
  const float *arr = arr1 - 1;  // 1-based like the reference
  int i = 1, j = n;
  // ascending axes only (all LUT presets are ascending)
  for (;;) {
    const int k = (i + j) / 2;
    if (val < arr[k]) j = k;
    else i = k;
    if (i + 1 >= j) {
      float inc = 0.0f;
      if (i != j) inc = __fdiv_rn(__fsub_rn(val, arr[i]), __fsub_rn(arr[j], arr[i]));
      float res = __fadd_rn((float)i, inc);
      res = fmaxf(1.0f, res);
      res = fminf((float)n, res);
      return res;
    }
  }
}

template <int NDIM>
__device__ __forceinline__ void tsx_lut_weights(const TsxLutDev &L, const float (&sample)[NDIM], int &ninterp,
                                                long long &ofs_base, long long (&ioff_lo)[NDIM], long long (&ioff_hi)[NDIM],
                                                float (&wlo)[NDIM], float (&whi)[NDIM]) {
  ninterp = 0;
  ofs_base = 0;  // 0-based entry offset
#pragma unroll
  for (int d = 0; d < NDIM; ++d) {
    const float pti = tsx_search_sorted_bisection(L.axes + L.axis_off[d], L.n[d], sample[d]);
    const float frac = __fsub_rn(pti, (float)(int)pti);
    const bool interp = !(frac < 1e-3f) && !(frac > __fsub_rn(1.0f, 1e-3f));
    if (interp) {
      const int b = (int)pti;
      whi[ninterp] = __fsub_rn(pti, (float)b);
      wlo[ninterp] = __fsub_rn(1.0f, whi[ninterp]);
      ioff_lo[ninterp] = L.offs[d] * (b - 1);
      ioff_hi[ninterp] = L.offs[d] * b;
      ++ninterp;
    } else {
      ofs_base += L.offs[d] * ((long long)lrintf(pti) - 1);  // nint; .5 cannot occur (snapped range only)
    }
  }
}

// the clamps of get_coeff on a cell's LUT coordinates (aspect, w0, tauz, g as tsx_k_cell_samples leaves them): aspect from below,
// tauz and w0 into their axes (src/pprts_base.F90:1517-1533).  The interpolated block is a deterministic function of the result.
__device__ __forceinline__ float4 tsx_lut_diff_clamp(const TsxLutDev &L, float4 v) {
  const float *ax = L.axes;
  v.x = fmaxf(ax[L.axis_off[2]], v.x);
  v.z = fmaxf(ax[L.axis_off[0]], fminf(ax[L.axis_off[0] + L.n[0] - 1], v.z));
  v.y = fmaxf(ax[L.axis_off[1]], fminf(ax[L.axis_off[1] + L.n[1] - 1], v.y));
  return v;
}
// N-linear interpolation of one diffuse block at clamped coordinates (aspect, w0, tauz, g) -- the body of tsx_k_lut_diff2diff
template <int DD>
__device__ __forceinline__ void tsx_lut_diff_block(const TsxLutDev &L, float4 cv, float (&acc)[DD]) {
  const float sample[4] = {cv.z, cv.y, cv.x, cv.w};
  int ninterp;
  long long ofs_base, ioff_lo[4], ioff_hi[4];
  float wlo[4], whi[4];
  tsx_lut_weights<4>(L, sample, ninterp, ofs_base, ioff_lo, ioff_hi, wlo, whi);
#pragma unroll
  for (int q = 0; q < DD; ++q) acc[q] = 0.0f;
  for (int b = 0; b < (1 << ninterp); ++b) {
    long long ofs = ofs_base;
    float w = 1.0f;
    for (int d = 0; d < ninterp; ++d) {
      if (b & (1 << d)) {
        ofs += ioff_hi[d];
        w = __fmul_rn(w, whi[d]);
      } else {
        ofs += ioff_lo[d];
        w = __fmul_rn(w, wlo[d]);
      }
    }
    const float4 *__restrict__ colp = reinterpret_cast<const float4 *>(L.table + (size_t)ofs * DD);
#pragma unroll
    for (int q4 = 0; q4 < DD / 4; ++q4) {
      const float4 v = colp[q4];
      acc[4 * q4 + 0] = __fadd_rn(acc[4 * q4 + 0], __fmul_rn(w, v.x));
      acc[4 * q4 + 1] = __fadd_rn(acc[4 * q4 + 1], __fmul_rn(w, v.y));
      acc[4 * q4 + 2] = __fadd_rn(acc[4 * q4 + 2], __fmul_rn(w, v.z));
      acc[4 * q4 + 3] = __fadd_rn(acc[4 * q4 + 3], __fmul_rn(w, v.w));
    }
  }
}

// Sharing keyed on the LUT coordinates (round 4, tsx_dedup.hip "coordinates first"): only the distinct coordinate tuples are
// interpolated, straight into the shared storage -- plane-major Cd[q * nent + id] (what the packing reads) and entry-major
// Ce[id * DD + q] (what the operator reads).  No dense per-cell planes are written (1.68 GB per g-point at 256 x 256 x 64).
// ent_cell[id] = the representative cell of entry id; the entry of the 1-D layers' cells holds zeros (never read).
template <int DD>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_lut_diff2diff_ent(TsxGeo g, TsxLutDev L, const uint8_t *__restrict__ l1d,
                                                                     long long nent, const int *__restrict__ ent_cell,
                                                                     const float4 *__restrict__ samp, float *__restrict__ Cd,
                                                                     float *__restrict__ Ce) {
  for (long long id = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; id < nent; id += (long long)gridDim.x * TSX_BLOCK) {
    const int cell = ent_cell[id];
    float acc[DD];
    if (l1d[cell / g.ncol]) {
#pragma unroll
      for (int q = 0; q < DD; ++q) acc[q] = 0.0f;
    } else {
      tsx_lut_diff_block<DD>(L, tsx_lut_diff_clamp(L, samp[cell]), acc);
    }
#pragma unroll
    for (int q = 0; q < DD; ++q) Cd[(size_t)q * nent + id] = acc[q];
    float4 *__restrict__ row = reinterpret_cast<float4 *>(Ce + (size_t)id * DD);  // DD * 4 bytes is a multiple of 16
#pragma unroll
    for (int q4 = 0; q4 < DD / 4; ++q4) row[q4] = make_float4(acc[4 * q4], acc[4 * q4 + 1], acc[4 * q4 + 2], acc[4 * q4 + 3]);
  }
}

