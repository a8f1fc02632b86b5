// tsx_pipeline.hpp -- kernels around the diffuse solve so that a whole g-point stays resident on the GPU:
// direct-beam coefficient lookup (K5), direct sweep (K6), source term (K7), flux divergence (K9), result
// assembly (K8/K12).  Same conventions as tsx_kernels.hpp: lanes along x, one plane per stream.
//
// Direct streams are stored on the column they sit on (reference ownership), one plane per stream:
//     E[s * Ncl + (k * ym + j) * xm + i],  k = 0..Nz (levels),  Ncl = (Nz+1) * ncol
#pragma once
#include "tsx_kernels.hpp"

struct TsxSun {
  double phi, theta, mu, costheta, symmetry_phi;
  int xinc, yinc;
};

// Where coefficient q = dst * S + src of the table's entry goes after the sun-quadrant symmetries: the tables only hold
// phi in [0, 90]; for a sun in the east (xinc == 0) / north (yinc == 0) the streams of the mirrored box are relabelled
// (get_coeff_cube, src/optprop.F90:571-576): first the east routine, then the north routine, each a list of assignments
// `coeff(dst block) = newcoeff(src permutation + other dst block)`.  Only the dst blocks the reference ASSIGNS change; a block
// whose line is commented out there keeps its values *and its source order* (pinned by tests/golden/coeff_symmetry.json,
// which is the reference's text interpreted statement by statement):
//   3_10  dir2dir: none (dir2dir_coeff_symmetry_none, :1256-1266);  dir2diff (dir3_to_diff10_coeff_symmetry, :1009-1045):
//         east: dst 3<->4, 5<->6; north: dst 7<->8, 9<->10 (1-based), sources as they are
//   8_16  dir2dir (dir2dir8_coeff_symmetry, :1268-1302): all 8 dst blocks assigned.  east: dst 1<->2, 3<->4, src [2,1,4,3,5..8];
//         north: dst 1<->3, 2<->4, src [3,4,1,2,5..8]
//         dir2diff (dir8_to_diff16_coeff_symmetry, :1186-1240): east assigns dst 3<->7, 4<->8, 9<->10, 11<->12 with
//         src [2,1,4,3,5..8] -- dst 1,2,5,6,13..16 untouched; north assigns dst 1<->5, 2<->6, 13<->14, 15<->16 with
//         src [3,4,1,2,5..8] -- dst 3,4,7..12 untouched
// The device scatters: the table's entry (d0, s0) is stored at the position the assignments move it to (the block maps and
// the source permutations are involutions, so "moved to" = "read from").
template <int NV, int S, bool DIR2DIFF>
__device__ __forceinline__ int tsx_dir_symmetry(int q, int east, int north) {
  int d = q / S, s = q % S;
  if (S == 3) {
    if (!DIR2DIFF) return q;
    if (east && d >= 2 && d <= 5) d ^= 1;
    if (north && d >= 6 && d <= 9) d ^= 1;
    return d * S + s;
  }
  // S == 8
  if (east) {
    bool moved = true;
    if (DIR2DIFF) {
      if (d == 2 || d == 3) d += 4;
      else if (d == 6 || d == 7) d -= 4;
      else if (d >= 8 && d <= 11) d ^= 1;
      else moved = false;
    } else {
      if (d < 4) d ^= 1;  // dst 5..8 are assigned from themselves, with the source permutation
    }
    if (moved && s < 4) s ^= 1;  // [2,1,4,3]
  }
  if (north) {
    bool moved = true;
    if (DIR2DIFF) {
      if (d == 0 || d == 1) d += 4;
      else if (d == 4 || d == 5) d -= 4;
      else if (d >= 12) d ^= 1;
      else moved = false;
    } else {
      if (d < 4) d ^= 2;
    }
    if (moved && s < 4) s ^= 2;  // [3,4,1,2]
  }
  return d * S + s;
}

// ---- K5: 6-D lookup [tau, w0, aspect, g, phi, theta] -> NV planes, stream relabelling for a sun from east / north
//      (tsx_dir_symmetry).
template <int NV, int S, bool DIR2DIFF>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_lut_dir(TsxGeo g, TsxLutDev L, const double *__restrict__ kabs,
                                                           const double *__restrict__ ksca, const double *__restrict__ gg,
                                                           const double *__restrict__ dz, double dx, float sym_phi, float theta,
                                                           int lswitch_east, int lswitch_north,
                                                           const uint8_t *__restrict__ l1d, float *__restrict__ C,
                                                           const float4 *__restrict__ samp) {
  const int xm = g.xm, ym = g.ym, Nz = g.Nz;
  const long long Nc = g.Nc;
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const int i = (int)(c % xm);
    const long long t = c / xm;
    const int j = (int)(t % ym);
    const int k = (int)(t / ym);
    if (l1d[k]) continue;
    float aspect, w0, tauz, gcell;
    if (samp) {  // the cells' LUT coordinates in cell order (tsx_k_cell_samples)
      const float4 v = samp[c];
      aspect = v.x, w0 = v.y, tauz = v.z, gcell = v.w;
    } else {
      const size_t r = (size_t)k + (size_t)Nz * ((size_t)i + (size_t)xm * j);
      const double ka = kabs[r], ks = ksca[r], dzz = dz[r];
      aspect = (float)(dzz / dx);
      w0 = (float)(ks / fmax(ka + ks, 2.220446049250313e-16));
      tauz = (float)((ka + ks) * dzz);
      gcell = (float)gg[r];
    }
    const float *ax = L.axes;
    aspect = fmaxf(ax[L.axis_off[2]], aspect);
    tauz = fmaxf(ax[L.axis_off[0]], fminf(ax[L.axis_off[0] + L.n[0] - 1], tauz));
    w0 = fmaxf(ax[L.axis_off[1]], fminf(ax[L.axis_off[1] + L.n[1] - 1], w0));
    const float sample[6] = {tauz, w0, aspect, gcell, sym_phi, theta};
    int ninterp;
    long long ofs_base, ioff_lo[6], ioff_hi[6];
    float wlo[6], whi[6];
    tsx_lut_weights<6>(L, sample, ninterp, ofs_base, ioff_lo, ioff_hi, wlo, whi);
    float acc[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) acc[q] = 0.0f;
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
      const float *__restrict__ colp = L.table + (size_t)ofs * NV;
#pragma unroll
      for (int q = 0; q < NV; ++q) acc[q] = __fadd_rn(acc[q], __fmul_rn(w, colp[q]));
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) C[(size_t)tsx_dir_symmetry<NV, S, DIR2DIFF>(q, lswitch_east, lswitch_north) * Nc + c] = acc[q];
  }
}

// ---- one raw table lookup (pprts_f2c_opp_get_coeff, c_wrapper/f2c_pprts.F90:627-685 -> get_coeff_cube,
//      src/optprop.F90:549-582): the caller's (tauz, w0, g, aspect_zx[, phi, theta]) go to the interpolation as they are,
//      only aspect_zx is raised to the table's lower bound; quadrant symmetry like the cell kernels.  One thread.
template <int NDIM>
__global__ void tsx_k_opp_point(TsxLutDev L, float tauz, float w0, float g, float aspect, float phi, float theta, int S,
                                int dir2diff, int east, int north, float *__restrict__ out) {
  if (threadIdx.x || blockIdx.x) return;
  aspect = fmaxf(L.axes[L.axis_off[2]], aspect);
  float sample[NDIM];
  sample[0] = tauz;
  sample[1] = w0;
  sample[2] = aspect;
  sample[3] = g;
  if (NDIM == 6) {
    sample[NDIM - 2] = phi;
    sample[NDIM - 1] = theta;
  }
  int ninterp;
  long long ofs_base, ioff_lo[NDIM], ioff_hi[NDIM];
  float wlo[NDIM], whi[NDIM];
  tsx_lut_weights<NDIM>(L, sample, ninterp, ofs_base, ioff_lo, ioff_hi, wlo, whi);
  const int NV = L.nvec;
  for (int q = 0; q < NV; ++q) {
    float acc = 0.0f;
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
      acc = __fadd_rn(acc, __fmul_rn(w, L.table[(size_t)ofs * NV + q]));
    }
    int qo = q;
    if (NDIM == 6) {
      if (S == 3) qo = dir2diff ? tsx_dir_symmetry<30, 3, true>(q, east, north) : q;
      else qo = dir2diff ? tsx_dir_symmetry<128, 8, true>(q, east, north) : tsx_dir_symmetry<64, 8, false>(q, east, north);
    }
    out[qo] = acc;
  }
}

// face buffers of the direct beam on several ranks (W, E, S, N); all null on one periodic rank
struct TsxDirHalo {
  double *sendW, *sendE, *sendS, *sendN;
  const double *recvW, *recvE, *recvS, *recvN;
};

// ---- K6: one sweep of the direct beam (explicit_edir_forward_sweep, src/pprts_explicit.F90:330-459): the thread of a
//      column marches down in k (fresh top stream), side streams of the neighbouring columns come from the previous
//      sweep.  Same fixed point as the reference's lexicographic sweep; iterated until ||x_new - x_old|| converges
//      (src/pprts_explicit.F90:168-218).  slot0 += ||new - old||^2.
template <int DTOP, int DSIDE>
__global__ __launch_bounds__(64) void tsx_k_edir_sweep(TsxGeo g, TsxSun sun, const float *__restrict__ T,
                                                       const uint8_t *__restrict__ l1d, const double *__restrict__ a33,
                                                       double inc_solar, const double *__restrict__ xo, double *__restrict__ xn,
                                                       double *__restrict__ partials, const int *__restrict__ done,
                                                       TsxDirHalo hb) {
  // Several ranks (g.wrap_x / g.wrap_y false): the one x (y) face per row that is off-rank is either my downwind output
  // (sun moving +x: cell xm-1 writes it -> send buffer, the east rank stores it as its face 0 in tsx_k_edir_unpack) or my
  // upwind source (sun moving -x: cell xm-1 reads face xm = the east rank's face 0 -> received buffer), i.e. exactly
  // exchange_direct_boundary (src/pprts_explicit.F90:1076-1140).  Buffers: [q][k][j] resp. [q][k][i].
  constexpr int S = DTOP + 2 * DSIDE;
  if (done && *done) return;
  const bool offx = g.wrap_x == 0, offy = g.wrap_y == 0;
  const int xm = g.xm, ym = g.ym, Nz = g.Nz, ncol = g.ncol;
  const long long Nc = g.Nc, Ncl = (long long)(Nz + 1) * ncol;
  const int col = blockIdx.x * 64 + threadIdx.x;
  double sum[1] = {0.0};
  if (col < ncol) {
    const int i = col % xm, j = col / xm;
    // upwind neighbours (periodic): x-side source sits at face i+1-xinc, y-side at face j+1-yinc
    const int iu = sun.xinc ? i : (i + 1 == xm ? 0 : i + 1);
    const int ju = sun.yinc ? j : (j + 1 == ym ? 0 : j + 1);
    // destination faces: i+xinc, j+yinc
    const int id = sun.xinc ? (i + 1 == xm ? 0 : i + 1) : i;
    const int jd = sun.yinc ? (j + 1 == ym ? 0 : j + 1) : j;
    double top[DTOP];
#pragma unroll
    for (int q = 0; q < DTOP; ++q) {
      top[q] = inc_solar;  // setup_incSolar: level 0 top streams (src/pprts_base.F90:1164-1176)
      const double old = xo[(size_t)q * Ncl + col];
      xn[(size_t)q * Ncl + col] = top[q];
      sum[0] += (top[q] - old) * (top[q] - old);
    }
    for (int k = 0; k < Nz; ++k) {
      const size_t c = (size_t)k * ncol + col;
      if (l1d[k]) {
        const double t33 = a33[c];
#pragma unroll
        for (int q = 0; q < DTOP; ++q) top[q] *= t33;
        // side streams of 1-D layers are never written by the reference: carry the old values
#pragma unroll
        for (int q = 0; q < 2 * DSIDE; ++q) {
          const size_t o = (size_t)(DTOP + q) * Ncl + c;
          xn[o] = xo[o];
        }
      } else {
        double src[S];
#pragma unroll
        for (int q = 0; q < DTOP; ++q) src[q] = top[q];
#pragma unroll
        for (int q = 0; q < DSIDE; ++q) {
          src[DTOP + q] = (offx && !sun.xinc && i + 1 == xm) ? hb.recvE[((size_t)q * Nz + k) * ym + j]
                                                             : xo[(size_t)(DTOP + q) * Ncl + (size_t)k * ncol + (size_t)j * xm + iu];
          src[DTOP + DSIDE + q] = (offy && !sun.yinc && j + 1 == ym)
                                      ? hb.recvN[((size_t)q * Nz + k) * xm + i]
                                      : xo[(size_t)(DTOP + DSIDE + q) * Ncl + (size_t)k * ncol + (size_t)ju * xm + i];
        }
        double out[S];
#pragma unroll
        for (int d = 0; d < S; ++d) {
          double a = 0.0;
#pragma unroll
          for (int s = 0; s < S; ++s) a += src[s] * (double)T[(size_t)(d * S + s) * Nc + c];
          out[d] = a;
        }
#pragma unroll
        for (int q = 0; q < DTOP; ++q) top[q] = out[q];
#pragma unroll
        for (int q = 0; q < DSIDE; ++q) {
          const size_t ox = (size_t)(DTOP + q) * Ncl + (size_t)k * ncol + (size_t)j * xm + id;
          const size_t oy = (size_t)(DTOP + DSIDE + q) * Ncl + (size_t)k * ncol + (size_t)jd * xm + i;
          if (offx && sun.xinc && i + 1 == xm) {
            hb.sendE[((size_t)q * Nz + k) * ym + j] = out[DTOP + q];  // lands on the east rank's face 0
          } else {
            const double dx_ = out[DTOP + q] - xo[ox];
            xn[ox] = out[DTOP + q];
            sum[0] += dx_ * dx_;
            if (offx && !sun.xinc && i == 0) hb.sendW[((size_t)q * Nz + k) * ym + j] = out[DTOP + q];  // the west rank's face xm
          }
          if (offy && sun.yinc && j + 1 == ym) {
            hb.sendN[((size_t)q * Nz + k) * xm + i] = out[DTOP + DSIDE + q];
          } else {
            const double dy_ = out[DTOP + DSIDE + q] - xo[oy];
            xn[oy] = out[DTOP + DSIDE + q];
            sum[0] += dy_ * dy_;
            if (offy && !sun.yinc && j == 0) hb.sendS[((size_t)q * Nz + k) * xm + i] = out[DTOP + DSIDE + q];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < DTOP; ++q) {
        const size_t o = (size_t)q * Ncl + (size_t)(k + 1) * ncol + col;
        const double old = xo[o];
        xn[o] = top[q];
        sum[0] += (top[q] - old) * (top[q] - old);
      }
    }
    // side entries at the bottom level are dummies: keep
#pragma unroll
    for (int q = 0; q < 2 * DSIDE; ++q) {
      const size_t o = (size_t)(DTOP + q) * Ncl + (size_t)Nz * ncol + col;
      xn[o] = xo[o];
    }
  }
  // block of 64 = one wave
  double r = tsx_wave_sum(sum[0]);
  if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

// ---- K6, tiled (round 3): one wave = a tile of 8 x 8 columns marching down together.  Within a layer the side streams that
//      leave a column enter its downwind neighbour IN THE SAME SWEEP when both are lanes of the tile: the layer's 64 cells are
//      resolved in registers by 15 substeps (lane (ii, jj), counted in the sun's order of travel, is final after substep
//      ii + jj), each a shuffle of the upwind lanes' side streams and the S x S product -- arithmetic that the memory-bound sweep
//      has to spare.  The one-column-per-thread sweep above moves the beam ONE column sideways per sweep (35 sweeps per solve of
//      the spectral loop's 256 x 256 x 64 domain); this one moves it a tile per sweep.  Across tile edges (and rank faces) the
//      sources come from the previous sweep exactly as above: same fixed point, same residual definition, same stop rule.
//      (A skewed variant -- lane (ii, jj) on layer t - ii - jj -- was measured first: every lane of a load on another layer,
//      6.6 x the time per sweep.)
template <int DTOP, int DSIDE>
__global__ __launch_bounds__(64) void tsx_k_edir_sweep_tiled(TsxGeo g, TsxSun sun, const float *__restrict__ T,
                                                             const uint8_t *__restrict__ l1d, const double *__restrict__ a33,
                                                             double inc_solar, const double *__restrict__ xo,
                                                             double *__restrict__ xn, double *__restrict__ partials,
                                                             const int *__restrict__ done, TsxDirHalo hb, int tiles_x) {
  constexpr int S = DTOP + 2 * DSIDE, TW = 8;
  if (done && *done) return;
  const bool offx = g.wrap_x == 0, offy = g.wrap_y == 0;
  const int xm = g.xm, ym = g.ym, Nz = g.Nz, ncol = g.ncol;
  const long long Nc = g.Nc, Ncl = (long long)(Nz + 1) * ncol;
  const int ii = threadIdx.x % TW, jj = threadIdx.x / TW;  // position in the tile in the sun's order: upwind = ii - 1, jj - 1
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int i_ = tx * TW + (sun.xinc ? ii : TW - 1 - ii), j_ = ty * TW + (sun.yinc ? jj : TW - 1 - jj);
  const bool inside = i_ < xm && j_ < ym;
  const int i = inside ? i_ : 0, j = inside ? j_ : 0;  // lanes beyond the domain shadow column 0 (loads valid, nothing stored)
  const int col = j * xm + i;
  // the upwind neighbour is a lane of this wave and a column of this rank: its side streams arrive by shuffle
  const bool shx = inside && ii > 0 && (sun.xinc || i + 1 < xm), shy = inside && jj > 0 && (sun.yinc || j + 1 < ym);
  const int iu = sun.xinc ? i : (i + 1 == xm ? 0 : i + 1);
  const int ju = sun.yinc ? j : (j + 1 == ym ? 0 : j + 1);
  const int id = sun.xinc ? (i + 1 == xm ? 0 : i + 1) : i;
  const int jd = sun.yinc ? (j + 1 == ym ? 0 : j + 1) : j;
  double sum = 0.0;
  double top[DTOP];
#pragma unroll
  for (int q = 0; q < DTOP; ++q) top[q] = inc_solar;  // setup_incSolar: level 0 top streams (src/pprts_base.F90:1164-1176)
  if (inside) {
#pragma unroll
    for (int q = 0; q < DTOP; ++q) {
      const double old = xo[(size_t)q * Ncl + col];
      xn[(size_t)q * Ncl + col] = top[q];
      sum += (top[q] - old) * (top[q] - old);
    }
  }
  for (int k = 0; k < Nz; ++k) {
    const size_t c = (size_t)k * ncol + col;
    if (l1d[k]) {  // the same for every lane
      const double t33 = a33[c];
#pragma unroll
      for (int q = 0; q < DTOP; ++q) top[q] *= t33;
      // side streams of 1-D layers are never written by the reference: carry the old values
      if (inside) {
#pragma unroll
        for (int q = 0; q < 2 * DSIDE; ++q) {
          const size_t o = (size_t)(DTOP + q) * Ncl + c;
          xn[o] = xo[o];
        }
      }
    } else {
      float Tm[S * S];
#pragma unroll
      for (int e = 0; e < S * S; ++e) Tm[e] = T[(size_t)e * Nc + c];
      double mx[DSIDE], my[DSIDE];  // sources that do not come from a lane of the tile: the previous sweep / the received faces
#pragma unroll
      for (int q = 0; q < DSIDE; ++q) {
        mx[q] = (offx && !sun.xinc && i + 1 == xm) ? hb.recvE[((size_t)q * Nz + k) * ym + j]
                                                   : xo[(size_t)(DTOP + q) * Ncl + (size_t)k * ncol + (size_t)j * xm + iu];
        my[q] = (offy && !sun.yinc && j + 1 == ym) ? hb.recvN[((size_t)q * Nz + k) * xm + i]
                                                   : xo[(size_t)(DTOP + DSIDE + q) * Ncl + (size_t)k * ncol + (size_t)ju * xm + i];
      }
      double out[S];
#pragma unroll
      for (int d = 0; d < S; ++d) out[d] = 0.0;
      for (int sub = 0; sub <= 2 * (TW - 1); ++sub) {  // after substep ii + jj this lane's inputs are final
        double src[S];
#pragma unroll
        for (int q = 0; q < DTOP; ++q) src[q] = top[q];
#pragma unroll
        for (int q = 0; q < DSIDE; ++q) {
          const double sx = __shfl_up(out[DTOP + q], 1), sy = __shfl_up(out[DTOP + DSIDE + q], TW);
          src[DTOP + q] = shx ? sx : mx[q];
          src[DTOP + DSIDE + q] = shy ? sy : my[q];
        }
#pragma unroll
        for (int d = 0; d < S; ++d) {
          double a = 0.0;
#pragma unroll
          for (int s2 = 0; s2 < S; ++s2) a += src[s2] * (double)Tm[d * S + s2];
          out[d] = a;
        }
      }
#pragma unroll
      for (int q = 0; q < DTOP; ++q) top[q] = out[q];
      if (inside) {
#pragma unroll
        for (int q = 0; q < DSIDE; ++q) {
          const size_t oxi = (size_t)(DTOP + q) * Ncl + (size_t)k * ncol + (size_t)j * xm + id;
          const size_t oyi = (size_t)(DTOP + DSIDE + q) * Ncl + (size_t)k * ncol + (size_t)jd * xm + i;
          if (offx && sun.xinc && i + 1 == xm) {
            hb.sendE[((size_t)q * Nz + k) * ym + j] = out[DTOP + q];  // lands on the east rank's face 0
          } else {
            const double dx_ = out[DTOP + q] - xo[oxi];
            xn[oxi] = out[DTOP + q];
            sum += dx_ * dx_;
            if (offx && !sun.xinc && i == 0) hb.sendW[((size_t)q * Nz + k) * ym + j] = out[DTOP + q];  // the west rank's face xm
          }
          if (offy && sun.yinc && j + 1 == ym) {
            hb.sendN[((size_t)q * Nz + k) * xm + i] = out[DTOP + DSIDE + q];
          } else {
            const double dy_ = out[DTOP + DSIDE + q] - xo[oyi];
            xn[oyi] = out[DTOP + DSIDE + q];
            sum += dy_ * dy_;
            if (offy && !sun.yinc && j == 0) hb.sendS[((size_t)q * Nz + k) * xm + i] = out[DTOP + DSIDE + q];
          }
        }
      }
    }
    if (inside) {
#pragma unroll
      for (int q = 0; q < DTOP; ++q) {
        const size_t o = (size_t)q * Ncl + (size_t)(k + 1) * ncol + col;
        const double old = xo[o];
        xn[o] = top[q];
        sum += (top[q] - old) * (top[q] - old);
      }
    }
  }
  if (inside) {  // side entries at the bottom level are dummies: keep
#pragma unroll
    for (int q = 0; q < 2 * DSIDE; ++q) {
      const size_t o = (size_t)(DTOP + q) * Ncl + (size_t)Nz * ncol + col;
      xn[o] = xo[o];
    }
  }
  const double r = tsx_wave_sum(sum);
  if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

// After the exchange: the faces received from the upwind rank become my face 0 (sun moving +x: recvW -> faces i = 0;
// +y: recvS -> faces j = 0); their change enters my residual like any owned entry (the reference's norm runs over
// owned entries after exchange_direct_boundary).  1-D layers carry no side streams.
template <int DTOP, int DSIDE>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_edir_unpack(TsxGeo g, TsxSun sun, const uint8_t *__restrict__ l1d,
                                                               const double *__restrict__ xo, double *__restrict__ xn,
                                                               TsxDirHalo hb, double *__restrict__ partial,
                                                               const int *__restrict__ done) {
  if (done && *done) return;
  const int xm = g.xm, ym = g.ym, Nz = g.Nz, ncol = g.ncol;
  const long long Ncl = (long long)(Nz + 1) * ncol;
  const bool ux = g.wrap_x == 0 && sun.xinc, uy = g.wrap_y == 0 && sun.yinc;
  const long long nx = ux ? (long long)DSIDE * Nz * ym : 0, ny = uy ? (long long)DSIDE * Nz * xm : 0;
  double sum[1] = {0.0};
  for (long long t = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; t < nx + ny; t += (long long)gridDim.x * TSX_BLOCK) {
    size_t o;
    double v;
    int k;
    if (t < nx) {
      const int j = (int)(t % ym);
      k = (int)((t / ym) % Nz);
      const int q = (int)(t / ((long long)ym * Nz));
      o = (size_t)(DTOP + q) * Ncl + (size_t)k * ncol + (size_t)j * xm;
      v = hb.recvW[t];
    } else {
      const long long u = t - nx;
      const int i = (int)(u % xm);
      k = (int)((u / xm) % Nz);
      const int q = (int)(u / ((long long)xm * Nz));
      o = (size_t)(DTOP + DSIDE + q) * Ncl + (size_t)k * ncol + i;
      v = hb.recvS[u];
    }
    if (l1d[k]) continue;
    const double d = v - xo[o];
    xn[o] = v;
    sum[0] += d * d;
  }
  tsx_block_reduce_store<1>(sum, partial);
}

// scalar stage of the direct iteration: residual(iter) = max(tiny, sqrt(sum)); stop rule src/pprts_explicit.F90:175-208.
// mode 1: local residual (norm2 over what this rank owns) -> sc->res; mode 2: stop rule on sc->res (after the mean over
// ranks, imp_allreduce_mean :184); mode 3: both (one rank)
struct TsxDirScalars {
  double res1, res, rtol, atol;
  int iter, maxit, done, converged;
};
__global__ __launch_bounds__(1024) void tsx_k_edir_scalar(TsxDirScalars *__restrict__ sc, const double *__restrict__ partials,
                                                          int nblocks, int mode, double scale) {
  __shared__ double sm[16];
  if (sc->done) return;
  const double tiny = 2.2250738585072014e-308;
  if (mode & 1) {
    double v = 0.0;
    for (int q = threadIdx.x; q < nblocks; q += 1024) v += partials[q];
    v = tsx_wave_sum(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int q = 0; q < 16; ++q) t += sm[q];
      const double res = sqrt(t);
      sc->res = res > tiny ? res : tiny;
    }
  }
  if (!(mode & 2) || threadIdx.x != 0) return;
  const double res = sc->res * scale;  // scale = 1/nranks after the sum over ranks
  sc->res = res;
  sc->iter += 1;
  if (sc->iter == 1) sc->res1 = res;
  const double rel = sc->res1 <= 1.4916681462400413e-154 ? 0.0 : res / sc->res1;
  if (res < sc->atol || rel < sc->rtol) {
    sc->converged = 1;
    sc->done = 1;
  } else if (sc->iter >= sc->maxit) {
    sc->done = 1;
  }
}

// ---- K7: source term in dst-owned storage.  Solar: set_solar_source (src/pprts.F90:4684-4846): every contribution of
//      cell (k,i,j) lands on a stream leaving that cell, i.e. on internal index c -- no halo reduce needed.
template <int NTOP, int NSIDE, int DTOP, int DSIDE>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_setup_b_solar(TsxGeo g, TsxSun sun, const float *__restrict__ Sd,
                                                                 const uint8_t *__restrict__ l1d, const double *__restrict__ a13,
                                                                 const double *__restrict__ a23, const double *__restrict__ albedo,
                                                                 const double *__restrict__ E, double *__restrict__ b,
                                                                 TsxDirHalo hb) {
  constexpr int D = NTOP + 2 * NSIDE, S = DTOP + 2 * DSIDE;
  const bool gx = g.wrap_x == 0 && !sun.xinc, gy = g.wrap_y == 0 && !sun.yinc;  // upwind face of the last cell is off-rank
  const int xm = g.xm, ym = g.ym, Nz = g.Nz, ncol = g.ncol;
  const long long Nc = g.Nc, Ncl = (long long)(Nz + 1) * ncol;
  const double streams = (double)(NTOP / 2);
  double *__restrict__ bt = b + (size_t)D * Nc;
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const int i = (int)(c % xm);
    const long long t = c / xm;
    const int j = (int)(t % ym);
    const int k = (int)(t / ym);
    const int col = j * xm + i;
    const int iu = sun.xinc ? i : (i + 1 == xm ? 0 : i + 1);
    const int ju = sun.yinc ? j : (j + 1 == ym ? 0 : j + 1);
    double own[S], src[S];
    bool any = false;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      own[s] = E[(size_t)s * Ncl + (size_t)k * ncol + col];
      any |= own[s] > 2.220446049250313e-16;  // epsilon(one), :4706
    }
#pragma unroll
    for (int q = 0; q < DTOP; ++q) src[q] = own[q];
#pragma unroll
    for (int q = 0; q < DSIDE; ++q) {
      src[DTOP + q] = (gx && i + 1 == xm) ? hb.recvE[((size_t)q * Nz + k) * ym + j]
                                          : E[(size_t)(DTOP + q) * Ncl + (size_t)k * ncol + (size_t)j * xm + iu];
      src[DTOP + DSIDE + q] = (gy && j + 1 == ym) ? hb.recvN[((size_t)q * Nz + k) * xm + i]
                                                  : E[(size_t)(DTOP + DSIDE + q) * Ncl + (size_t)k * ncol + (size_t)ju * xm + i];
    }
    double out[D];
#pragma unroll
    for (int d = 0; d < D; ++d) out[d] = 0.0;
    if (any) {
      if (l1d[k]) {
        const double t13 = a13[c], t23 = a23[c];
#pragma unroll
        for (int s = 0; s < DTOP; ++s)
#pragma unroll
          for (int q = 0; q < NTOP; ++q) out[q] += own[s] * (tsx_inward(q) ? t23 : t13) / streams;
      } else {
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
          for (int s = 0; s < S; ++s) out[d] += src[s] * (double)Sd[(size_t)(d * S + s) * Nc + c];
        }
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) b[(size_t)d * Nc + c] = out[d];
    if (k == Nz - 1) {  // tail rows: surface albedo reflecting the direct beam (:4829-4843); TOA and dummies carry no source
      double esrf = 0.0;
#pragma unroll
      for (int q = 0; q < DTOP; ++q) esrf += E[(size_t)q * Ncl + (size_t)Nz * ncol + col];
#pragma unroll
      for (int d = 0; d < D; ++d)
        bt[(size_t)d * ncol + col] = (d < NTOP && !tsx_inward(d)) ? esrf * albedo[col] / streams : 0.0;
    }
  }
}

// B_eff (src/schwarzschild.F90:36-67): 2-point Gauss-Legendre on (0,1)
__device__ __forceinline__ double tsx_B_eff(double B_far, double B_near, double tau) {
  const double pt[2] = {0.5 - 0.5 / 1.7320508075688772, 0.5 + 0.5 / 1.7320508075688772};
  double B = 0.0;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const double mu = pt[q];
    const double dtau = tau / mu;
    double bmu;
    if (dtau < 1e-3) {
      bmu = (B_far + B_near) * .5;
    } else {
      const double tm1 = expm1(-dtau);
      bmu = (-B_near + B_far * (tm1 + 1)) / (tm1) + ((B_far - B_near) * mu) / tau;
    }
    B += bmu * mu * 0.5;
  }
  return B * 2;
}

// Thermal: set_thermal_source (src/pprts.F90:4848-4987); planck at levels, reference layout (k over L fastest); bsrfc (xm, ym) = atm%Bsrfc or null
template <int NTOP, int NSIDE, typename CT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_setup_b_thermal(TsxGeo g, const CT *__restrict__ C,
                                                                   const uint8_t *__restrict__ l1d, const double *__restrict__ a11,
                                                                   const double *__restrict__ a12, const double *__restrict__ albedo,
                                                                   const double *__restrict__ planck, const double *__restrict__ bsrfc,
                                                                   const double *__restrict__ kabs,
                                                                   const double *__restrict__ dz, double dx, double dy,
                                                                   double *__restrict__ b, const double *__restrict__ colsum,
                                                                   const int *__restrict__ cidx, long long nent) {
  // colsum (nullable): sum over dst of c(src, :) per distinct block [D][nent] behind the per-cell index cidx (tsx_k_dd_colsum)
  constexpr int D = NTOP + 2 * NSIDE;
  const int xm = g.xm, ym = g.ym, Nz = g.Nz, ncol = g.ncol, L = Nz + 1;
  const long long Nc = g.Nc;
  const double pi = 3.14159265358979323846;
  const double tstreams = (double)(NTOP / 2), sstreams = (double)(NSIDE / 2);
  const double Az = dx * dy;
  double *__restrict__ bt = b + (size_t)D * Nc;
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const int i = (int)(c % xm);
    const long long t = c / xm;
    const int j = (int)(t % ym);
    const int k = (int)(t / ym);
    const int col = j * xm + i;
    const size_t r = (size_t)k + (size_t)Nz * ((size_t)i + (size_t)xm * j);
    const size_t rl = (size_t)k + (size_t)L * ((size_t)i + (size_t)xm * j);
    const double b0 = planck[rl], b1 = planck[rl + 1];
    const double dzz = dz[r];
    const double tauz = kabs[r] * dzz;
    const double btop = tsx_B_eff(b1, b0, tauz), bbot = tsx_B_eff(b0, b1, tauz);
    if (l1d[k]) {
      const double bfac = pi * Az / tstreams;
      double emis = 1.0 - a11[c] - a12[c];
      emis = fmax(0.0, fmin(1.0, emis));
#pragma unroll
      for (int q = 0; q < NTOP; ++q) b[(size_t)q * Nc + c] = (tsx_inward(q) ? bbot : btop) * bfac * emis;
#pragma unroll
      for (int d = NTOP; d < D; ++d) b[(size_t)d * Nc + c] = 0.0;
    } else {
      const double Ax = dy * dzz, Ay = dx * dzz;
#pragma unroll
      for (int s = 0; s < D; ++s) {
        double sum = 0.0;
        if (colsum) {
          sum = colsum[(size_t)s * nent + cidx[c]];
        } else {
#pragma unroll
          for (int d = 0; d < D; ++d) sum += (double)C[(size_t)(d * D + s) * Nc + c];
        }
        double emis = fmax(0.0, fmin(1.0, 1.0 - sum));
        double v;
        if (s < NTOP) {
          v = (tsx_inward(s) ? bbot : btop) * (pi * Az / tstreams) * emis;
        } else {
          const int q = (s - NTOP) % NSIDE;
          const double area = s < NTOP + NSIDE ? Ax : Ay;
          v = ((q + 1 > NSIDE / 2) ? btop : bbot) * emis * (pi * area / sstreams);
        }
        b[(size_t)s * Nc + c] = v;  // every stream's emission lands on the face it leaves through == index c
      }
    }
    if (k == Nz - 1) {
      // surface emission (src/pprts.F90:4958-4985): atm%Bsrfc with the emissivity 1 - albedo clamped to [0, 1] where the caller
      // gave planck_srfc (:4960-4970), else planck at the lowest level with 1 - albedo as it is (:4971-4984)
      const double srf = bsrfc ? bsrfc[col] * Az * fmax(0.0, fmin(1.0, 1.0 - albedo[col])) * pi / tstreams
                               : planck[(size_t)Nz + (size_t)L * ((size_t)i + (size_t)xm * j)] * Az * (1.0 - albedo[col]) * pi / tstreams;
#pragma unroll
      for (int d = 0; d < D; ++d) bt[(size_t)d * ncol + col] = (d < NTOP && !tsx_inward(d)) ? srf : 0.0;
    }
  }
}

// ---- K9: absorption by coefficient divergence (calc_flx_div, src/pprts.F90:5286-5398) / volume (:5477, 5483-5503).
//      ediff in internal storage (W), edir planes (W).  abso out: reference layout (k fastest), W/m3.
template <int NTOP, int NSIDE, int DTOP, int DSIDE, typename CT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_flx_div(TsxGeo g, TsxSun sun, int lsolar, int lthermal,
                                                           const CT *__restrict__ C, const float *__restrict__ T,
                                                           const float *__restrict__ Sd, const uint8_t *__restrict__ l1d,
                                                           const double *__restrict__ a11, const double *__restrict__ a12,
                                                           const double *__restrict__ kabs, const double *__restrict__ dz,
                                                           double dx, double dy, const double *__restrict__ E,
                                                           const double *__restrict__ x, const double *__restrict__ bsrc,
                                                           double *__restrict__ abso, TsxDirHalo hb,
                                                           const double *__restrict__ hW, const double *__restrict__ hE,
                                                           const double *__restrict__ hS, const double *__restrict__ hN,
                                                           const double *__restrict__ colsum, const int *__restrict__ cidx,
                                                           long long nent) {
  // colsum (nullable): sum over dst of c(src, :) per distinct block [D][nent] behind the per-cell index cidx (tsx_k_dd_colsum)
  // hb: direct-beam faces from the upwind ranks; hW..hN: the diffuse halo of x (entering side streams, as the operator
  // reads them: [slot][k][j] / [slot][k][i]); only dereferenced where the rank does not wrap onto itself
  constexpr int D = NTOP + 2 * NSIDE, S = DTOP + 2 * DSIDE;
  const bool offx = g.wrap_x == 0, offy = g.wrap_y == 0;
  const int xm = g.xm, ym = g.ym, Nz = g.Nz, ncol = g.ncol;
  const long long Nc = g.Nc, Ncl = (long long)(Nz + 1) * ncol;
  const double *__restrict__ xt = x + (size_t)D * Nc;
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const int i = (int)(c % xm);
    const long long t = c / xm;
    const int j = (int)(t % ym);
    const int k = (int)(t / ym);
    const int col = j * xm + i;
    const size_t r = (size_t)k + (size_t)Nz * ((size_t)i + (size_t)xm * j);
    const bool is1d = l1d[k] != 0;
    double a = 0.0;
    if (lsolar) {
      const int iu = sun.xinc ? i : (i + 1 == xm ? 0 : i + 1);
      const int ju = sun.yinc ? j : (j + 1 == ym ? 0 : j + 1);
      if (is1d) {
        const double cdiv = kabs[r] * dz[r] / sun.costheta;
#pragma unroll
        for (int q = 0; q < DTOP; ++q) a += E[(size_t)q * Ncl + (size_t)k * ncol + col] * (-expm1(-cdiv));
      } else {
        double src[S];
#pragma unroll
        for (int q = 0; q < DTOP; ++q) src[q] = E[(size_t)q * Ncl + (size_t)k * ncol + col];
#pragma unroll
        for (int q = 0; q < DSIDE; ++q) {
          src[DTOP + q] = (offx && !sun.xinc && i + 1 == xm) ? hb.recvE[((size_t)q * Nz + k) * ym + j]
                                                             : E[(size_t)(DTOP + q) * Ncl + (size_t)k * ncol + (size_t)j * xm + iu];
          src[DTOP + DSIDE + q] = (offy && !sun.yinc && j + 1 == ym)
                                      ? hb.recvN[((size_t)q * Nz + k) * xm + i]
                                      : E[(size_t)(DTOP + DSIDE + q) * Ncl + (size_t)k * ncol + (size_t)ju * xm + i];
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
          double s1 = 0.0, s2 = 0.0;
#pragma unroll
          for (int d = 0; d < S; ++d) s1 += (double)T[(size_t)(d * S + s) * Nc + c];
#pragma unroll
          for (int d = 0; d < D; ++d) s2 += (double)Sd[(size_t)(d * S + s) * Nc + c];
          a += src[s] * (1.0 - s1 - s2);
        }
      }
    }
    // diffuse sources of the cell (same gather as the operator)
    double xs[D];
#pragma unroll
    for (int q = 0; q < NTOP; ++q) {
      if (tsx_inward(q)) xs[q] = k > 0 ? x[(size_t)q * Nc + c - ncol] : xt[(size_t)q * ncol + col];
      else xs[q] = k + 1 < Nz ? x[(size_t)q * Nc + c + ncol] : xt[(size_t)q * ncol + col];
    }
    if (!is1d) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) {
        const int d = NTOP + q, slot = q >> 1;
        if (tsx_inward(q)) xs[d] = (offx && i == 0) ? hW[((size_t)slot * Nz + k) * ym + j] : x[(size_t)d * Nc + c + (i > 0 ? -1 : xm - 1)];
        else xs[d] = (offx && i == xm - 1) ? hE[((size_t)slot * Nz + k) * ym + j] : x[(size_t)d * Nc + c + (i < xm - 1 ? 1 : -(xm - 1))];
      }
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) {
        const int d = NTOP + NSIDE + q, slot = q >> 1;
        if (tsx_inward(q))
          xs[d] = (offy && j == 0) ? hS[((size_t)slot * Nz + k) * xm + i]
                                   : x[(size_t)d * Nc + c + (j > 0 ? -(long long)xm : (long long)(ym - 1) * xm)];
        else
          xs[d] = (offy && j == ym - 1) ? hN[((size_t)slot * Nz + k) * xm + i]
                                        : x[(size_t)d * Nc + c + (j < ym - 1 ? (long long)xm : -(long long)(ym - 1) * xm)];
      }
#pragma unroll
      for (int s = 0; s < D; ++s) {
        double sum = 0.0;
        if (colsum) {
          sum = colsum[(size_t)s * nent + cidx[c]];
        } else {
#pragma unroll
          for (int d = 0; d < D; ++d) sum += (double)C[(size_t)(d * D + s) * Nc + c];
        }
        a += xs[s] * (1.0 - sum);
      }
    } else {
      const double cdiv = fmax(0.0, 1.0 - a11[c] - a12[c]);
#pragma unroll
      for (int q = 0; q < NTOP; ++q) a += xs[q] * cdiv;
    }
    if (lthermal) {  // minus the emitted source on every stream leaving the cell (:5373-5398) == b at index c
#pragma unroll
      for (int d = 0; d < D; ++d) a -= bsrc[(size_t)d * Nc + c];
    }
    abso[r] = a * (1.0 / (dx * dy * dz[r]));
  }
}

// ---- K8 + K12: W -> W/m2 (gen_scale_*_flx_vec_arr, src/pprts.F90:3901-3987) and pprts_get_result (:5850-5888).
//      Outputs in the reference layout (level fastest): redir/redn/reup (L, xm, ym), rabso (Nz, xm, ym) in place.
//      The solver's fields are column-fastest, the results level-fastest: a tile of 32 columns x 32 levels goes through LDS, read
//      along the columns and written along the levels (round 3: the direct version wrote 8 bytes per 520: 379 us per call on
//      256 x 256 x 64, with the transpose 55 us).  Grid: (ceil(ncol / 32), ceil(L / 32)), 256 threads.
template <int NTOP, int NSIDE, int DTOP, int DSIDE>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_get_result(TsxGeo g, TsxSun sun, int lsolar, double dx, double dy,
                                                              int top_div, const double *__restrict__ E,
                                                              const double *__restrict__ x, double *__restrict__ redir,
                                                              double *__restrict__ redn, double *__restrict__ reup,
                                                              double *__restrict__ rabso) {
  constexpr int D = NTOP + 2 * NSIDE, TC = 32, TK = 32;
  static_assert(TSX_BLOCK == 256, "tile loops assume 256 threads");
  __shared__ double sdn[TK][TC + 1], sup[TK][TC + 1], sdi[TK][TC + 1];
  const int xm = g.xm, Nz = g.Nz, ncol = g.ncol, L = Nz + 1;
  const long long Nc = g.Nc, Ncl = (long long)L * ncol;
  const double mu = lsolar ? sun.mu : 1.0;
  const double invA = 1.0 / (dx * dy);  // difftop%area_divider = 1
  const double *__restrict__ xt = x + (size_t)D * Nc;
  const int c0 = blockIdx.x * TC, k0 = blockIdx.y * TK;
  // ---- read: lanes along the columns
  for (int e = threadIdx.x; e < TC * TK; e += TSX_BLOCK) {
    const int cc = e % TC, kk = e / TC;
    const int col = c0 + cc, k = k0 + kk;
    double dn = 0.0, up = 0.0, di = 0.0;
    if (col < ncol && k < L) {
#pragma unroll
      for (int d = 0; d < NTOP; ++d) {
        double v;  // stream d at level k
        if (tsx_inward(d)) v = k >= 1 ? x[(size_t)d * Nc + (size_t)(k - 1) * ncol + col] : xt[(size_t)d * ncol + col];
        else v = k < Nz ? x[(size_t)d * Nc + (size_t)k * ncol + col] : xt[(size_t)d * ncol + col];
        if (tsx_inward(d)) dn += v * invA;
        else up += v * invA;
      }
      if (redir && lsolar) {
#pragma unroll
        for (int s = 0; s < DTOP; ++s) di += E[(size_t)s * Ncl + (size_t)k * ncol + col] * (1.0 / (dx * dy / (double)top_div));
        di = di / (double)top_div * mu;
      }
    }
    sdn[kk][cc] = dn * mu;
    sup[kk][cc] = up * mu;
    sdi[kk][cc] = di;
  }
  __syncthreads();
  // ---- write: lanes along the levels
  for (int e = threadIdx.x; e < TC * TK; e += TSX_BLOCK) {
    const int kk = e % TK, cc = e / TK;
    const int col = c0 + cc, k = k0 + kk;
    if (col >= ncol || k >= L) continue;
    const int i = col % xm, j = col / xm;
    const size_t o = (size_t)k + (size_t)L * ((size_t)i + (size_t)xm * j);
    redn[o] = sdn[kk][cc];
    reup[o] = sup[kk][cc];
    if (redir) redir[o] = sdi[kk][cc];
    if (k < Nz) {
      const size_t r = (size_t)k + (size_t)Nz * ((size_t)i + (size_t)xm * j);
      rabso[r] *= mu;
    }
  }
}

// scalar field in reference layout (k fastest) -> cell-indexed (i fastest) is tsx_k_import_cellfield (tsx_kernels.hpp)

// ------------------------------------------------------------------------------------------------
// set_optical_properties on the device (src/pprts.F90:1764-2000): delta scaling, 1-D layer detection, Eddington
// coefficients of the 1-D layers.  Optical property fields are in the reference layout (z fastest).

// delta_scale with f = g**2 (src/helper_functions.fypp:1622-1666), in place
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_delta_scale(long long n, double *__restrict__ kabs, double *__restrict__ ksca,
                                                               double *__restrict__ g) {
  const double eps = 2.220446049250313e-16;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * TSX_BLOCK) {
    const double ka = kabs[q], ks = ksca[q], gq = g[q];
    double tau = ka + ks;
    if (tau < eps) continue;
    double w0 = ks / tau, gn;
    if (gq >= 1.0 - 10.0 * eps) {  // pure forward peak: everything scattered stays in the beam
      tau *= 1.0 - w0;
      w0 = 0.0;
      gn = 0.0;
    } else {
      const double f = gq * gq;
      tau *= 1.0 - w0 * f;
      gn = (gq - f) / (1.0 - f);
      w0 = w0 * (1.0 - f) / (1.0 - f * w0);
    }
    g[q] = gn;
    kabs[q] = tau * (1.0 - w0);
    ksca[q] = tau * w0;
  }
}

// flags[k] = 1 if dz/dx > ratio anywhere in layer k (src/pprts.F90:669-677; the "and every layer above" part is host logic)
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_flag_1d(long long n, int Nz, const double *__restrict__ dz, double dx, double ratio,
                                                           int *__restrict__ flags) {
  // a thick layer is thick in (nearly) every column: one global atomic per cell on one word took 0.75 ms, a look before the
  // atomic 0.18 ms (thousands of lanes polling the same 16 words); flags per workgroup in LDS, then at most Nz atomics each
  constexpr int KMAX = 512;
  __shared__ int sf[KMAX];
  const bool lds = Nz <= KMAX;
  if (lds) {
    for (int k = threadIdx.x; k < Nz; k += TSX_BLOCK) sf[k] = 0;
    __syncthreads();
  }
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * TSX_BLOCK)
    if (dz[q] / dx > ratio) {
      const int k = (int)(q % Nz);
      if (lds) {
        if (sf[k] == 0) sf[k] = 1;  // benign race: every writer stores 1
      } else {
        int *f = &flags[k];
        if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(f, 1);
      }
    }
  if (lds) {
    __syncthreads();
    for (int k = threadIdx.x; k < Nz; k += TSX_BLOCK)
      if (sf[k] && __hip_atomic_load(&flags[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(&flags[k], 1);
  }
}

// eddington_coeff_ec (src/eddington.F90:173-241) for the cells of 1-D layers; inputs in the reference layout, outputs
// in cell order.  a11 = t, a12 = r, a13 = rdir, a23 = sdir, a33 = tdir.
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_eddington(TsxGeo g, const double *__restrict__ kabs, const double *__restrict__ ksca,
                                                             const double *__restrict__ gas, const double *__restrict__ dz, double mu0,
                                                             const uint8_t *__restrict__ l1d, double *__restrict__ a11,
                                                             double *__restrict__ a12, double *__restrict__ a13,
                                                             double *__restrict__ a23, double *__restrict__ a33) {
  const int xm = g.xm, ym = g.ym, Nz = g.Nz;
  const double eps = 2.220446049250313e-16, tiny = 2.2250738585072014e-308;
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < g.Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const int i = (int)(c % xm);
    const long long t_ = c / xm;
    const int j = (int)(t_ % ym), k = (int)(t_ / ym);
    if (!l1d[k]) continue;
    const size_t r = (size_t)k + (size_t)Nz * ((size_t)i + (size_t)xm * j);
    const double ext = fmax(tiny, kabs[r] + ksca[r]);
    const double dtau = dz[r] * ext, w0 = ksca[r] / ext, gg = gas[r];
    const double f = 0.75 * gg;
    const double g1 = 2.0 - w0 * (1.25 + f), g2 = w0 * (0.75 - f), g3 = 0.5 - mu0 * f;
    const double slant = fmax(dtau / fmax(sqrt(tiny), mu0), 0.0);
    double tt, rr, rdir, sdir, tdir;
    if (slant > 1e-6) {
      const double g4 = 1.0 - g3;
      const double al1 = g1 * g4 + g2 * g3, al2 = g1 * g3 + g2 * g4;
      const double A = sqrt(fmax((g1 - g2) * (g1 + g2), 1e-12));
      double kmu = A * mu0;
      if (kmu <= 1.0 + 10.0 * eps && kmu >= 1.0 - 10.0 * eps) kmu = 1.0 - 10.0 * eps;  // approx(), helper_functions.fypp:1272
      const double kg3 = A * g3, kg4 = A * g4;
      const double e0 = exp(-slant), e = exp(-A * dtau), e2 = e * e, k2e = 2.0 * A * e;
      double beta = 1.0 / (A + g1 + (A - g1) * e2);
      rr = g2 * (1.0 - e2) * beta;
      tt = k2e * beta;
      beta = w0 * beta / (1.0 - kmu * kmu);
      sdir = beta * (k2e * (g4 + al1 * mu0) - e0 * ((1.0 + kmu) * (al1 + kg4) - (1.0 - kmu) * (al1 - kg4) * e2));
      rdir = beta * ((1.0 - kmu) * (al2 + kg3) - (1.0 + kmu) * (al2 - kg3) * e2 - k2e * (g3 - al2 * mu0) * e0);
      tdir = e0;
    } else {
      tt = 1.0 - g1 * dtau;
      rr = g2 * dtau;
      sdir = (1.0 - g3) * (w0 * dtau);
      rdir = g3 * (w0 * dtau);
      tdir = 1.0 - slant;
    }
    a11[c] = tt;
    a12[c] = rr;
    a13[c] = rdir;
    a23[c] = sdir;
    a33[c] = tdir;
  }
}
