// tsx_kernels.hpp -- hand-written HIP kernels for gfx950 (MI355X): 64-wide wavefronts, HBM-bound.
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

// part: 0 = every cell; 1 = interior only (cells whose gather touches no received face: launched while the exchange is
// in flight); 2 = frame only (the complement, enumerated directly: per level the first/last row and the first/last
// group of every other row).  Partial sums of launch 2 go behind those of launch 1 (partials pointer is offset).
// HALO: some face of the rank is not a periodic self-neighbour (edge threads then read the received face buffers);
// HAS1D: some layer is 1-D.  Both are kernel-uniform and compiled out in the common case.  The gather is branch-free
// (offset / pointer selects, unconditional loads): a conditional load ends a basic block and forces an s_waitcnt.
template <int NTOP, int NSIDE, typename CT, int FUSE, int CPT, typename XT, typename WT, bool HALO, bool HAS1D>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_spmv_w(
    TsxGeo g, const CT *__restrict__ C, const uint8_t *__restrict__ l1d, const double *__restrict__ a11,
    const double *__restrict__ a12, const double *__restrict__ albedo, const XT *__restrict__ x,
    double *__restrict__ y, const double *__restrict__ hW, const double *__restrict__ hE,
    const double *__restrict__ hS, const double *__restrict__ hN, const WT *__restrict__ w,
    double *__restrict__ partials, const int *__restrict__ done, int part) {
  constexpr int D = NTOP + 2 * NSIDE;
  using V = TsxVec<CPT>;
  if (done && *done) return;
  double sum[3] = {0.0, 0.0, 0.0};
  const long long Nc = g.Nc;
  const int xm = g.xm, ym = g.ym, Nz = g.Nz, ncol = g.ncol;
  const bool wrapx = g.wrap_x != 0, wrapy = g.wrap_y != 0;
  const int gx = xm / CPT;                                   // groups per row
  const int nfull = wrapy ? 0 : (ym >= 2 ? 2 : 1);           // frame: rows that belong to it entirely
  const int ex = wrapx ? 0 : (gx >= 2 ? 2 : 1);              // frame: groups of every other row
  const int nframe = nfull * gx + (ym - nfull) * ex;         // frame groups per level
  const long long ngroups = part == 2 ? (long long)Nz * nframe : Nc / CPT;
  const long long nchunks = (ngroups + TSX_BLOCK - 1) / TSX_BLOCK;
  const XT *__restrict__ xt = x + (size_t)D * Nc;
  double *__restrict__ yt = y + (size_t)D * Nc;
  const WT *__restrict__ wt = (FUSE & 1) ? w + (size_t)D * Nc : nullptr;

  for (long long base = 0; base < nchunks; base += gridDim.x) {
    const long long nb = (nchunks - base) < (long long)gridDim.x ? (nchunks - base) : (long long)gridDim.x;
    if ((long long)blockIdx.x >= nb) break;
    const long long grp = (base + tsx_swizzle(blockIdx.x, nb)) * TSX_BLOCK + threadIdx.x;
    if (grp >= ngroups) continue;
    long long c;
    int i, j, k;
    if (part == 2) {
      k = (int)(grp / nframe);
      const int f = (int)(grp - (long long)k * nframe);
      int ig;
      if (f < nfull * gx) {
        j = (f / gx) == 0 ? 0 : ym - 1;
        ig = f % gx;
      } else {
        const int f2 = f - nfull * gx, exs = ex > 0 ? ex : 1;
        j = f2 / exs + (wrapy ? 0 : 1);
        ig = (f2 % exs) == 0 ? 0 : gx - 1;
      }
      i = ig * CPT;
      c = ((long long)k * ym + j) * xm + i;
    } else {
      c = grp * CPT;
      i = (int)(c % xm);
      const long long t = c / xm;
      j = (int)(t % ym);
      k = (int)(t / ym);
      if (part == 1) {
        const bool fr = (!wrapx && (i == 0 || i + CPT >= xm)) || (!wrapy && (j == 0 || j + 1 >= ym));
        if (fr) continue;
      }
    }
    const int col = j * xm + i;

    double xs[D][CPT];
    // ---- gather the D source streams of the CPT cells (all loads unconditional)
#pragma unroll
    for (int q = 0; q < NTOP; ++q) {
      const bool tail = tsx_inward(q) ? (k == 0) : (k + 1 >= Nz);
      const XT *p = tail ? xt + (size_t)q * ncol + col
                         : x + (size_t)q * Nc + c + (tsx_inward(q) ? -(long long)ncol : (long long)ncol);
      V::ld(p, xs[q]);
    }
#pragma unroll
    for (int q = 0; q < NSIDE; ++q) {
      const int d = NTOP + q, slot = q >> 1;
      double own[CPT];
      V::ld(x + (size_t)d * Nc + c, own);
      if (tsx_inward(q)) {  // +x stream: leaves the cell to the west
        const bool edge = i == 0;
        const long long off = edge ? (wrapx ? (long long)(xm - 1) : 0) : -1;
        double e = (double)x[(size_t)d * Nc + c + off];
        if (HALO) {
          const double h = hW[((size_t)slot * Nz + k) * ym + j];
          e = (edge && !wrapx) ? h : e;
        }
        xs[d][0] = e;
#pragma unroll
        for (int m = 1; m < CPT; ++m) xs[d][m] = own[m - 1];
      } else {  // -x stream: leaves the cell to the east
        const bool edge = i + CPT >= xm;
        const long long off = edge ? (wrapx ? (long long)CPT - xm : 0) : CPT;
        double e = (double)x[(size_t)d * Nc + c + off];
        if (HALO) {
          const double h = hE[((size_t)slot * Nz + k) * ym + j];
          e = (edge && !wrapx) ? h : e;
        }
        xs[d][CPT - 1] = e;
#pragma unroll
        for (int m = 0; m < CPT - 1; ++m) xs[d][m] = own[m + 1];
      }
    }
#pragma unroll
    for (int q = 0; q < NSIDE; ++q) {
      const int d = NTOP + NSIDE + q, slot = q >> 1;
      const bool edge = tsx_inward(q) ? (j == 0) : (j + 1 >= ym);
      const long long wrapoff = tsx_inward(q) ? (long long)(ym - 1) * xm : -(long long)(ym - 1) * xm;
      const long long off = edge ? (wrapy ? wrapoff : 0) : (tsx_inward(q) ? -(long long)xm : (long long)xm);
      V::ld(x + (size_t)d * Nc + c + off, xs[d]);
      if (HALO) {
        double h[CPT];
        V::ld((tsx_inward(q) ? hS : hN) + ((size_t)slot * Nz + k) * xm + i, h);
#pragma unroll
        for (int m = 0; m < CPT; ++m) xs[d][m] = (edge && !wrapy) ? h[m] : xs[d][m];
      }
    }

    bool is1d = false;
    double t11[CPT], t12[CPT];
    if (HAS1D) {
      is1d = l1d[k] != 0;
      V::ld(a11 + c, t11);
      V::ld(a12 + c, t12);
    }
    double down[CPT];
#pragma unroll
    for (int m = 0; m < CPT; ++m) down[m] = 0.0;
    // ---- one destination stream (coefficient row) at a time, software-pipelined: the D coefficient loads of row d+1
    // (plus its diagonal / w operands) are issued before the FMAs of row d.  Without the explicit staging hipcc
    // serialises load -> wait -> fma per coefficient (one memory latency each).
    using CV = typename TsxRaw<CT, CPT>::type;
    using XV = typename TsxRaw<XT, CPT>::type;
    using WV = typename TsxRaw<WT, CPT>::type;
    CV cfc[D], cfn[D];
    XV xoc, xon;
    WV wc, wn;
    auto issue_row = [&](int d, CV(&cf)[D], XV &xo_, WV &w_) {
#pragma unroll
      for (int s2 = 0; s2 < D; ++s2) cf[s2] = TsxRaw<CT, CPT>::ld(C + (size_t)(d * D + s2) * Nc + c);
      xo_ = TsxRaw<XT, CPT>::ld(x + (size_t)d * Nc + c);
      if (FUSE & 1) w_ = TsxRaw<WT, CPT>::ld(w + (size_t)d * Nc + c);
    };
    issue_row(0, cfc, xoc, wc);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (d + 1 < D) issue_row(d + 1, cfn, xon, wn);
      __builtin_amdgcn_sched_barrier(0);
      double xo[CPT], acc[CPT];
      TsxRaw<XT, CPT>::cvt(xoc, xo);
      if (HAS1D && is1d) {
#pragma unroll
        for (int m = 0; m < CPT; ++m)
          acc[m] = d < NTOP ? xo[m] - t11[m] * xs[d][m] - t12[m] * xs[d < NTOP ? (d ^ 1) : d][m] : xo[m];
      } else {
#pragma unroll
        for (int m = 0; m < CPT; ++m) acc[m] = 0.0;
#pragma unroll
        for (int s2 = 0; s2 < D; ++s2) {
          double cf[CPT];
          TsxRaw<CT, CPT>::cvt(cfc[s2], cf);
#pragma unroll
          for (int m = 0; m < CPT; ++m) acc[m] += cf[m] * xs[s2][m];
        }
#pragma unroll
        for (int m = 0; m < CPT; ++m) acc[m] = xo[m] - acc[m];
      }
      V::st(y + (size_t)d * Nc + c, acc);
      if (d < NTOP && tsx_inward(d)) {
#pragma unroll
        for (int m = 0; m < CPT; ++m) down[m] += xo[m];
      }
      if (FUSE & 1) {
        double wv[CPT];
        TsxRaw<WT, CPT>::cvt(wc, wv);
#pragma unroll
        for (int m = 0; m < CPT; ++m) sum[0] += wv[m] * acc[m];
      }
      if (FUSE & 2) {
#pragma unroll
        for (int m = 0; m < CPT; ++m) sum[1] += xo[m] * acc[m];
      }
      if (FUSE & 4) {
#pragma unroll
        for (int m = 0; m < CPT; ++m) sum[2] += acc[m] * acc[m];
      }
      if (d + 1 < D) {
#pragma unroll
        for (int s2 = 0; s2 < D; ++s2) cfc[s2] = cfn[s2];
        xoc = xon;
        wc = wn;
      }
    }
    if (k == Nz - 1) {  // rows no cell writes: TOA Edn, surface Eup (albedo), bottom side dummies
      double alb[CPT];
      V::ld(albedo + col, alb);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        double xv[CPT], yv[CPT];
        V::ld(xt + (size_t)d * ncol + col, xv);
#pragma unroll
        for (int m = 0; m < CPT; ++m)
          yv[m] = (d < NTOP && !tsx_inward(d)) ? xv[m] - alb[m] / (double)(NTOP / 2) * down[m] : xv[m];
        V::st(yt + (size_t)d * ncol + col, yv);
        if (FUSE & 1) {
          double wv[CPT];
          V::ld(wt + (size_t)d * ncol + col, wv);
#pragma unroll
          for (int m = 0; m < CPT; ++m) sum[0] += wv[m] * yv[m];
        }
        if (FUSE & 2) {
#pragma unroll
          for (int m = 0; m < CPT; ++m) sum[1] += xv[m] * yv[m];
        }
        if (FUSE & 4) {
#pragma unroll
          for (int m = 0; m < CPT; ++m) sum[2] += yv[m] * yv[m];
        }
      }
    }
  }
  if (FUSE) tsx_block_reduce_store<3>(sum, partials);
}

// ------------------------------------------------------------------------------------------------
// Column preconditioner  z = M^-1 r,  M = the column-diagonal blocks of A in dst-owned storage.
// Inside one column only the top streams couple vertically (a cell's in-column sources are Eup(k+1) and
// Edn(k)); the side streams leaving the column depend on those but nothing in the column depends on them.
// So M^-1 is an exact two-stream (adding-method) solve per column followed by a substitution for the side
// streams.  This is the GPU-native counterpart of the reference's ILU(0) in z-fastest ordering
// (src/pprts.F90:4350-4371): ILU captures the strong vertical coupling approximately, this captures it
// exactly, and every column is independent (no triangular-solve dependency across the domain).
//   H = NTOP/2 up/down pairs.  With U_k (up, level k), V_k (down, level k):
//     U_k     = ru_k     + Tuu U_{k+1} + Rud V_k
//     V_{k+1} = rd_{k+1} + Rdu U_{k+1} + Tdd V_k ,   V_0 = rd_0 ,  U_Nz = ru_Nz + Alb V_Nz
//   upward sweep:   U_k = A_k V_k + B_k  (A_Nz = Alb, B_Nz = ru_Nz), stores per cell Gw, GT, A_k, B_k with
//                   G = (I - Rdu A_{k+1})^-1, Gw = G (rd_{k+1} + Rdu B_{k+1}), GT = G Tdd
//   downward sweep: V_{k+1} = Gw + GT V_k ; U_k = A_k V_k + B_k ; side dst = r + c(up->d) U_{k+1} + c(dn->d) V_k
// One thread per column, lanes along x: every plane access is a coalesced 256/512-byte span.
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

// temp planes per cell: [Gw: H][GT: H*H][A: H*H][B: H]
template <int NTOP>
__host__ __device__ constexpr int tsx_pc_ntmp() { return (NTOP / 2) * 2 * ((NTOP / 2) + 1); }

// ROWS: 0 = every row; 1 / 2 = only rows with even / odd j (zebra line ordering).  GS: the right-hand side is
// r + N_y z, the contribution of the +-y side streams of the neighbouring rows held in z (line Gauss-Seidel in y:
// rows of one colour only see rows of the other colour, so all columns of a pass stay independent).
template <int NTOP, int NSIDE, typename CT, int ROWS, bool GS, typename ZT>
__global__ __launch_bounds__(64) void tsx_k_pc_column(TsxGeo g, const CT *__restrict__ C, const uint8_t *__restrict__ l1d,
                                                      const double *__restrict__ a11, const double *__restrict__ a12,
                                                      const double *__restrict__ albedo, const double *__restrict__ r,
                                                      ZT *__restrict__ z, const ZT *__restrict__ zc,
                                                      double *__restrict__ tmp, const int *__restrict__ done) {
  // zc aliases z but is only read at rows of the *other* colour, which this launch never writes: declaring it as a
  // separate restrict pointer lets the compiler issue those loads ahead of the stores to z (otherwise every level
  // waits for the previous level's stores to retire: vmcnt is in-order)
  constexpr int D = NTOP + 2 * NSIDE;
  constexpr int H = NTOP / 2;
  using SM = TsxSm<H>;
  if (done && *done) return;
  int col = blockIdx.x * 64 + threadIdx.x;
  if (ROWS) {  // enumerate only the rows of this colour
    const int nrows = ROWS == 1 ? (g.ym + 1) / 2 : g.ym / 2;
    if (col >= nrows * g.xm) return;
    col = (2 * (col / g.xm) + (ROWS - 1)) * g.xm + col % g.xm;
  }
  if (col >= g.ncol) return;
  const long long Nc = g.Nc;
  const int Nz = g.Nz, ncol = g.ncol;
  // neighbour rows for the y coupling (periodic wrap inside the rank; rank edges are block-Jacobi)
  // (with an odd number of rows the two rows meeting at the periodic seam have the same colour: no coupling there)
  const int jrow = col / g.xm;
  const bool seam = g.wrap_y && (g.ym % 2 == 0);
  const long long offN = (jrow + 1 < g.ym) ? (long long)g.xm : (seam ? -(long long)(g.ym - 1) * g.xm : 0);
  const long long offS = (jrow > 0) ? -(long long)g.xm : (seam ? (long long)(g.ym - 1) * g.xm : 0);
  (void)offN;
  (void)offS;
  const double *__restrict__ rt = r + (size_t)D * Nc;
  ZT *__restrict__ zt = z + (size_t)D * Nc;
  double *__restrict__ tGw = tmp, *__restrict__ tGT = tmp + (size_t)H * Nc, *__restrict__ tA = tmp + (size_t)(H + H * H) * Nc,
                      *__restrict__ tB = tmp + (size_t)(H + 2 * H * H) * Nc;

  // ---- upward sweep
  double A[H][H], B[H];
  {
    const double alb = albedo[col] / (double)H;  // assembled surface row: albedo/streams on every pair
#pragma unroll
    for (int a = 0; a < H; ++a) {
      B[a] = rt[(size_t)(2 * a) * ncol + col];
#pragma unroll
      for (int b = 0; b < H; ++b) A[a][b] = alb;
    }
  }
  for (int k = Nz - 1; k >= 0; --k) {
    const size_t c = (size_t)k * ncol + col;
    double Tuu[H][H], Rud[H][H], Rdu[H][H], Tdd[H][H], ru[H], rd[H];
    if (l1d[k]) {
      const double t11 = a11[c], t12 = a12[c];
#pragma unroll
      for (int a = 0; a < H; ++a)
#pragma unroll
        for (int b = 0; b < H; ++b) {
          Tuu[a][b] = Tdd[a][b] = (a == b ? t11 : 0.0);
          Rud[a][b] = Rdu[a][b] = (a == b ? t12 : 0.0);
        }
    } else {
#pragma unroll
      for (int a = 0; a < H; ++a)
#pragma unroll
        for (int b = 0; b < H; ++b) {  // C[dst*D + src]
          Tuu[a][b] = (double)C[(size_t)((2 * a) * D + 2 * b) * Nc + c];
          Rud[a][b] = (double)C[(size_t)((2 * a) * D + 2 * b + 1) * Nc + c];
          Rdu[a][b] = (double)C[(size_t)((2 * a + 1) * D + 2 * b) * Nc + c];
          Tdd[a][b] = (double)C[(size_t)((2 * a + 1) * D + 2 * b + 1) * Nc + c];
        }
    }
#pragma unroll
    for (int a = 0; a < H; ++a) {
      ru[a] = r[(size_t)(2 * a) * Nc + c];
      rd[a] = r[(size_t)(2 * a + 1) * Nc + c];
    }
    if (GS && !l1d[k]) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) {
        const int sd = NTOP + NSIDE + q;
        const long long off = tsx_inward(q) ? offS : offN;
        const double zl = (double)zc[(size_t)sd * Nc + c + off];  // unconditional load, then select: no branch, no wait
        const double zv = off ? zl : 0.0;
#pragma unroll
        for (int a = 0; a < H; ++a) {
          ru[a] += (double)C[(size_t)((2 * a) * D + sd) * Nc + c] * zv;
          rd[a] += (double)C[(size_t)((2 * a + 1) * D + sd) * Nc + c] * zv;
        }
      }
    }
    double RA[H][H], G[H][H], GT[H][H], w[H], Gw[H], AGw[H], TA[H][H], An[H][H], Bn[H];
    SM::matmul(Rdu, A, RA);
    SM::inv_i_minus(RA, G);
    SM::matvec(Rdu, B, w);
#pragma unroll
    for (int a = 0; a < H; ++a) w[a] += rd[a];
    SM::matvec(G, w, Gw);
    SM::matmul(G, Tdd, GT);
    SM::matvec(A, Gw, AGw);
#pragma unroll
    for (int a = 0; a < H; ++a) AGw[a] += B[a];
    SM::matvec(Tuu, AGw, Bn);
    SM::matmul(Tuu, A, TA);
    SM::matmul(TA, GT, An);
#pragma unroll
    for (int a = 0; a < H; ++a) {
      Bn[a] += ru[a];
      tGw[(size_t)a * Nc + c] = Gw[a];
      tB[(size_t)a * Nc + c] = Bn[a];
      B[a] = Bn[a];
#pragma unroll
      for (int b = 0; b < H; ++b) {
        An[a][b] += Rud[a][b];
        tGT[(size_t)(a * H + b) * Nc + c] = GT[a][b];
        tA[(size_t)(a * H + b) * Nc + c] = An[a][b];
        A[a][b] = An[a][b];
      }
    }
  }

  // ---- downward sweep
  double V[H];
#pragma unroll
  for (int a = 0; a < H; ++a) {
    V[a] = rt[(size_t)(2 * a + 1) * ncol + col];       // V_0 = rd_0 (TOA identity row)
    zt[(size_t)(2 * a + 1) * ncol + col] = (ZT)V[a];
  }
  // U_0 = A_0 V_0 + B_0: A, B hold level 0 after the upward sweep
  double U[H];
  SM::matvec(A, V, U);
#pragma unroll
  for (int a = 0; a < H; ++a) U[a] += B[a];
  for (int k = 0; k < Nz; ++k) {
    const size_t c = (size_t)k * ncol + col;
    double Gw[H], GT[H][H], Vn[H], Un[H];
#pragma unroll
    for (int a = 0; a < H; ++a) {
      Gw[a] = tGw[(size_t)a * Nc + c];
#pragma unroll
      for (int b = 0; b < H; ++b) GT[a][b] = tGT[(size_t)(a * H + b) * Nc + c];
    }
    SM::matvec(GT, V, Vn);
#pragma unroll
    for (int a = 0; a < H; ++a) Vn[a] += Gw[a];
    // U_{k+1}
    if (k + 1 < Nz) {
      const size_t cn = c + ncol;
      double An[H][H];
#pragma unroll
      for (int a = 0; a < H; ++a) {
        Un[a] = tB[(size_t)a * Nc + cn];
#pragma unroll
        for (int b = 0; b < H; ++b) An[a][b] = tA[(size_t)(a * H + b) * Nc + cn];
      }
      double t[H];
      SM::matvec(An, Vn, t);
#pragma unroll
      for (int a = 0; a < H; ++a) Un[a] += t[a];
    } else {
      const double alb = albedo[col] / (double)H;
      double sv = 0.0;
#pragma unroll
      for (int a = 0; a < H; ++a) sv += Vn[a];
#pragma unroll
      for (int a = 0; a < H; ++a) {
        Un[a] = rt[(size_t)(2 * a) * ncol + col] + alb * sv;
        zt[(size_t)(2 * a) * ncol + col] = (ZT)Un[a];
      }
    }
    // outputs of cell k: up streams at level k, down streams at level k+1
#pragma unroll
    for (int a = 0; a < H; ++a) {
      z[(size_t)(2 * a) * Nc + c] = (ZT)U[a];
      z[(size_t)(2 * a + 1) * Nc + c] = (ZT)Vn[a];
    }
    // side streams leaving cell k: sources Eup(k+1) = Un, Edn(k) = V
    if (l1d[k]) {
#pragma unroll
      for (int d = NTOP; d < D; ++d) z[(size_t)d * Nc + c] = (ZT)r[(size_t)d * Nc + c];
    } else {
      double zy[NSIDE];
      if (GS) {
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) {
          const long long off = tsx_inward(q) ? offS : offN;
          const double zl = (double)zc[(size_t)(NTOP + NSIDE + q) * Nc + c + off];
          zy[q] = off ? zl : 0.0;
        }
      }
#pragma unroll
      for (int d = NTOP; d < D; ++d) {
        double acc = r[(size_t)d * Nc + c];
#pragma unroll
        for (int a = 0; a < H; ++a) {
          acc += (double)C[(size_t)(d * D + 2 * a) * Nc + c] * Un[a];
          acc += (double)C[(size_t)(d * D + 2 * a + 1) * Nc + c] * V[a];
        }
        if (GS) {
#pragma unroll
          for (int q = 0; q < NSIDE; ++q) acc += (double)C[(size_t)(d * D + NTOP + NSIDE + q) * Nc + c] * zy[q];
        }
        z[(size_t)d * Nc + c] = (ZT)acc;
      }
    }
#pragma unroll
    for (int a = 0; a < H; ++a) {
      V[a] = Vn[a];
      U[a] = Un[a];
    }
  }
  // bottom side dummies: identity rows
#pragma unroll
  for (int d = NTOP; d < D; ++d) zt[(size_t)d * ncol + col] = (ZT)rt[(size_t)d * ncol + col];
}

// ---- 3_10 (H = 1) specialisation with explicit software prefetch: the loads of level k-1 (k+1) are issued before the
// arithmetic of level k, so that the sequential sweep is paced by bandwidth, not by one memory latency per level.
// The loop bodies are branch-free (unconditional loads + selects): a conditional load ends a basic block and costs a
// full s_waitcnt vmcnt(0) per level.  Same mathematics as tsx_k_pc_column<2,4,...>; ROWS / GS as there; HAS1D = some
// layer is 1-D (then a11/a12 are valid arrays).
struct TsxUpIn {   // what one level of the upward sweep needs
  double tuu, rud, rdu, tdd, ru, rd;
};
struct TsxDnIn {   // what one level of the downward sweep needs
  double gw, gt, an, bn;      // Gw, GT of cell k;  A, B of cell k+1 (or the surface closure)
  double rs[8];               // right-hand side of the 8 side streams (incl. y coupling)
  float cu[8], cv[8];         // c(Eup -> side d), c(Edn -> side d); zero in 1-D layers
};

// XL: additionally the +-x side streams of the same row enter the right-hand side with their values of this colour's
// previous pass (zx, a different buffer than the one being written): Jacobi in x on top of Gauss-Seidel in y.
template <typename CT, int ROWS, bool GS, typename ZT, bool HAS1D, bool XL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void tsx_k_pc_column_h1(TsxGeo g, const CT *__restrict__ C, const uint8_t *__restrict__ l1d,
                                                         const double *__restrict__ a11, const double *__restrict__ a12,
                                                         const double *__restrict__ albedo, const double *__restrict__ r,
                                                         ZT *__restrict__ z, const ZT *__restrict__ zc,
                                                         const ZT *__restrict__ zx, void *__restrict__ tmp_,
                                                         const int *__restrict__ done) {
  constexpr int D = 10, NTOP = 2, NSIDE = 4;
  if (done && *done) return;
  int col = blockIdx.x * 64 + threadIdx.x;
  if (ROWS) {
    const int nrows = ROWS == 1 ? (g.ym + 1) / 2 : g.ym / 2;
    if (col >= nrows * g.xm) return;
    col = (2 * (col / g.xm) + (ROWS - 1)) * g.xm + col % g.xm;
  }
  if (col >= g.ncol) return;
  const long long Nc = g.Nc;
  const int Nz = g.Nz, ncol = g.ncol;
  const int jrow = col / g.xm;
  const bool seam = g.wrap_y && (g.ym % 2 == 0);
  const long long offN = (jrow + 1 < g.ym) ? (long long)g.xm : (seam ? -(long long)(g.ym - 1) * g.xm : 0);
  const long long offS = (jrow > 0) ? -(long long)g.xm : (seam ? (long long)(g.ym - 1) * g.xm : 0);
  const int icol = col % g.xm;
  const long long offE = (icol + 1 < g.xm) ? 1 : (g.wrap_x ? -(long long)(g.xm - 1) : 0);
  const long long offW = (icol > 0) ? -1 : (g.wrap_x ? (long long)(g.xm - 1) : 0);
  const double *__restrict__ rt = r + (size_t)D * Nc;
  ZT *__restrict__ zt = z + (size_t)D * Nc;
  // sweep temporaries in the precision of the output (fp32 for fp32 directions)
  ZT *__restrict__ tmp = (ZT *)tmp_;
  ZT *__restrict__ tGw = tmp, *__restrict__ tGT = tmp + Nc, *__restrict__ tA = tmp + 2 * Nc, *__restrict__ tB = tmp + 3 * Nc;
  const double albc = albedo[col], rsurf = rt[col];

  auto load_up = [&](int k) {
    TsxUpIn u;
    const size_t c = (size_t)k * ncol + col;
    u.ru = r[c];
    u.rd = r[(size_t)Nc + c];
    u.tuu = (double)C[(size_t)0 * Nc + c];   // c(src 0 -> dst 0)
    u.rud = (double)C[(size_t)1 * Nc + c];   // c(src 1 -> dst 0)
    u.rdu = (double)C[(size_t)10 * Nc + c];  // c(src 0 -> dst 1)
    u.tdd = (double)C[(size_t)11 * Nc + c];  // c(src 1 -> dst 1)
    double gu = 0.0, gd = 0.0;
    if (GS) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) {
        const int sd = NTOP + NSIDE + q;
        const double zl = (double)zc[(size_t)sd * Nc + c + (tsx_inward(q) ? offS : offN)];
        const double zv = (tsx_inward(q) ? offS : offN) ? zl : 0.0;  // select, not multiply: the unused slot may hold NaN
        gu += (double)C[(size_t)(0 * D + sd) * Nc + c] * zv;
        gd += (double)C[(size_t)(1 * D + sd) * Nc + c] * zv;
      }
    }
    if (XL) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) {
        const int sd = NTOP + q;
        const double zl = (double)zx[(size_t)sd * Nc + c + (tsx_inward(q) ? offW : offE)];
        const double zv = (tsx_inward(q) ? offW : offE) ? zl : 0.0;
        gu += (double)C[(size_t)(0 * D + sd) * Nc + c] * zv;
        gd += (double)C[(size_t)(1 * D + sd) * Nc + c] * zv;
      }
    }
    if (HAS1D) {
      const bool one = l1d[k] != 0;
      const double t11 = a11[c], t12 = a12[c];
      u.tuu = one ? t11 : u.tuu;
      u.tdd = one ? t11 : u.tdd;
      u.rud = one ? t12 : u.rud;
      u.rdu = one ? t12 : u.rdu;
      gu = one ? 0.0 : gu;
      gd = one ? 0.0 : gd;
    }
    u.ru += gu;
    u.rd += gd;
    return u;
  };

  // ---- upward sweep: U_k = A_k V_k + B_k
  double A = albc, B = rsurf;
  {
    TsxUpIn cu = load_up(Nz - 1);
    for (int k = Nz - 1; k >= 0; --k) {
      const TsxUpIn nx = load_up(k > 0 ? k - 1 : 0);  // prefetch: independent of the recurrence
      const size_t c = (size_t)k * ncol + col;
      const double G = 1.0 / (1.0 - cu.rdu * A);
      const double Gw = G * (cu.rd + cu.rdu * B);
      const double GT = G * cu.tdd;
      const double Bn = cu.ru + cu.tuu * (B + A * Gw);
      const double An = cu.tuu * A * GT + cu.rud;
      tGw[c] = (ZT)Gw;
      tGT[c] = (ZT)GT;
      tA[c] = (ZT)An;
      tB[c] = (ZT)Bn;
      A = An;
      B = Bn;
      cu = nx;
    }
  }

  auto load_dn = [&](int k) {
    TsxDnIn d;
    const size_t c = (size_t)k * ncol + col;
    const bool last = k + 1 >= Nz;
    const size_t cn = last ? c : c + ncol;
    d.gw = (double)tGw[c];
    d.gt = (double)tGT[c];
    const double an = (double)tA[cn], bn = (double)tB[cn];
    d.an = last ? albc : an;  // U_Nz = albedo V_Nz + ru_Nz
    d.bn = last ? rsurf : bn;
    bool one = false;
    if (HAS1D) one = l1d[k] != 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      d.rs[q] = r[(size_t)(NTOP + q) * Nc + c];
      const float fu = (float)C[(size_t)((NTOP + q) * D + 0) * Nc + c], fv = (float)C[(size_t)((NTOP + q) * D + 1) * Nc + c];
      d.cu[q] = one ? 0.0f : fu;
      d.cv[q] = one ? 0.0f : fv;
    }
    if (GS) {
      double zy[NSIDE];
#pragma unroll
      for (int q = 0; q < NSIDE; ++q)
      {
        const double zl = (double)zc[(size_t)(NTOP + NSIDE + q) * Nc + c + (tsx_inward(q) ? offS : offN)];
        zy[q] = (tsx_inward(q) ? offS : offN) ? zl : 0.0;
      }
#pragma unroll
      for (int dd = 0; dd < 8; ++dd) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) acc += (double)C[(size_t)((NTOP + dd) * D + NTOP + NSIDE + q) * Nc + c] * zy[q];
        d.rs[dd] += one ? 0.0 : acc;
      }
    }
    if (XL) {
      double zq[NSIDE];
#pragma unroll
      for (int q = 0; q < NSIDE; ++q)
      {
        const double zl = (double)zx[(size_t)(NTOP + q) * Nc + c + (tsx_inward(q) ? offW : offE)];
        zq[q] = (tsx_inward(q) ? offW : offE) ? zl : 0.0;
      }
#pragma unroll
      for (int dd = 0; dd < 8; ++dd) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) acc += (double)C[(size_t)((NTOP + dd) * D + NTOP + q) * Nc + c] * zq[q];
        d.rs[dd] += one ? 0.0 : acc;
      }
    }
    return d;
  };

  // ---- downward sweep
  double V = rt[(size_t)ncol + col];  // V_0 = rd_0 (TOA identity row)
  zt[(size_t)ncol + col] = (ZT)V;
  double U = A * V + B;               // A, B hold level 0
  {
    TsxDnIn cd = load_dn(0);
    for (int k = 0; k < Nz; ++k) {
      const TsxDnIn nx = load_dn(k + 1 < Nz ? k + 1 : k);
      const size_t c = (size_t)k * ncol + col;
      const double Vn = cd.gw + cd.gt * V;
      const double Un = cd.an * Vn + cd.bn;
      z[c] = (ZT)U;
      z[(size_t)Nc + c] = (ZT)Vn;
#pragma unroll
      for (int q = 0; q < 8; ++q) z[(size_t)(NTOP + q) * Nc + c] = (ZT)(cd.rs[q] + (double)cd.cu[q] * Un + (double)cd.cv[q] * V);
      V = Vn;
      U = Un;
      cd = nx;
    }
  }
  zt[col] = (ZT)U;  // U_Nz
#pragma unroll
  for (int d = NTOP; d < D; ++d) zt[(size_t)d * ncol + col] = (ZT)rt[(size_t)d * ncol + col];
}

// ---- 3_10 preconditioner on a *packed fp16* copy of the transport blocks (fp32 directions only) -------------------
// The sweep above is paced by memory latency, not bytes: one wave can only track 63 outstanding vector-memory
// operations (s_waitcnt vmcnt is 6 bits), and with one 4-byte load per coefficient a single level already needs > 100.
// Here the 100 coefficients of a cell are regrouped into 13 records of 8 halves (16 B) in the order the two sweeps
// consume them, P[(grp * Nc + cell)] as uint4: a level costs 3 (up) + 10 (down) coefficient loads of 16 B per lane, so
// several levels fit under the counter and the sweeps are software-pipelined PU / PD levels deep.
//   grp 0: tuu rud rdu tdd | c(y0->0) c(y0->1) c(y1->0) c(y1->1)          (y_q = src dof 6+q, x_q = src dof 2+q)
//   grp 1: c(y2->0) c(y2->1) c(y3->0) c(y3->1) | c(x0->0) c(x0->1) c(x1->0) c(x1->1)
//   grp 2: c(x2->0) c(x2->1) c(x3->0) c(x3->1) | pad
//   grp 3: c(0 -> side d), d = 2..9          grp 4: c(1 -> side d)
//   grp 5+m: c(y_q -> 2+2m), c(y_q -> 3+2m)  grp 9+m: c(x_q -> 2+2m), c(x_q -> 3+2m)      (m = 0..3, q = 0..3)
typedef _Float16 tsx_h8 __attribute__((ext_vector_type(8)));
constexpr int TSX_P16_GROUPS = 13;

// plane index dst*10+src held by element e of group grp; -1 = padding
__host__ __device__ constexpr int tsx_p16_plane(int grp, int e) {
  if (grp == 0) {
    if (e < 4) return (e >> 1) * 10 + (e & 1);
    return (e & 1) * 10 + 6 + ((e - 4) >> 1);
  }
  if (grp == 1) {
    if (e < 4) return (e & 1) * 10 + 8 + (e >> 1);
    return (e & 1) * 10 + 2 + ((e - 4) >> 1);
  }
  if (grp == 2) return e < 4 ? (e & 1) * 10 + 4 + (e >> 1) : -1;
  if (grp == 3) return (2 + e) * 10 + 0;
  if (grp == 4) return (2 + e) * 10 + 1;
  if (grp < 9) return (2 + 2 * (grp - 5) + (e >> 2)) * 10 + 6 + (e & 3);
  return (2 + 2 * (grp - 9) + (e >> 2)) * 10 + 2 + (e & 3);
}

// 8_16 (D = 16): 32 records, no padding.  t = top dst 0..7, d = side dst 8..15, y_q = src 12+q, x_q = src 8+q.
//   grp 0..7:   c(src 0..7 -> top dst t = grp)                       (Tuu/Rud/Rdu/Tdd interleaved by stream parity)
//   grp 8+m:    c(y_q -> 2m), c(y_q -> 2m+1)        grp 12+m: c(x_q -> 2m), c(x_q -> 2m+1)
//   grp 16+dd:  c(src 0..7 -> side dst 8+dd)
//   grp 24+m:   c(y_q -> 8+2m), c(y_q -> 9+2m)      grp 28+m: c(x_q -> 8+2m), c(x_q -> 9+2m)
constexpr int TSX_P16H_GROUPS = 32;
__host__ __device__ constexpr int tsx_p16h_plane(int grp, int e) {
  if (grp < 8) return grp * 16 + e;
  if (grp < 12) return (2 * (grp - 8) + (e >> 2)) * 16 + 12 + (e & 3);
  if (grp < 16) return (2 * (grp - 12) + (e >> 2)) * 16 + 8 + (e & 3);
  if (grp < 24) return (8 + grp - 16) * 16 + e;
  if (grp < 28) return (8 + 2 * (grp - 24) + (e >> 2)) * 16 + 12 + (e & 3);
  return (8 + 2 * (grp - 28) + (e >> 2)) * 16 + 8 + (e & 3);
}

template <typename CT, int NTOP>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_pack_p16(long long Nc, const CT *__restrict__ C, tsx_h8 *__restrict__ P) {
  constexpr int NG = NTOP == 2 ? TSX_P16_GROUPS : TSX_P16H_GROUPS;
  const long long n = Nc * NG;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * TSX_BLOCK) {
    const int grp = (int)(q / Nc);
    const long long c = q - (long long)grp * Nc;
    tsx_h8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int pl = NTOP == 2 ? tsx_p16_plane(grp, e) : tsx_p16h_plane(grp, e);
      v[e] = pl >= 0 ? (_Float16)C[(size_t)pl * Nc + c] : (_Float16)0;
    }
    P[q] = v;
  }
}

struct TsxUpRaw {
  tsx_h8 c0, c1, c2;
  double ru, rd, t11, t12;
  float zy[4], zx[4];
};
struct TsxDnRaw {
  tsx_h8 cu, cv, cy[4], cx[4];
  float4 t;        // Gw_k, GT_k, A_{k+1}, B_{k+1}
  double rs[8];
  float zy[4], zx[4];
};

template <int ROWS, bool GS, bool HAS1D, bool XL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void tsx_k_pc_column_p16(
    TsxGeo g, const tsx_h8 *__restrict__ P, const uint8_t *__restrict__ l1d, const double *__restrict__ a11,
    const double *__restrict__ a12, const double *__restrict__ albedo, const double *__restrict__ r, float *__restrict__ z,
    const float *__restrict__ zc, const float *__restrict__ zx, float4 *__restrict__ tmp, const int *__restrict__ done) {
  constexpr int D = 10, NTOP = 2, NSIDE = 4;
  constexpr int PU = 4, PD = 2;  // prefetch depth of the upward / downward sweep (levels)
  if (done && *done) return;
  int col = blockIdx.x * 64 + threadIdx.x;
  if (ROWS) {
    const int nrows = ROWS == 1 ? (g.ym + 1) / 2 : g.ym / 2;
    if (col >= nrows * g.xm) return;
    col = (2 * (col / g.xm) + (ROWS - 1)) * g.xm + col % g.xm;
  }
  if (col >= g.ncol) return;
  const long long Nc = g.Nc;
  const int Nz = g.Nz, ncol = g.ncol;
  const int jrow = col / g.xm;
  const bool seam = g.wrap_y && (g.ym % 2 == 0);
  const long long offN = (jrow + 1 < g.ym) ? (long long)g.xm : (seam ? -(long long)(g.ym - 1) * g.xm : 0);
  const long long offS = (jrow > 0) ? -(long long)g.xm : (seam ? (long long)(g.ym - 1) * g.xm : 0);
  const int icol = col % g.xm;
  const long long offE = (icol + 1 < g.xm) ? 1 : (g.wrap_x ? -(long long)(g.xm - 1) : 0);
  const long long offW = (icol > 0) ? -1 : (g.wrap_x ? (long long)(g.xm - 1) : 0);
  const double *__restrict__ rt = r + (size_t)D * Nc;
  float *__restrict__ zt = z + (size_t)D * Nc;
  const double albc = albedo[col], rsurf = rt[col];

  // loads only: nothing here depends on loaded data, so that the whole record of a level is in flight at once
  auto load_up = [&](int k) {
    TsxUpRaw u;
    const size_t c = (size_t)k * ncol + col;
    u.c0 = P[(size_t)0 * Nc + c];
    if (GS) u.c1 = P[(size_t)1 * Nc + c];
    if (XL) u.c2 = P[(size_t)2 * Nc + c];
    u.ru = r[c];
    u.rd = r[(size_t)Nc + c];
    if (GS) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) u.zy[q] = zc[(size_t)(NTOP + NSIDE + q) * Nc + c + (tsx_inward(q) ? offS : offN)];
    }
    if (XL) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) u.zx[q] = zx[(size_t)(NTOP + q) * Nc + c + (tsx_inward(q) ? offW : offE)];
    }
    if (HAS1D) {
      u.t11 = a11[c];
      u.t12 = a12[c];
    }
    return u;
  };

  double A = albc, B = rsurf;
  auto step_up = [&](int k, const TsxUpRaw &u) {
    const size_t c = (size_t)k * ncol + col;
    double tuu = (double)u.c0[0], rud = (double)u.c0[1], rdu = (double)u.c0[2], tdd = (double)u.c0[3];
    double gu = 0.0, gd = 0.0;
    if (GS) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) {
        const double zv = (tsx_inward(q) ? offS : offN) ? (double)u.zy[q] : 0.0;  // select: the unused slot may hold NaN
        const double c0 = q < 2 ? (double)u.c0[4 + 2 * q] : (double)u.c1[2 * (q - 2)];
        const double c1 = q < 2 ? (double)u.c0[5 + 2 * q] : (double)u.c1[2 * (q - 2) + 1];
        gu += c0 * zv;
        gd += c1 * zv;
      }
    }
    if (XL) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) {
        const double zv = (tsx_inward(q) ? offW : offE) ? (double)u.zx[q] : 0.0;
        const double c0 = q < 2 ? (double)u.c1[4 + 2 * q] : (double)u.c2[2 * (q - 2)];
        const double c1 = q < 2 ? (double)u.c1[5 + 2 * q] : (double)u.c2[2 * (q - 2) + 1];
        gu += c0 * zv;
        gd += c1 * zv;
      }
    }
    if (HAS1D) {
      const bool one = l1d[k] != 0;
      tuu = one ? u.t11 : tuu;
      tdd = one ? u.t11 : tdd;
      rud = one ? u.t12 : rud;
      rdu = one ? u.t12 : rdu;
      gu = one ? 0.0 : gu;
      gd = one ? 0.0 : gd;
    }
    const double ru = u.ru + gu, rd = u.rd + gd;
    const double G = 1.0 / (1.0 - rdu * A);
    const double Gw = G * (rd + rdu * B);
    const double GT = G * tdd;
    tmp[c] = make_float4((float)Gw, (float)GT, (float)A, (float)B);
    const double Bn = ru + tuu * (B + A * Gw);
    const double An = tuu * A * GT + rud;
    A = An;
    B = Bn;
  };

  // ---- upward sweep: U_k = A_k V_k + B_k
  {
    int k = Nz - 1;
    for (int rr = Nz % PU; rr > 0; --rr, --k) {
      const TsxUpRaw u = load_up(k);
      step_up(k, u);
    }
    if (k >= 0) {  // k + 1 is a multiple of PU
      TsxUpRaw q[PU];
#pragma unroll
      for (int p = 0; p < PU; ++p) q[p] = load_up(k - p);
      for (; k >= 0; k -= PU) {
#pragma unroll
        for (int p = 0; p < PU; ++p) {
          const TsxUpRaw cu = q[p];
          const int kn = k - p - PU;
          q[p] = load_up(kn >= 0 ? kn : 0);
          step_up(k - p, cu);
        }
      }
    }
  }

  auto load_dn = [&](int k) {
    TsxDnRaw d;
    const size_t c = (size_t)k * ncol + col;
    d.cu = P[(size_t)3 * Nc + c];
    d.cv = P[(size_t)4 * Nc + c];
    d.t = tmp[c];
#pragma unroll
    for (int q = 0; q < 8; ++q) d.rs[q] = r[(size_t)(NTOP + q) * Nc + c];
    if (GS) {
#pragma unroll
      for (int m = 0; m < 4; ++m) d.cy[m] = P[(size_t)(5 + m) * Nc + c];
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) d.zy[q] = zc[(size_t)(NTOP + NSIDE + q) * Nc + c + (tsx_inward(q) ? offS : offN)];
    }
    if (XL) {
#pragma unroll
      for (int m = 0; m < 4; ++m) d.cx[m] = P[(size_t)(9 + m) * Nc + c];
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) d.zx[q] = zx[(size_t)(NTOP + q) * Nc + c + (tsx_inward(q) ? offW : offE)];
    }
    return d;
  };

  double V = rt[(size_t)ncol + col];  // V_0 = rd_0 (TOA identity row)
  zt[(size_t)ncol + col] = (float)V;
  double U = A * V + B;               // A, B hold level 0
  auto step_dn = [&](int k, const TsxDnRaw &d) {
    const size_t c = (size_t)k * ncol + col;
    bool one = false;
    if (HAS1D) one = l1d[k] != 0;
    const double Vn = (double)d.t.x + (double)d.t.y * V;
    const double Un = (double)d.t.z * Vn + (double)d.t.w;
    z[c] = (float)U;
    z[(size_t)Nc + c] = (float)Vn;
    double zy[NSIDE], zq[NSIDE];
    if (GS) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) zy[q] = (tsx_inward(q) ? offS : offN) ? (double)d.zy[q] : 0.0;
    }
    if (XL) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) zq[q] = (tsx_inward(q) ? offW : offE) ? (double)d.zx[q] : 0.0;
    }
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) {
      double acc = (double)d.cu[dd] * Un + (double)d.cv[dd] * V;
      if (GS) {
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) acc += (double)d.cy[dd >> 1][(dd & 1) * 4 + q] * zy[q];
      }
      if (XL) {
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) acc += (double)d.cx[dd >> 1][(dd & 1) * 4 + q] * zq[q];
      }
      z[(size_t)(NTOP + dd) * Nc + c] = (float)(d.rs[dd] + (one ? 0.0 : acc));
    }
    V = Vn;
    U = Un;
  };

  // ---- downward sweep
  {
    int k = 0;
    for (int rr = Nz % PD; rr > 0; --rr, ++k) {
      const TsxDnRaw d = load_dn(k);
      step_dn(k, d);
    }
    if (k < Nz) {
      TsxDnRaw q[PD];
#pragma unroll
      for (int p = 0; p < PD; ++p) q[p] = load_dn(k + p);
      for (; k < Nz; k += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
          const TsxDnRaw cd = q[p];
          const int kn = k + p + PD;
          q[p] = load_dn(kn < Nz ? kn : Nz - 1);
          step_dn(k + p, cd);
        }
      }
    }
  }
  zt[col] = (float)U;  // U_Nz
#pragma unroll
  for (int d = NTOP; d < D; ++d) zt[(size_t)d * ncol + col] = (float)rt[(size_t)d * ncol + col];
}

// ---- 8_16 (H = 4 up/down pairs) on the packed fp16 blocks: same mathematics as tsx_k_pc_column<8,4,...> (4x4 block
// recurrences), same software pipeline as tsx_k_pc_column_p16.  Temporaries per cell: 10 float4 records
// [Gw | GT rows 0..3 | A_{k+1} rows 0..3 | B_{k+1}].
struct TsxUpRawH {
  tsx_h8 row[8], cy[4], cx[4];
  double r[8], t11, t12;
  float zy[4], zx[4];
};
struct TsxDnRawH {
  tsx_h8 row[8], cy[4], cx[4];
  float4 t[10];
  double rs[8];
  float zy[4], zx[4];
};

template <int ROWS, bool GS, bool HAS1D, bool XL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void tsx_k_pc_column_p16h(
    TsxGeo g, const tsx_h8 *__restrict__ P, const uint8_t *__restrict__ l1d, const double *__restrict__ a11,
    const double *__restrict__ a12, const double *__restrict__ albedo, const double *__restrict__ r, float *__restrict__ z,
    const float *__restrict__ zc, const float *__restrict__ zx, float4 *__restrict__ tmp, const int *__restrict__ done) {
  constexpr int D = 16, NTOP = 8, NSIDE = 4, H = 4;
  constexpr int PU = 2, PD = 1;
  using SM = TsxSm<H>;
  if (done && *done) return;
  int col = blockIdx.x * 64 + threadIdx.x;
  if (ROWS) {
    const int nrows = ROWS == 1 ? (g.ym + 1) / 2 : g.ym / 2;
    if (col >= nrows * g.xm) return;
    col = (2 * (col / g.xm) + (ROWS - 1)) * g.xm + col % g.xm;
  }
  if (col >= g.ncol) return;
  const long long Nc = g.Nc;
  const int Nz = g.Nz, ncol = g.ncol;
  const int jrow = col / g.xm;
  const bool seam = g.wrap_y && (g.ym % 2 == 0);
  const long long offN = (jrow + 1 < g.ym) ? (long long)g.xm : (seam ? -(long long)(g.ym - 1) * g.xm : 0);
  const long long offS = (jrow > 0) ? -(long long)g.xm : (seam ? (long long)(g.ym - 1) * g.xm : 0);
  const int icol = col % g.xm;
  const long long offE = (icol + 1 < g.xm) ? 1 : (g.wrap_x ? -(long long)(g.xm - 1) : 0);
  const long long offW = (icol > 0) ? -1 : (g.wrap_x ? (long long)(g.xm - 1) : 0);
  const double *__restrict__ rt = r + (size_t)D * Nc;
  float *__restrict__ zt = z + (size_t)D * Nc;
  const double albh = albedo[col] / (double)H;  // assembled surface row: albedo/streams on every pair

  auto load_up = [&](int k) {
    TsxUpRawH u;
    const size_t c = (size_t)k * ncol + col;
#pragma unroll
    for (int t = 0; t < 8; ++t) u.row[t] = P[(size_t)t * Nc + c];
#pragma unroll
    for (int t = 0; t < 8; ++t) u.r[t] = r[(size_t)t * Nc + c];
    if (GS) {
#pragma unroll
      for (int m = 0; m < 4; ++m) u.cy[m] = P[(size_t)(8 + m) * Nc + c];
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) u.zy[q] = zc[(size_t)(NTOP + NSIDE + q) * Nc + c + (tsx_inward(q) ? offS : offN)];
    }
    if (XL) {
#pragma unroll
      for (int m = 0; m < 4; ++m) u.cx[m] = P[(size_t)(12 + m) * Nc + c];
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) u.zx[q] = zx[(size_t)(NTOP + q) * Nc + c + (tsx_inward(q) ? offW : offE)];
    }
    if (HAS1D) {
      u.t11 = a11[c];
      u.t12 = a12[c];
    }
    return u;
  };

  double A[H][H], B[H];
#pragma unroll
  for (int a = 0; a < H; ++a) {
    B[a] = rt[(size_t)(2 * a) * ncol + col];
#pragma unroll
    for (int b = 0; b < H; ++b) A[a][b] = albh;
  }
  auto step_up = [&](int k, const TsxUpRawH &u) {
    const size_t c = (size_t)k * ncol + col;
    bool one = false;
    if (HAS1D) one = l1d[k] != 0;
    double Tuu[H][H], Rud[H][H], Rdu[H][H], Tdd[H][H], ru[H], rd[H];
#pragma unroll
    for (int a = 0; a < H; ++a) {
#pragma unroll
      for (int b = 0; b < H; ++b) {
        const double dg = a == b ? 1.0 : 0.0;
        Tuu[a][b] = one ? dg * u.t11 : (double)u.row[2 * a][2 * b];
        Rud[a][b] = one ? dg * u.t12 : (double)u.row[2 * a][2 * b + 1];
        Rdu[a][b] = one ? dg * u.t12 : (double)u.row[2 * a + 1][2 * b];
        Tdd[a][b] = one ? dg * u.t11 : (double)u.row[2 * a + 1][2 * b + 1];
      }
      double gu = 0.0, gd = 0.0;
      if (GS) {
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) {
          const double zv = (tsx_inward(q) ? offS : offN) ? (double)u.zy[q] : 0.0;  // select: the unused slot may hold NaN
          gu += (double)u.cy[a][q] * zv;
          gd += (double)u.cy[a][4 + q] * zv;
        }
      }
      if (XL) {
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) {
          const double zv = (tsx_inward(q) ? offW : offE) ? (double)u.zx[q] : 0.0;
          gu += (double)u.cx[a][q] * zv;
          gd += (double)u.cx[a][4 + q] * zv;
        }
      }
      ru[a] = u.r[2 * a] + (one ? 0.0 : gu);
      rd[a] = u.r[2 * a + 1] + (one ? 0.0 : gd);
    }
    double RA[H][H], G[H][H], GT[H][H], w[H], Gw[H], AGw[H], TA[H][H], An[H][H], Bn[H];
    SM::matmul(Rdu, A, RA);
    SM::inv_i_minus(RA, G);
    SM::matvec(Rdu, B, w);
#pragma unroll
    for (int a = 0; a < H; ++a) w[a] += rd[a];
    SM::matvec(G, w, Gw);
    SM::matmul(G, Tdd, GT);
    tmp[(size_t)0 * Nc + c] = make_float4((float)Gw[0], (float)Gw[1], (float)Gw[2], (float)Gw[3]);
#pragma unroll
    for (int a = 0; a < H; ++a) {
      tmp[(size_t)(1 + a) * Nc + c] = make_float4((float)GT[a][0], (float)GT[a][1], (float)GT[a][2], (float)GT[a][3]);
      tmp[(size_t)(5 + a) * Nc + c] = make_float4((float)A[a][0], (float)A[a][1], (float)A[a][2], (float)A[a][3]);
    }
    tmp[(size_t)9 * Nc + c] = make_float4((float)B[0], (float)B[1], (float)B[2], (float)B[3]);
    SM::matvec(A, Gw, AGw);
#pragma unroll
    for (int a = 0; a < H; ++a) AGw[a] += B[a];
    SM::matvec(Tuu, AGw, Bn);
    SM::matmul(Tuu, A, TA);
    SM::matmul(TA, GT, An);
#pragma unroll
    for (int a = 0; a < H; ++a) {
      B[a] = Bn[a] + ru[a];
#pragma unroll
      for (int b = 0; b < H; ++b) A[a][b] = An[a][b] + Rud[a][b];
    }
  };

  // ---- upward sweep
  {
    int k = Nz - 1;
    for (int rr = Nz % PU; rr > 0; --rr, --k) {
      const TsxUpRawH u = load_up(k);
      step_up(k, u);
    }
    if (k >= 0) {
      TsxUpRawH q[PU];
#pragma unroll
      for (int p = 0; p < PU; ++p) q[p] = load_up(k - p);
      for (; k >= 0; k -= PU) {
#pragma unroll
        for (int p = 0; p < PU; ++p) {
          const TsxUpRawH cu = q[p];
          const int kn = k - p - PU;
          q[p] = load_up(kn >= 0 ? kn : 0);
          step_up(k - p, cu);
        }
      }
    }
  }

  auto load_dn = [&](int k) {
    TsxDnRawH d;
    const size_t c = (size_t)k * ncol + col;
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) d.row[dd] = P[(size_t)(16 + dd) * Nc + c];
#pragma unroll
    for (int q = 0; q < 10; ++q) d.t[q] = tmp[(size_t)q * Nc + c];
#pragma unroll
    for (int q = 0; q < 8; ++q) d.rs[q] = r[(size_t)(NTOP + q) * Nc + c];
    if (GS) {
#pragma unroll
      for (int m = 0; m < 4; ++m) d.cy[m] = P[(size_t)(24 + m) * Nc + c];
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) d.zy[q] = zc[(size_t)(NTOP + NSIDE + q) * Nc + c + (tsx_inward(q) ? offS : offN)];
    }
    if (XL) {
#pragma unroll
      for (int m = 0; m < 4; ++m) d.cx[m] = P[(size_t)(28 + m) * Nc + c];
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) d.zx[q] = zx[(size_t)(NTOP + q) * Nc + c + (tsx_inward(q) ? offW : offE)];
    }
    return d;
  };

  double V[H], U[H];
#pragma unroll
  for (int a = 0; a < H; ++a) {
    V[a] = rt[(size_t)(2 * a + 1) * ncol + col];  // V_0 = rd_0 (TOA identity rows)
    zt[(size_t)(2 * a + 1) * ncol + col] = (float)V[a];
  }
  SM::matvec(A, V, U);  // A, B hold level 0
#pragma unroll
  for (int a = 0; a < H; ++a) U[a] += B[a];

  auto f4 = [](const float4 &v, int i) { return (double)(i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w); };
  auto step_dn = [&](int k, const TsxDnRawH &d) {
    const size_t c = (size_t)k * ncol + col;
    bool one = false;
    if (HAS1D) one = l1d[k] != 0;
    double Vn[H], Un[H];
#pragma unroll
    for (int a = 0; a < H; ++a) {
      double v = f4(d.t[0], a);
#pragma unroll
      for (int b = 0; b < H; ++b) v += f4(d.t[1 + a], b) * V[b];
      Vn[a] = v;
    }
#pragma unroll
    for (int a = 0; a < H; ++a) {  // U_{k+1} = A_{k+1} V_{k+1} + B_{k+1} (the surface closure is what the sweep started from)
      double v = f4(d.t[9], a);
#pragma unroll
      for (int b = 0; b < H; ++b) v += f4(d.t[5 + a], b) * Vn[b];
      Un[a] = v;
    }
#pragma unroll
    for (int a = 0; a < H; ++a) {
      z[(size_t)(2 * a) * Nc + c] = (float)U[a];
      z[(size_t)(2 * a + 1) * Nc + c] = (float)Vn[a];
    }
    double zy[NSIDE], zq[NSIDE];
    if (GS) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) zy[q] = (tsx_inward(q) ? offS : offN) ? (double)d.zy[q] : 0.0;
    }
    if (XL) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) zq[q] = (tsx_inward(q) ? offW : offE) ? (double)d.zx[q] : 0.0;
    }
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) {
      double acc = 0.0;
#pragma unroll
      for (int a = 0; a < H; ++a) acc += (double)d.row[dd][2 * a] * Un[a] + (double)d.row[dd][2 * a + 1] * V[a];
      if (GS) {
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) acc += (double)d.cy[dd >> 1][(dd & 1) * 4 + q] * zy[q];
      }
      if (XL) {
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) acc += (double)d.cx[dd >> 1][(dd & 1) * 4 + q] * zq[q];
      }
      z[(size_t)(NTOP + dd) * Nc + c] = (float)(d.rs[dd] + (one ? 0.0 : acc));
    }
#pragma unroll
    for (int a = 0; a < H; ++a) {
      V[a] = Vn[a];
      U[a] = Un[a];
    }
  };

  // ---- downward sweep
  {
    TsxDnRawH cd = load_dn(0);
    for (int k = 0; k < Nz; ++k) {
      const TsxDnRawH nx = load_dn(k + 1 < Nz ? k + 1 : k);
      step_dn(k, cd);
      cd = nx;
    }
    (void)PD;
  }
#pragma unroll
  for (int a = 0; a < H; ++a) zt[(size_t)(2 * a) * ncol + col] = (float)U[a];  // U_Nz
#pragma unroll
  for (int d = NTOP; d < D; ++d) zt[(size_t)d * ncol + col] = (float)rt[(size_t)d * ncol + col];
}

__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_widen(long long n, const float *__restrict__ a, double *__restrict__ o) {
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * TSX_BLOCK) o[q] = (double)a[q];
}

// out = a - b   (second preconditioner sweep: residual of the first)
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_sub(long long n2, const double2 *__restrict__ a, const double2 *__restrict__ b,
                                                       double2 *__restrict__ o, const int *__restrict__ done) {
  if (done && *done) return;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n2; q += (long long)gridDim.x * TSX_BLOCK) {
    const double2 x = a[q], y = b[q];
    double2 r;
    r.x = x.x - y.x;
    r.y = x.y - y.y;
    o[q] = r;
  }
}
// o += a
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_addto(long long n2, const double2 *__restrict__ a, double2 *__restrict__ o,
                                                         const int *__restrict__ done) {
  if (done && *done) return;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n2; q += (long long)gridDim.x * TSX_BLOCK) {
    const double2 x = a[q];
    double2 r = o[q];
    r.x += x.x;
    r.y += x.y;
    o[q] = r;
  }
}

// ------------------------------------------------------------------------------------------------
// BLAS-1 stages of the flexible BiCGStab (KSPFBCGS, selected at src/pprts.F90:4342), fused so that a
// full iteration moves 19 N-vectors besides the two operator applications.
// r = b - y (y = A x0); rhat = r; p = r; slot0 = (r,r)
template <typename RT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_residual0(long long n, const double *__restrict__ b,
                                                             const double *__restrict__ y, double *__restrict__ r,
                                                             RT *__restrict__ rhat, double *__restrict__ p,
                                                             double *__restrict__ partials) {
  double sum[2] = {0.0, 0.0};  // slot0 = (rhat, r) with rhat as stored, slot1 = (r, r)
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * TSX_BLOCK) {
    const double v = b[q] - y[q];
    r[q] = v;
    rhat[q] = (RT)v;
    p[q] = v;
    sum[0] += (double)(RT)v * v;
    sum[1] += v * v;
  }
  tsx_block_reduce_store<2>(sum, partials);
}

// p = r + beta (p - omega v)
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_pupdate(long long n2, const TsxScalars *__restrict__ sc,
                                                           const double2 *__restrict__ r, double2 *__restrict__ p,
                                                           const double2 *__restrict__ v) {
  if (sc->done) return;
  const double beta = sc->beta, omega = sc->omega;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n2; q += (long long)gridDim.x * TSX_BLOCK) {
    const double2 rr = r[q], pp = p[q], vv = v[q];
    double2 o;
    o.x = rr.x + beta * (pp.x - omega * vv.x);
    o.y = rr.y + beta * (pp.y - omega * vv.y);
    p[q] = o;
  }
}

// s = r - alpha v
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_supdate(long long n2, const TsxScalars *__restrict__ sc,
                                                           const double2 *__restrict__ r, const double2 *__restrict__ v,
                                                           double2 *__restrict__ s) {
  if (sc->done) return;
  const double alpha = sc->alpha;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n2; q += (long long)gridDim.x * TSX_BLOCK) {
    const double2 rr = r[q], vv = v[q];
    double2 o;
    o.x = rr.x - alpha * vv.x;
    o.y = rr.y - alpha * vv.y;
    s[q] = o;
  }
}

// x += alpha ph + omega sh; r = s - omega t; slot0 = (rhat, r), slot1 = (r, r).  PT: storage of ph/sh, RT: of rhat
template <typename PT, typename RT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_xrupdate(long long n2, const TsxScalars *__restrict__ sc,
                                                            double2 *__restrict__ x, const PT *__restrict__ ph,
                                                            const PT *__restrict__ sh, const double2 *__restrict__ s,
                                                            const double2 *__restrict__ t, const RT *__restrict__ rhat,
                                                            double2 *__restrict__ r, double *__restrict__ partials) {
  if (sc->done) return;
  const double alpha = sc->alpha, omega = sc->omega;
  double sum[2] = {0.0, 0.0};
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n2; q += (long long)gridDim.x * TSX_BLOCK) {
    const double2 xx = x[q], ss = s[q], tt = t[q];
    double pp[2], ss2[2], rh[2];
    TsxVec<2>::ld(ph + 2 * q, pp);
    TsxVec<2>::ld(sh + 2 * q, ss2);
    TsxVec<2>::ld(rhat + 2 * q, rh);
    double2 xo, ro;
    xo.x = xx.x + alpha * pp[0] + omega * ss2[0];
    xo.y = xx.y + alpha * pp[1] + omega * ss2[1];
    ro.x = ss.x - omega * tt.x;
    ro.y = ss.y - omega * tt.y;
    x[q] = xo;
    r[q] = ro;
    sum[0] += rh[0] * ro.x + rh[1] * ro.y;
    sum[1] += ro.x * ro.x + ro.y * ro.y;
  }
  tsx_block_reduce_store<2>(sum, partials);
}

// ------------------------------------------------------------------------------------------------
// Scalar stage: one block.  mode bit0: reduce the per-block partials into sc->red (fixed order);
// mode bit1: run the stage's scalar algebra (after the all-reduce when ranks > 1).
// Stop rule restates MyKSPConverged (src/pprts.F90:4437-4486).
enum { TSX_STAGE_INIT = 0, TSX_STAGE_ALPHA = 1, TSX_STAGE_OMEGA = 2, TSX_STAGE_RHO = 3 };

__global__ __launch_bounds__(1024) void tsx_k_scalar(TsxScalars *__restrict__ sc, const double *__restrict__ partials,
                                                     int nblocks, int nslots, int stage, int mode) {
  __shared__ double sm[TSX_NSLOTS][16];
  if (sc->done) return;
  if (mode & 1) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int s = 0; s < nslots; ++s) {
      double v = 0.0;
      for (int q = threadIdx.x; q < nblocks; q += 1024) v += partials[(size_t)s * TSX_MAX_PARTIAL_BLOCKS + q];
      v = tsx_wave_sum(v);
      if (lane == 0) sm[s][wv] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int s = 0; s < nslots; ++s) {
        double v = 0.0;
        for (int q = 0; q < 16; ++q) v += sm[s][q];
        sc->red[s] = v;
      }
    }
    __syncthreads();
  }
  if (!(mode & 2) || threadIdx.x != 0) return;
  const double tiny = 2.2250738585072014e-308;
  switch (stage) {
    case TSX_STAGE_INIT: {
      const double rn = sqrt(sc->red[1]);
      sc->rnorm = rn;
      if (!sc->restart) {
        sc->rnorm0 = rn > tiny ? rn : tiny;  // n == 0: store initial norm, no test (:4455-4458)
        sc->hist[0] = rn;
        sc->nhist = 1;
        sc->its = 0;
      }
      sc->rho = sc->red[0];
      sc->rho_old = 1.0;
      sc->alpha = 1.0;
      sc->omega = 1.0;
      sc->beta = 0.0;
      sc->reason = 0;
      if (sc->red[1] == 0.0) {  // exact initial guess: nothing to do
        sc->reason = 2;
        sc->done = 1;
      } else if (rn != rn) {
        sc->reason = -9;
        sc->done = 1;
      } else if (sc->restart && (rn / sc->rnorm0 <= sc->rtol || rn <= sc->atol)) {
        sc->reason = rn / sc->rnorm0 <= sc->rtol ? 2 : 3;
        sc->done = 1;
      }
    } break;
    case TSX_STAGE_ALPHA: {
      const double d1 = sc->red[0];
      if (d1 == 0.0 || d1 != d1) {
        sc->reason = d1 != d1 ? -9 : -5;
        sc->done = 1;
      } else {
        sc->alpha = sc->rho / d1;
      }
    } break;
    case TSX_STAGE_OMEGA: {
      const double ts = sc->red[0], tt = sc->red[2];  // SpMV_2 runs with w = s: slot0 = (s,t), slot2 = (t,t)
      sc->omega = tt == 0.0 ? 0.0 : ts / tt;
    } break;
    case TSX_STAGE_RHO: {
      sc->rho_old = sc->rho;
      sc->rho = sc->red[0];
      const double rn = sqrt(sc->red[1]);
      sc->rnorm = rn;
      sc->its += 1;
      if (sc->nhist < 100) sc->hist[sc->nhist++] = rn;
      int reason = 0;
      if (rn / sc->rnorm0 <= sc->rtol) reason = 2;
      else if (rn <= sc->atol) reason = 3;
      else if (sc->its > sc->maxit) reason = -3;
      else if (rn / sc->rnorm0 >= sc->dtol) reason = -4;
      else if (rn != rn) reason = -9;
      else if (sc->rho == 0.0 || sc->omega == 0.0) reason = -5;
      if (reason) {
        sc->reason = reason;
        sc->done = 1;
      } else {
        sc->beta = (sc->rho / sc->rho_old) * (sc->alpha / sc->omega);
      }
    } break;
  }
}

// ------------------------------------------------------------------------------------------------
// Layout conversion reference <-> internal (see tsx_internal.hpp).  One thread per (level, i, j).
// Streams that leave the *neighbouring* cell across the low x / low y face of the owned block belong to
// the neighbour rank's cell: they are routed through the halo buffers.
//   import: ref value of +x stream at face i=0  -> sendW (west rank stores it at its cell xm-1)
//           ref value of +y stream at face j=0  -> sendS
//   export: ref value of +x stream at face i=0  <- recvW (== SpMV halo of x), same for y
template <int NTOP, int NSIDE, bool EXPORT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_convert_vec(TsxGeo g, double *__restrict__ ref,
                                                               double *__restrict__ v, double *__restrict__ bufW,
                                                               double *__restrict__ bufS) {
  constexpr int D = NTOP + 2 * NSIDE;
  const int L = g.Nz + 1, xm = g.xm, ym = g.ym, Nz = g.Nz, ncol = g.ncol;
  const long long Nc = g.Nc;
  const long long total = (long long)L * ncol;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < total; q += (long long)gridDim.x * TSX_BLOCK) {
    // q enumerates (i fastest, j, k) so that internal accesses coalesce
    const int i = (int)(q % xm);
    const long long t = q / xm;
    const int j = (int)(t % ym);
    const int k = (int)(t / ym);
    const int col = j * xm + i;
    double *__restrict__ rp = ref + (size_t)D * ((size_t)k + (size_t)L * ((size_t)i + (size_t)xm * j));
    double *__restrict__ vt = v + (size_t)D * Nc;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      double *loc = nullptr;  // internal location of ref (d,k,i,j)
      if (d < NTOP) {
        if (tsx_inward(d)) loc = k >= 1 ? v + (size_t)d * Nc + ((size_t)(k - 1) * ym + j) * xm + i : vt + (size_t)d * ncol + col;
        else loc = k < Nz ? v + (size_t)d * Nc + ((size_t)k * ym + j) * xm + i : vt + (size_t)d * ncol + col;
      } else if (k == Nz) {
        loc = vt + (size_t)d * ncol + col;
      } else if (d < NTOP + NSIDE) {
        const int qd = d - NTOP;
        if (!tsx_inward(qd)) loc = v + (size_t)d * Nc + ((size_t)k * ym + j) * xm + i;
        else if (i > 0) loc = v + (size_t)d * Nc + ((size_t)k * ym + j) * xm + (i - 1);
        else if (g.wrap_x) loc = v + (size_t)d * Nc + ((size_t)k * ym + j) * xm + (xm - 1);
        else loc = bufW + ((size_t)(qd >> 1) * Nz + k) * ym + j;
      } else {
        const int qd = d - NTOP - NSIDE;
        if (!tsx_inward(qd)) loc = v + (size_t)d * Nc + ((size_t)k * ym + j) * xm + i;
        else if (j > 0) loc = v + (size_t)d * Nc + ((size_t)k * ym + (j - 1)) * xm + i;
        else if (g.wrap_y) loc = v + (size_t)d * Nc + ((size_t)k * ym + (ym - 1)) * xm + i;
        else loc = bufS + ((size_t)(qd >> 1) * Nz + k) * xm + i;
      }
      if (EXPORT) rp[d] = *loc;
      else *loc = rp[d];
    }
  }
}

// after an import exchange: +x streams received from the east rank (its face i=0) land on my cells xm-1,
// +y streams received from the north rank land on my cells ym-1.
template <int NTOP, int NSIDE>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_import_unpack(TsxGeo g, double *__restrict__ v,
                                                                 const double *__restrict__ recvE,
                                                                 const double *__restrict__ recvN) {
  const int xm = g.xm, ym = g.ym, Nz = g.Nz;
  const long long Nc = g.Nc;
  const long long nx = (long long)(NSIDE / 2) * Nz * ym, ny = (long long)(NSIDE / 2) * Nz * xm;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < nx + ny; q += (long long)gridDim.x * TSX_BLOCK) {
    if (q < nx) {
      const int j = (int)(q % ym);
      const int k = (int)((q / ym) % Nz);
      const int slot = (int)(q / ((long long)ym * Nz));
      const int d = NTOP + 2 * slot + 1;
      if (recvE) v[(size_t)d * Nc + ((size_t)k * ym + j) * xm + (xm - 1)] = recvE[q];
    } else {
      const long long p = q - nx;
      const int i = (int)(p % xm);
      const int k = (int)((p / xm) % Nz);
      const int slot = (int)(p / ((long long)xm * Nz));
      const int d = NTOP + NSIDE + 2 * slot + 1;
      if (recvN) v[(size_t)d * Nc + ((size_t)k * ym + (ym - 1)) * xm + i] = recvN[p];
    }
  }
}

// SpMV halo pack (exchange_diffuse_boundary, src/pprts_explicit.F90:769-800, in dst-owned storage):
//   sendE = +x streams of my cells i = xm-1   (east rank reads them as its west halo)
//   sendW = -x streams of my cells i = 0
//   sendN = +y streams of my cells j = ym-1 ; sendS = -y streams of my cells j = 0
template <int NTOP, int NSIDE, typename XT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_halo_pack(TsxGeo g, const XT *__restrict__ v,
                                                             double *__restrict__ sendW, double *__restrict__ sendE,
                                                             double *__restrict__ sendS, double *__restrict__ sendN,
                                                             const int *__restrict__ done) {
  if (done && *done) return;
  const int xm = g.xm, ym = g.ym, Nz = g.Nz;
  const long long Nc = g.Nc;
  const long long nx = (long long)(NSIDE / 2) * Nz * ym, ny = (long long)(NSIDE / 2) * Nz * xm;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < nx + ny; q += (long long)gridDim.x * TSX_BLOCK) {
    if (q < nx) {
      const int j = (int)(q % ym);
      const int k = (int)((q / ym) % Nz);
      const int slot = (int)(q / ((long long)ym * Nz));
      const size_t row = ((size_t)k * ym + j) * xm;
      if (!g.wrap_x) {
        sendE[q] = (double)v[(size_t)(NTOP + 2 * slot + 1) * Nc + row + (xm - 1)];
        sendW[q] = (double)v[(size_t)(NTOP + 2 * slot) * Nc + row];
      }
    } else {
      const long long p = q - nx;
      const int i = (int)(p % xm);
      const int k = (int)((p / xm) % Nz);
      const int slot = (int)(p / ((long long)xm * Nz));
      if (!g.wrap_y) {
        sendN[p] = (double)v[(size_t)(NTOP + NSIDE + 2 * slot + 1) * Nc + ((size_t)k * ym + (ym - 1)) * xm + i];
        sendS[p] = (double)v[(size_t)(NTOP + NSIDE + 2 * slot) * Nc + (size_t)k * ym * xm + i];
      }
    }
  }
}

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

// diffuse coefficients for every 3-D cell -> planes C[q*Nc + cell] (float).  Inputs in reference layout (k fastest).
template <int DD>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_lut_diff2diff(TsxGeo g, TsxLutDev L, const double *__restrict__ kabs,
                                                                 const double *__restrict__ ksca, const double *__restrict__ gg,
                                                                 const double *__restrict__ dz, double dx,
                                                                 const uint8_t *__restrict__ l1d, float *__restrict__ C) {
  const int xm = g.xm, ym = g.ym, Nz = g.Nz;
  const long long Nc = g.Nc;
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const int i = (int)(c % xm);
    const long long t = c / xm;
    const int j = (int)(t % ym);
    const int k = (int)(t / ym);
    if (l1d[k]) continue;
    const size_t r = (size_t)k + (size_t)Nz * ((size_t)i + (size_t)xm * j);
    const double ka = kabs[r], ks = ksca[r], dzz = dz[r];
    // src/pprts_base.F90:1517-1533
    float aspect = (float)(dzz / dx);
    float w0 = (float)(ks / fmax(ka + ks, 2.220446049250313e-16));
    float tauz = (float)((ka + ks) * dzz);
    const float *ax = L.axes;
    aspect = fmaxf(ax[L.axis_off[2]], aspect);
    tauz = fmaxf(ax[L.axis_off[0]], fminf(ax[L.axis_off[0] + L.n[0] - 1], tauz));
    w0 = fmaxf(ax[L.axis_off[1]], fminf(ax[L.axis_off[1] + L.n[1] - 1], w0));
    const float sample[4] = {tauz, w0, aspect, (float)gg[r]};
    int ninterp;
    long long ofs_base, ioff_lo[4], ioff_hi[4];
    float wlo[4], whi[4];
    tsx_lut_weights<4>(L, sample, ninterp, ofs_base, ioff_lo, ioff_hi, wlo, whi);
    float acc[DD];
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
#pragma unroll
    for (int q = 0; q < DD; ++q) C[(size_t)q * Nc + c] = acc[q];
  }
}

// planes -> reference block layout (c fastest, then k, i, j), as real64: what solver%diff2diff holds
template <typename CT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_export_coeff(TsxGeo g, int DD, const CT *__restrict__ C,
                                                                double *__restrict__ ref) {
  const int xm = g.xm, ym = g.ym, Nz = g.Nz;
  const long long total = g.Nc * DD;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < total; q += (long long)gridDim.x * TSX_BLOCK) {
    const long long c = q % g.Nc;
    const int cc = (int)(q / g.Nc);
    const int i = (int)(c % xm);
    const long long t = c / xm;
    const int j = (int)(t % ym);
    const int k = (int)(t / ym);
    ref[(size_t)cc + (size_t)DD * ((size_t)k + (size_t)Nz * ((size_t)i + (size_t)xm * j))] = (double)C[q];
  }
}

// ------------------------------------------------------------------------------------------------
// Operator values: reference block layout (c = dst*D+src fastest, then k, i, j) -> one plane per c,
// x fastest.  LDS-tiled transpose so both sides coalesce.
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_import_coeff(TsxGeo g, int DD, int TI, const TIN *__restrict__ ref,
                                                                TOUT *__restrict__ C) {
  // one block per (j,k, tile of TI i): tile[TI][DD+1]
  extern __shared__ unsigned char smem_raw[];
  TOUT *tile = reinterpret_cast<TOUT *>(smem_raw);
  const int xm = g.xm, ym = g.ym, Nz = g.Nz;
  const int tiles_x = (xm + TI - 1) / TI;
  const long long nt = (long long)tiles_x * ym * Nz;
  for (long long b = blockIdx.x; b < nt; b += gridDim.x) {
    const int tx = (int)(b % tiles_x);
    const int j = (int)((b / tiles_x) % ym);
    const int k = (int)(b / ((long long)tiles_x * ym));
    const int i0 = tx * TI;
    const int ni = xm - i0 < TI ? xm - i0 : TI;
    for (int q = threadIdx.x; q < ni * DD; q += TSX_BLOCK) {
      const int ii = q / DD, c = q % DD;
      tile[ii * (DD + 1) + c] =
          (TOUT)ref[(size_t)c + (size_t)DD * ((size_t)k + (size_t)Nz * ((size_t)(i0 + ii) + (size_t)xm * j))];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < ni * DD; q += TSX_BLOCK) {
      const int c = q / ni, ii = q % ni;
      C[(size_t)c * g.Nc + ((size_t)k * ym + j) * xm + i0 + ii] = tile[ii * (DD + 1) + c];
    }
    __syncthreads();
  }
}

// flag[0] |= 1 if any value is not exactly representable in fp32
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_check_fp32_lossless(long long n, const double *__restrict__ v,
                                                                       int *__restrict__ flag) {
  int bad = 0;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * TSX_BLOCK) {
    const double a = v[q];
    if ((double)(float)a != a) bad = 1;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// (k,i,j) reference scalar field (z fastest) -> cell-indexed (i fastest)
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_import_cellfield(TsxGeo g, const double *__restrict__ ref,
                                                                    double *__restrict__ out) {
  const int xm = g.xm, ym = g.ym, Nz = g.Nz;
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < g.Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const int i = (int)(c % xm);
    const long long t = c / xm;
    const int j = (int)(t % ym);
    const int k = (int)(t / ym);
    out[c] = ref[(size_t)k + (size_t)Nz * ((size_t)i + (size_t)xm * j)];
  }
}

__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_copy16(long long n, const float4 *__restrict__ a, float4 *__restrict__ b) {
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * TSX_BLOCK) b[q] = a[q];
}
