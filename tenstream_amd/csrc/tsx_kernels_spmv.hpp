// tsx_kernels_spmv.hpp -- the operator apply y = (I - T) x and its halo pack kernel (see tsx_dev.hpp)
#pragma once
#include <type_traits>

#include "tsx_dev.hpp"

// part: 0 = every cell; 1 = interior only (cells whose gather touches no received face: launched while the exchange is
// in flight); 2 = frame only (the complement, enumerated directly: per level the first/last row and the first/last
// group of every other row).  Partial sums of launch 2 go behind those of launch 1 (partials pointer is offset).
// HALO: some face of the rank is not a periodic self-neighbour (edge threads then read the received face buffers);
// HAS1D: some layer is 1-D.  Both are kernel-uniform and compiled out in the common case.  The gather is branch-free
// (offset / pointer selects, unconditional loads): a conditional load ends a basic block and forces an s_waitcnt.
// IDX: the blocks are stored once per *distinct* block (tsx_dedup.hip): C holds planes over nent entries and cidx[c] is
// the entry of cell c.  Cells that share an entry read the same addresses (one cache line per wave instruction), runs of
// unique cells have consecutive entries; same numbers, same order of operations as the dense planes.
// shared blocks are stored entry-major for the operator, Ce[id * D*D + dst * D + src]: a cell's row of D coefficients is one
// contiguous run (D / 2 eight-byte loads), and every byte of a fetched line belongs to the cell that fetched it -- with
// plane-major storage a cloudy cell touched D*D lines that it shared with its neighbours in the table, and those lines were
// fetched about three times over (L2 does not hold them between the workgroups of adjacent rows)
template <int CPT> struct TsxIdx;
template <> struct TsxIdx<1> {
  static __device__ __forceinline__ void ids(const int *p, int (&o)[1]) { o[0] = p[0]; }
  template <int D> static __device__ __forceinline__ void ld_row(const float *Ce, const int (&id)[1], int d, float (&cf)[D]) {
    const float2 *r = reinterpret_cast<const float2 *>(Ce + (size_t)id[0] * (D * D) + d * D);
#pragma unroll
    for (int h = 0; h < D / 2; ++h) {
      const float2 v = r[h];
      cf[2 * h] = v.x;
      cf[2 * h + 1] = v.y;
    }
  }
};
template <> struct TsxIdx<2> {
  static __device__ __forceinline__ void ids(const int *p, int (&o)[2]) {
    const int2 v = *reinterpret_cast<const int2 *>(p);
    o[0] = v.x;
    o[1] = v.y;
  }
  template <int D> static __device__ __forceinline__ void ld_row(const float *Ce, const int (&id)[2], int d, float2 (&cf)[D]) {
    const float2 *r0 = reinterpret_cast<const float2 *>(Ce + (size_t)id[0] * (D * D) + d * D);
    const float2 *r1 = reinterpret_cast<const float2 *>(Ce + (size_t)id[1] * (D * D) + d * D);
#pragma unroll
    for (int h = 0; h < D / 2; ++h) {
      const float2 a = r0[h], b = r1[h];
      cf[2 * h] = make_float2(a.x, b.x);
      cf[2 * h + 1] = make_float2(a.y, b.y);
    }
  }
};

// YT: storage of the result (float: the fp32 Krylov vectors v, t of tsx_ksp_opts.fp32_directions = 2; the fused dots then use
// the value as stored)
template <int NTOP, int NSIDE, typename CT, int FUSE, int CPT, typename XT, typename WT, bool HALO, bool HAS1D, bool IDX = false,
          typename YT = double>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_spmv_w(
    TsxGeo g, const CT *__restrict__ C, const int *__restrict__ cidx, long long nent, const uint8_t *__restrict__ l1d,
    const double *__restrict__ a11,
    const double *__restrict__ a12, const double *__restrict__ albedo, const XT *__restrict__ x,
    YT *__restrict__ y, const XT *__restrict__ hW, const XT *__restrict__ hE,
    const XT *__restrict__ hS, const XT *__restrict__ hN, const WT *__restrict__ w,
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
  YT *__restrict__ yt = y + (size_t)D * Nc;
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
          const double h = (double)hW[((size_t)slot * Nz + k) * ym + j];
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
          const double h = (double)hE[((size_t)slot * Nz + k) * ym + j];
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
    int eid[CPT];
    int id0 = 0;
    bool wuni = false;
    if constexpr (IDX) {
      // all cells of this wave share one block (the usual case inside a homogeneous background): its 100 coefficients
      // come through the scalar cache into SGPRs instead of 64 identical vector loads each
      TsxIdx<CPT>::ids(cidx + c, eid);
      id0 = __builtin_amdgcn_readfirstlane(eid[0]);
      bool same = true;
#pragma unroll
      for (int m = 0; m < CPT; ++m) same &= eid[m] == id0;
      wuni = __all(same);
    }
    auto rows = [&](auto uni_tag) {
      constexpr bool UNI = decltype(uni_tag)::value;
      using CH = typename std::conditional<UNI, float, CV>::type;  // a coefficient as held: wave-uniform scalar / per lane
      CH cfc[D], cfn[D];
      XV xoc, xon;
      WV wc, wn;
      auto issue_row = [&](int d, CH(&cf)[D], XV &xo_, WV &w_) {
        if constexpr (UNI) {
#pragma unroll
          for (int s2 = 0; s2 < D; ++s2) cf[s2] = C[(size_t)id0 * (D * D) + d * D + s2];
        } else if constexpr (IDX) {
          TsxIdx<CPT>::template ld_row<D>(C, eid, d, cf);
        } else {
#pragma unroll
          for (int s2 = 0; s2 < D; ++s2) cf[s2] = TsxRaw<CT, CPT>::ld(C + (size_t)(d * D + s2) * Nc + c);
        }
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
            if constexpr (UNI) {
#pragma unroll
              for (int m = 0; m < CPT; ++m) cf[m] = (double)cfc[s2];
            } else {
              TsxRaw<CT, CPT>::cvt(cfc[s2], cf);
            }
#pragma unroll
            for (int m = 0; m < CPT; ++m) acc[m] += cf[m] * xs[s2][m];
          }
#pragma unroll
          for (int m = 0; m < CPT; ++m) acc[m] = xo[m] - acc[m];
        }
        if constexpr (std::is_same<YT, float>::value) {  // the dots see the value as stored
#pragma unroll
          for (int m = 0; m < CPT; ++m) acc[m] = (double)(float)acc[m];
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
    };
    if constexpr (IDX) {
      if (wuni) rows(std::true_type{});
      else rows(std::false_type{});
    } else {
      rows(std::false_type{});
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
        if constexpr (std::is_same<YT, float>::value) {
#pragma unroll
          for (int m = 0; m < CPT; ++m) yv[m] = (double)(float)yv[m];
        }
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

// SpMV halo pack (exchange_diffuse_boundary, src/pprts_explicit.F90:769-800, in dst-owned storage); the messages carry
// the vector's own precision (fp32 for the preconditioned directions: half the bytes):
//   sendE = +x streams of my cells i = xm-1   (east rank reads them as its west halo)
//   sendW = -x streams of my cells i = 0
//   sendN = +y streams of my cells j = ym-1 ; sendS = -y streams of my cells j = 0
template <int NTOP, int NSIDE, typename XT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_halo_pack(TsxGeo g, const XT *__restrict__ v,
                                                             XT *__restrict__ sendW, XT *__restrict__ sendE,
                                                             XT *__restrict__ sendS, XT *__restrict__ sendN,
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
        sendE[q] = v[(size_t)(NTOP + 2 * slot + 1) * Nc + row + (xm - 1)];
        sendW[q] = v[(size_t)(NTOP + 2 * slot) * Nc + row];
      }
    } else {
      const long long p = q - nx;
      const int i = (int)(p % xm);
      const int k = (int)((p / xm) % Nz);
      const int slot = (int)(p / ((long long)xm * Nz));
      if (!g.wrap_y) {
        sendN[p] = v[(size_t)(NTOP + NSIDE + 2 * slot + 1) * Nc + ((size_t)k * ym + (ym - 1)) * xm + i];
        sendS[p] = v[(size_t)(NTOP + NSIDE + 2 * slot) * Nc + (size_t)k * ym * xm + i];
      }
    }
  }
}
