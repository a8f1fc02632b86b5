// tsx_kernels_pc.hpp -- column-block preconditioner kernels (see tsx_dev.hpp)
#pragma once
#include "tsx_dev.hpp"
#include "tsx_pack.hpp"

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

// ---- 3_10 preconditioner on a *packed* reduced-precision copy of the transport blocks (fp32 directions only) -----
// The sweep above is paced by memory latency, not bytes: one wave can only track 63 outstanding vector-memory
// operations (s_waitcnt vmcnt is 6 bits), and with one 4-byte load per coefficient a single level already needs > 100.
// Here the 100 coefficients of a cell are regrouped into 8 records of 16 B in the order the two sweeps consume them,
// P[(grp * Nc + cell)] as uint4: a level costs 2 (up) + 6 (down) coefficient loads of 16 B per lane, so several levels fit
// under the counter and the sweeps are software-pipelined PU / PD levels deep.  The 20 coefficients of the column block
// itself (the exact part of M) are fp16; the 80 couplings to neighbouring columns -- which only enter the right-hand side
// with lagged / Gauss-Seidel values -- are OCP fp8 e4m3 scaled by 64 (measured: same iteration counts as fp16).
//   grp 0: tuu rud rdu tdd (fp16) | c(y_q->0) c(y_q->1), q = 0..3 (fp8)       (y_q = src dof 6+q, x_q = src dof 2+q)
//   grp 1: c(x_q->0) c(x_q->1), q = 0..3 (fp8) | pad
//   grp 2: c(0 -> side d), d = 2..9 (fp16)        grp 3: c(1 -> side d) (fp16)
//   grp 4, 5: c(y_q -> side 2+dd), byte 4 dd + q (fp8)       grp 6, 7: c(x_q -> side 2+dd) (fp8)
// 8_16 (D = 16): 24 records.  t = top dst 0..7, d = side dst 8..15, y_q = src 12+q, x_q = src 8+q.
//   grp 0..7:    c(src 0..7 -> top dst t = grp), fp16                (Tuu/Rud/Rdu/Tdd interleaved by stream parity)
//   grp 8, 9:    c(y_q -> t), byte 4 t + q, fp8        grp 10, 11: c(x_q -> t), fp8
//   grp 12..19:  c(src 0..7 -> side dst 8 + dd), fp16
//   grp 20, 21:  c(y_q -> 8 + dd), byte 4 dd + q, fp8  grp 22, 23: c(x_q -> 8 + dd), fp8
constexpr int TSX_P16H_GROUPS = 24;

template <typename CT, int NTOP>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_pack_p16(long long Nc, const CT *__restrict__ C, uint4 *__restrict__ P, int split_xm,
                                                            int split_ym) {
  constexpr int NG = NTOP == 2 ? TSX_P16_GROUPS : TSX_P16H_GROUPS;
  const long long n = Nc * NG;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * TSX_BLOCK) {
    const int grp = (int)(q / Nc);
    const long long c = q - (long long)grp * Nc;
    auto cf = [&](int dst, int src) { return (float)C[(size_t)(dst * (NTOP + 8) + src) * Nc + c]; };
    uint4 v = make_uint4(0, 0, 0, 0);
    if (NTOP == 2) {
      if (grp == 0) {
        v.x = tsx_to_h2(cf(0, 0), cf(0, 1));
        v.y = tsx_to_h2(cf(1, 0), cf(1, 1));
        v.z = tsx_to_fp8x4(cf(0, 6), cf(1, 6), cf(0, 7), cf(1, 7));
        v.w = tsx_to_fp8x4(cf(0, 8), cf(1, 8), cf(0, 9), cf(1, 9));
      } else if (grp == 1) {
        v.x = tsx_to_fp8x4(cf(0, 2), cf(1, 2), cf(0, 3), cf(1, 3));
        v.y = tsx_to_fp8x4(cf(0, 4), cf(1, 4), cf(0, 5), cf(1, 5));
      } else if (grp == 2 || grp == 3) {
        const int s = grp - 2;
        v.x = tsx_to_h2(cf(2, s), cf(3, s));
        v.y = tsx_to_h2(cf(4, s), cf(5, s));
        v.z = tsx_to_h2(cf(6, s), cf(7, s));
        v.w = tsx_to_h2(cf(8, s), cf(9, s));
      } else {
        const int s0 = grp < 6 ? 6 : 2, d0 = 2 + 4 * ((grp - 4) & 1);  // y sources 6..9 / x sources 2..5; side dst d0..d0+3
        v.x = tsx_to_fp8x4(cf(d0 + 0, s0), cf(d0 + 0, s0 + 1), cf(d0 + 0, s0 + 2), cf(d0 + 0, s0 + 3));
        v.y = tsx_to_fp8x4(cf(d0 + 1, s0), cf(d0 + 1, s0 + 1), cf(d0 + 1, s0 + 2), cf(d0 + 1, s0 + 3));
        v.z = tsx_to_fp8x4(cf(d0 + 2, s0), cf(d0 + 2, s0 + 1), cf(d0 + 2, s0 + 2), cf(d0 + 2, s0 + 3));
        v.w = tsx_to_fp8x4(cf(d0 + 3, s0), cf(d0 + 3, s0 + 1), cf(d0 + 3, s0 + 2), cf(d0 + 3, s0 + 3));
      }
    } else {
      const bool up = grp < 12;
      const int g2 = up ? grp : grp - 12, d0 = up ? 0 : 8;  // the two halves of the layout are built alike
      if (g2 < 8) {
        v.x = tsx_to_h2(cf(d0 + g2, 0), cf(d0 + g2, 1));
        v.y = tsx_to_h2(cf(d0 + g2, 2), cf(d0 + g2, 3));
        v.z = tsx_to_h2(cf(d0 + g2, 4), cf(d0 + g2, 5));
        v.w = tsx_to_h2(cf(d0 + g2, 6), cf(d0 + g2, 7));
      } else {
        const int s0 = g2 < 10 ? 12 : 8, t0 = d0 + 4 * ((g2 - 8) & 1);
        v.x = tsx_to_fp8x4(cf(t0 + 0, s0), cf(t0 + 0, s0 + 1), cf(t0 + 0, s0 + 2), cf(t0 + 0, s0 + 3));
        v.y = tsx_to_fp8x4(cf(t0 + 1, s0), cf(t0 + 1, s0 + 1), cf(t0 + 1, s0 + 2), cf(t0 + 1, s0 + 3));
        v.z = tsx_to_fp8x4(cf(t0 + 2, s0), cf(t0 + 2, s0 + 1), cf(t0 + 2, s0 + 2), cf(t0 + 2, s0 + 3));
        v.w = tsx_to_fp8x4(cf(t0 + 3, s0), cf(t0 + 3, s0 + 1), cf(t0 + 3, s0 + 2), cf(t0 + 3, s0 + 3));
      }
    }
    if (split_xm > 0) {  // colour-split order for the red-black preconditioner (tsx_split_col)
      const int i = (int)(c % split_xm);
      const long long t = c / split_xm;
      const int j = (int)(t % split_ym);
      P[(long long)grp * Nc + (t / split_ym) * ((long long)split_xm * split_ym) + tsx_split_col(i, j, split_xm)] = v;
    } else {
      P[q] = v;
    }
  }
}

struct TsxUpRaw {
  uint4 c0;        // grp 0
  uint2 c1;        // grp 1 (first 8 bytes)
  float ru, rd;
  double t11, t12;
  float zy[4], zx[4];
};
struct TsxDnRaw {
  uint4 cu, cv, cy[2], cx[2];
  float4 t;        // Gw_k, GT_k, A_{k+1}, B_{k+1}
  float rs[8];
  float zy[4], zx[4];
};

// LDST: the sweep temporaries (16 B per level and column) live in LDS instead of global memory -- a lane only ever touches
// its own column's slots, so no barrier is needed; 64 columns x Nz levels x 16 B (64 KiB at Nz = 64).
template <int ROWS, bool GS, bool HAS1D, bool XL, bool LDST>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void tsx_k_pc_column_p16(
    TsxGeo g, const uint4 *__restrict__ P, const uint8_t *__restrict__ l1d, const double *__restrict__ a11,
    const double *__restrict__ a12, const double *__restrict__ albedo, const float *__restrict__ r, float *__restrict__ z,
    const float *__restrict__ zc, const float *__restrict__ zx, float4 *__restrict__ tmp, const int *__restrict__ done) {
  constexpr int D = 10, NTOP = 2, NSIDE = 4;
  constexpr int PU = 4, PD = 2;  // prefetch depth of the upward / downward sweep (levels)
  extern __shared__ float4 tsx_pc_lds[];
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
  long long offN = (jrow + 1 < g.ym) ? (long long)g.xm : (seam ? -(long long)(g.ym - 1) * g.xm : 0);
  long long offS = (jrow > 0) ? -(long long)g.xm : (seam ? (long long)(g.ym - 1) * g.xm : 0);
  const int icol = col % g.xm;
  long long offE = (icol + 1 < g.xm) ? 1 : (g.wrap_x ? -(long long)(g.xm - 1) : 0);
  long long offW = (icol > 0) ? -1 : (g.wrap_x ? (long long)(g.xm - 1) : 0);
  if (g.pc_tile_x > 0) {  // analysis knob: behave like a rank of pc_tile_x x pc_tile_y columns
    if ((icol + 1) % g.pc_tile_x == 0) offE = 0;
    if (icol % g.pc_tile_x == 0) offW = 0;
  }
  if (g.pc_tile_y > 0) {
    if ((jrow + 1) % g.pc_tile_y == 0) offN = 0;
    if (jrow % g.pc_tile_y == 0) offS = 0;
  }
  const float *__restrict__ rt = r + (size_t)D * Nc;
  float *__restrict__ zt = z + (size_t)D * Nc;
  const double albc = albedo[col], rsurf = rt[col];

  // loads only: nothing here depends on loaded data, so that the whole record of a level is in flight at once
  auto load_up = [&](int k) {
    TsxUpRaw u;
    const size_t c = (size_t)k * ncol + col;
    u.c0 = P[(size_t)0 * Nc + c];
    if (XL) u.c1 = *reinterpret_cast<const uint2 *>(P + (size_t)1 * Nc + c);
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
    const tsx_h4 tt = __builtin_bit_cast(tsx_h4, make_uint2(u.c0.x, u.c0.y));
    double tuu = (double)tt[0], rud = (double)tt[1], rdu = (double)tt[2], tdd = (double)tt[3];
    float gu8 = 0.0f, gd8 = 0.0f;  // coupling sums in fp8 units (x TSX_FP8_SCALE), fp32 accumulation
    if (GS) {
      float ca[4], cb[4];  // [c(y0->0) c(y0->1) c(y1->0) c(y1->1)], [y2, y3]
      tsx_fp8x4(u.c0.z, ca);
      tsx_fp8x4(u.c0.w, cb);
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) {
        const float zv = (tsx_inward(q) ? offS : offN) ? u.zy[q] : 0.0f;  // select: the unused slot may hold NaN
        gu8 += (q < 2 ? ca[2 * q] : cb[2 * (q - 2)]) * zv;
        gd8 += (q < 2 ? ca[2 * q + 1] : cb[2 * (q - 2) + 1]) * zv;
      }
    }
    if (XL) {
      float ca[4], cb[4];
      tsx_fp8x4(u.c1.x, ca);
      tsx_fp8x4(u.c1.y, cb);
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) {
        const float zv = (tsx_inward(q) ? offW : offE) ? u.zx[q] : 0.0f;
        gu8 += (q < 2 ? ca[2 * q] : cb[2 * (q - 2)]) * zv;
        gd8 += (q < 2 ? ca[2 * q + 1] : cb[2 * (q - 2) + 1]) * zv;
      }
    }
    double gu = (double)gu8 * (1.0 / TSX_FP8_SCALE), gd = (double)gd8 * (1.0 / TSX_FP8_SCALE);
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
    if (LDST) tsx_pc_lds[k * 64 + threadIdx.x] = make_float4((float)Gw, (float)GT, (float)A, (float)B);
    else tmp[c] = make_float4((float)Gw, (float)GT, (float)A, (float)B);
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
    d.cu = P[(size_t)2 * Nc + c];
    d.cv = P[(size_t)3 * Nc + c];
    d.t = LDST ? tsx_pc_lds[k * 64 + threadIdx.x] : tmp[c];  // prefetched with the rest of the level: off the recurrence
#pragma unroll
    for (int q = 0; q < 8; ++q) d.rs[q] = r[(size_t)(NTOP + q) * Nc + c];
    if (GS) {
#pragma unroll
      for (int m = 0; m < 2; ++m) d.cy[m] = P[(size_t)(4 + m) * Nc + c];
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) d.zy[q] = zc[(size_t)(NTOP + NSIDE + q) * Nc + c + (tsx_inward(q) ? offS : offN)];
    }
    if (XL) {
#pragma unroll
      for (int m = 0; m < 2; ++m) d.cx[m] = P[(size_t)(6 + m) * Nc + c];
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
    const float4 t = d.t;
    const double Vn = (double)t.x + (double)t.y * V;
    const double Un = (double)t.z * Vn + (double)t.w;
    z[c] = (float)U;
    z[(size_t)Nc + c] = (float)Vn;
    float zy[NSIDE], zq[NSIDE];
    if (GS) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) zy[q] = (tsx_inward(q) ? offS : offN) ? d.zy[q] : 0.0f;
    }
    if (XL) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) zq[q] = (tsx_inward(q) ? offW : offE) ? d.zx[q] : 0.0f;
    }
    const tsx_h8 hcu = __builtin_bit_cast(tsx_h8, d.cu), hcv = __builtin_bit_cast(tsx_h8, d.cv);
    const unsigned wy[8] = {d.cy[0].x, d.cy[0].y, d.cy[0].z, d.cy[0].w, d.cy[1].x, d.cy[1].y, d.cy[1].z, d.cy[1].w};
    const unsigned wx[8] = {d.cx[0].x, d.cx[0].y, d.cx[0].z, d.cx[0].w, d.cx[1].x, d.cx[1].y, d.cx[1].z, d.cx[1].w};
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) {
      double acc = (double)hcu[dd] * Un + (double)hcv[dd] * V;
      float a8 = 0.0f;  // couplings: fp8 units, fp32 accumulation
      if (GS) {
        float cq[4];
        tsx_fp8x4(wy[dd], cq);
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) a8 += cq[q] * zy[q];
      }
      if (XL) {
        float cq[4];
        tsx_fp8x4(wx[dd], cq);
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) a8 += cq[q] * zq[q];
      }
      acc += (double)a8 * (1.0 / TSX_FP8_SCALE);
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

// ---- red-black (checkerboard) ordering of the column blocks: all four lateral neighbours of a column have the other
// colour, so every pass is a true Gauss-Seidel step in x *and* y with the other colour's latest values (zebra rows only
// get that in y and lag in x).  Measured on the CPU model (tests/studies/pc_study.py): the same iteration count with 2/3 of the
// passes, or ~27 % fewer iterations at the same number of passes.  Lanes run over every other column, so everything the
// preconditioner owns -- packed blocks P, fp32 right-hand side r, its iterate zs -- is stored colour-split: within a row
// the xm/2 columns of colour 0 first, then colour 1 (tsx_split_col); loads and stores stay contiguous.
// rbc = colour of this pass, (i + j) & 1.
// MODE 0: an intermediate pass -- its result is only ever read as a neighbour value by later passes, i.e. the 8 side streams,
//         stored as bf16 in zb (the top streams and the tail rows are not stored at all); reads its neighbours from zb.
// MODE 1: the last pass of the first colour: all 10 streams in fp32 to z (colour-split); neighbours from zb.
// MODE 2: the very last pass: neighbours and the row partner's final values from z (fp32), result for both colours as
//         aligned pairs in the Krylov layout zfin.
struct TsxUpRawB {
  uint4 c0;
  uint2 c1;
  float ru, rd;
  double t11, t12;
  unsigned zy[4], zx[4];  // neighbour values as loaded: fp32 bits (MODE 2) or bf16 (MODE 0/1)
};
struct TsxDnRawB {
  uint4 cu, cv, cy[2], cx[2];
  float4 t;
  float rs[8];
  unsigned zy[4], zx[4];
  float pz[10];  // MODE 2: the row partner's final values (prefetched with the level, not loaded at the store)
};

template <bool GS, bool HAS1D, bool LDST, int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void tsx_k_pc_column_rb(
    TsxGeo g, const uint4 *__restrict__ P, const uint8_t *__restrict__ l1d, const double *__restrict__ a11,
    const double *__restrict__ a12, const double *__restrict__ albedo, const float *__restrict__ r, float *__restrict__ z,
    unsigned short *__restrict__ zb, float *__restrict__ zfin, float4 *__restrict__ tmp, const int *__restrict__ done,
    int rbc) {
  constexpr int D = 10, NTOP = 2, NSIDE = 4;
  constexpr int PU = 4, PD = 2;  // prefetch depth of the upward / downward sweep (levels)
  constexpr bool XL = GS;        // x and y couplings alike
  constexpr bool FINAL = MODE == 2;
  using real = float;  // the sweeps run in fp32: M^-1 is an approximation anyway (fp8 couplings), fp64 only costs issue slots
  using TsxUpRaw = TsxUpRawB;
  using TsxDnRaw = TsxDnRawB;
  auto nbr_ld = [&](size_t idx) -> unsigned { return MODE == 2 ? __float_as_uint(z[idx]) : (unsigned)zb[idx]; };
  auto nbr_val = [](unsigned v) -> float { return MODE == 2 ? __uint_as_float(v) : __uint_as_float(v << 16); };
  extern __shared__ float4 tsx_pc_lds[];
  if (done && *done) return;
  const int h = g.xm >> 1;
  const int t_ = blockIdx.x * 64 + threadIdx.x;
  if (t_ >= g.ym * h) return;
  const long long Nc = g.Nc;
  const int Nz = g.Nz, ncol = g.ncol;
  const int jrow = t_ / h, qh = t_ - jrow * h;
  const int par = (jrow + rbc) & 1;
  const int icol = 2 * qh + par;
  const int ncl = jrow * g.xm + icol;                 // natural column index (albedo, a11/a12, Krylov-layout output)
  const int col = jrow * g.xm + rbc * h + qh;         // colour-split column index (P, r, z, zc)
  // neighbours (other colour) in split space; 0 = no neighbour (rank face / tile edge)
  const long long oc = (long long)(1 - 2 * rbc) * h;  // from my colour's half of the row to the other one
  const int jn = jrow + 1 < g.ym ? jrow + 1 : (g.wrap_y ? 0 : -1), js = jrow > 0 ? jrow - 1 : (g.wrap_y ? g.ym - 1 : -1);
  const int qw = par ? qh : (qh > 0 ? qh - 1 : (g.wrap_x ? h - 1 : -1)), qe = par ? (qh + 1 < h ? qh + 1 : (g.wrap_x ? 0 : -1)) : qh;
  long long offN = jn >= 0 ? (long long)(jn - jrow) * g.xm + oc : 0;
  long long offS = js >= 0 ? (long long)(js - jrow) * g.xm + oc : 0;
  long long offE = qe >= 0 ? oc + (qe - qh) : 0;
  long long offW = qw >= 0 ? oc + (qw - qh) : 0;
  // FINAL (the very last pass): this lane's value and its row partner's (columns 2q, 2q+1: one of each colour; the
  // partner's final value sits at the same q in the other colour's half, offset oc) go out as one aligned float2 in the
  // Krylov layout -- contiguous stores instead of two stride-2 passes
  const int ncp = jrow * g.xm + 2 * qh;  // natural index of the pair's first column
  auto cn0 = [&](int k) { return (size_t)k * ncol + ncp; };
  auto wpair = [&](float *dst, float mine, float partner) {
    *reinterpret_cast<float2 *>(dst) = par ? make_float2(partner, mine) : make_float2(mine, partner);
  };
  if (g.pc_tile_x > 0) {  // analysis knob: behave like a rank of pc_tile_x x pc_tile_y columns
    if ((icol + 1) % g.pc_tile_x == 0) offE = 0;
    if (icol % g.pc_tile_x == 0) offW = 0;
  }
  if (g.pc_tile_y > 0) {
    if ((jrow + 1) % g.pc_tile_y == 0) offN = 0;
    if (jrow % g.pc_tile_y == 0) offS = 0;
  }
  const float *__restrict__ rt = r + (size_t)D * Nc;
  float *__restrict__ zt = z + (size_t)D * Nc;
  float *__restrict__ zft = FINAL ? zfin + (size_t)D * Nc : nullptr;
  const real albc = (real)albedo[ncl], rsurf = rt[col];

  // loads only: nothing here depends on loaded data, so that the whole record of a level is in flight at once
  auto load_up = [&](int k) {
    TsxUpRaw u;
    const size_t c = (size_t)k * ncol + col;
    u.c0 = P[(size_t)0 * Nc + c];
    if (XL) u.c1 = *reinterpret_cast<const uint2 *>(P + (size_t)1 * Nc + c);
    u.ru = r[c];
    u.rd = r[(size_t)Nc + c];
    if (GS) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) u.zy[q] = nbr_ld((size_t)(NTOP + NSIDE + q) * Nc + c + (tsx_inward(q) ? offS : offN));
    }
    if (XL) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) u.zx[q] = nbr_ld((size_t)(NTOP + q) * Nc + c + (tsx_inward(q) ? offW : offE));
    }
    if (HAS1D) {
      u.t11 = a11[(size_t)k * ncol + ncl];
      u.t12 = a12[(size_t)k * ncol + ncl];
    }
    return u;
  };

  real A = albc, B = rsurf;
  auto step_up = [&](int k, const TsxUpRaw &u) {
    const size_t c = (size_t)k * ncol + col;
    const tsx_h4 tt = __builtin_bit_cast(tsx_h4, make_uint2(u.c0.x, u.c0.y));
    real tuu = (real)tt[0], rud = (real)tt[1], rdu = (real)tt[2], tdd = (real)tt[3];
    float gu8 = 0.0f, gd8 = 0.0f;  // coupling sums in fp8 units (x TSX_FP8_SCALE), fp32 accumulation
    if (GS) {
      float ca[4], cb[4];  // [c(y0->0) c(y0->1) c(y1->0) c(y1->1)], [y2, y3]
      tsx_fp8x4(u.c0.z, ca);
      tsx_fp8x4(u.c0.w, cb);
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) {
        const float zv = (tsx_inward(q) ? offS : offN) ? nbr_val(u.zy[q]) : 0.0f;  // select: the unused slot may hold NaN
        gu8 += (q < 2 ? ca[2 * q] : cb[2 * (q - 2)]) * zv;
        gd8 += (q < 2 ? ca[2 * q + 1] : cb[2 * (q - 2) + 1]) * zv;
      }
    }
    if (XL) {
      float ca[4], cb[4];
      tsx_fp8x4(u.c1.x, ca);
      tsx_fp8x4(u.c1.y, cb);
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) {
        const float zv = (tsx_inward(q) ? offW : offE) ? nbr_val(u.zx[q]) : 0.0f;
        gu8 += (q < 2 ? ca[2 * q] : cb[2 * (q - 2)]) * zv;
        gd8 += (q < 2 ? ca[2 * q + 1] : cb[2 * (q - 2) + 1]) * zv;
      }
    }
    real gu = (real)gu8 * (real)(1.0 / TSX_FP8_SCALE), gd = (real)gd8 * (real)(1.0 / TSX_FP8_SCALE);
    if (HAS1D) {
      const bool one = l1d[k] != 0;
      tuu = one ? u.t11 : tuu;
      tdd = one ? u.t11 : tdd;
      rud = one ? u.t12 : rud;
      rdu = one ? u.t12 : rdu;
      gu = one ? (real)0.0 : gu;
      gd = one ? (real)0.0 : gd;
    }
    const real ru = u.ru + gu, rd = u.rd + gd;
    const real G = (real)1.0 / ((real)1.0 - rdu * A);
    const real Gw = G * (rd + rdu * B);
    const real GT = G * tdd;
    if (LDST) tsx_pc_lds[k * 64 + threadIdx.x] = make_float4((float)Gw, (float)GT, (float)A, (float)B);
    else tmp[c] = make_float4((float)Gw, (float)GT, (float)A, (float)B);
    const real Bn = ru + tuu * (B + A * Gw);
    const real An = tuu * A * GT + rud;
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
    d.cu = P[(size_t)2 * Nc + c];
    d.cv = P[(size_t)3 * Nc + c];
    d.t = LDST ? tsx_pc_lds[k * 64 + threadIdx.x] : tmp[c];  // prefetched with the rest of the level: off the recurrence
#pragma unroll
    for (int q = 0; q < 8; ++q) d.rs[q] = r[(size_t)(NTOP + q) * Nc + c];
    if (FINAL) {
#pragma unroll
      for (int q = 0; q < D; ++q) d.pz[q] = z[(size_t)q * Nc + c + oc];
    }
    if (GS) {
#pragma unroll
      for (int m = 0; m < 2; ++m) d.cy[m] = P[(size_t)(4 + m) * Nc + c];
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) d.zy[q] = nbr_ld((size_t)(NTOP + NSIDE + q) * Nc + c + (tsx_inward(q) ? offS : offN));
    }
    if (XL) {
#pragma unroll
      for (int m = 0; m < 2; ++m) d.cx[m] = P[(size_t)(6 + m) * Nc + c];
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) d.zx[q] = nbr_ld((size_t)(NTOP + q) * Nc + c + (tsx_inward(q) ? offW : offE));
    }
    return d;
  };

  real V = rt[(size_t)ncol + col];  // V_0 = rd_0 (TOA identity row)
  if (MODE == 1) zt[(size_t)ncol + col] = (float)V;
  if (FINAL) wpair(zft + (size_t)ncol + ncp, (float)V, zt[(size_t)ncol + col + oc]);
  real U = A * V + B;               // A, B hold level 0
  auto step_dn = [&](int k, const TsxDnRaw &d) {
    const size_t c = (size_t)k * ncol + col;
    bool one = false;
    if (HAS1D) one = l1d[k] != 0;
    const float4 t = d.t;
    const real Vn = (real)t.x + (real)t.y * V;
    const real Un = (real)t.z * Vn + (real)t.w;
    if (MODE == 1) {
      z[c] = (float)U;
      z[(size_t)Nc + c] = (float)Vn;
    }
    if (FINAL) {
      wpair(zfin + cn0(k), (float)U, d.pz[0]);
      wpair(zfin + (size_t)Nc + cn0(k), (float)Vn, d.pz[1]);
    }
    float zy[NSIDE], zq[NSIDE];
    if (GS) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) zy[q] = (tsx_inward(q) ? offS : offN) ? nbr_val(d.zy[q]) : 0.0f;
    }
    if (XL) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) zq[q] = (tsx_inward(q) ? offW : offE) ? nbr_val(d.zx[q]) : 0.0f;
    }
    const tsx_h8 hcu = __builtin_bit_cast(tsx_h8, d.cu), hcv = __builtin_bit_cast(tsx_h8, d.cv);
    const unsigned wy[8] = {d.cy[0].x, d.cy[0].y, d.cy[0].z, d.cy[0].w, d.cy[1].x, d.cy[1].y, d.cy[1].z, d.cy[1].w};
    const unsigned wx[8] = {d.cx[0].x, d.cx[0].y, d.cx[0].z, d.cx[0].w, d.cx[1].x, d.cx[1].y, d.cx[1].z, d.cx[1].w};
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) {
      real acc = (real)hcu[dd] * Un + (real)hcv[dd] * V;
      float a8 = 0.0f;  // couplings: fp8 units, fp32 accumulation
      if (GS) {
        float cq[4];
        tsx_fp8x4(wy[dd], cq);
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) a8 += cq[q] * zy[q];
      }
      if (XL) {
        float cq[4];
        tsx_fp8x4(wx[dd], cq);
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) a8 += cq[q] * zq[q];
      }
      acc += (real)a8 * (real)(1.0 / TSX_FP8_SCALE);
      const float zo = (float)(d.rs[dd] + (one ? (real)0.0 : acc));
      if (MODE == 0) zb[(size_t)(NTOP + dd) * Nc + c] = tsx_to_bf16(zo);
      if (MODE == 1) z[(size_t)(NTOP + dd) * Nc + c] = zo;
      if (FINAL) wpair(zfin + (size_t)(NTOP + dd) * Nc + cn0(k), zo, d.pz[NTOP + dd]);
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
  if (MODE == 1) zt[col] = (float)U;  // U_Nz
  if (FINAL) wpair(zft + ncp, (float)U, zt[col + oc]);
#pragma unroll
  for (int d = NTOP; d < D; ++d) {
    const float v = rt[(size_t)d * ncol + col];
    if (MODE == 1) zt[(size_t)d * ncol + col] = v;
    if (FINAL) wpair(zft + (size_t)d * ncol + ncp, v, zt[(size_t)d * ncol + col + oc]);
  }
}

// ---- 8_16 (H = 4 up/down pairs) on the packed blocks: same mathematics as tsx_k_pc_column<8,4,...> (4x4 block
// recurrences), same software pipeline as tsx_k_pc_column_p16.  Temporaries per cell: 6 records of 16 B
// [Gw fp32 | GT rows 0..3 fp16 (2 records) | A_{k+1} rows 0..3 fp16 (2 records) | B_{k+1} fp32]: GT and A are products of
// transfer coefficients in [0, 1], Gw and B carry flux magnitudes.
struct TsxUpRawH {
  tsx_h8 row[8];
  uint4 cy[2], cx[2];
  float r[8];
  double t11, t12;
  float zy[4], zx[4];
};
struct TsxDnRawH {
  tsx_h8 row[8];
  uint4 cy[2], cx[2];
  uint4 t[6];
  float rs[8];
  float zy[4], zx[4];
};

template <int ROWS, bool GS, bool HAS1D, bool XL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void tsx_k_pc_column_p16h(
    TsxGeo g, const uint4 *__restrict__ P, const uint8_t *__restrict__ l1d, const double *__restrict__ a11,
    const double *__restrict__ a12, const double *__restrict__ albedo, const float *__restrict__ r, float *__restrict__ z,
    const float *__restrict__ zc, const float *__restrict__ zx, float4 *__restrict__ tmp_, const int *__restrict__ done) {
  constexpr int D = 16, NTOP = 8, NSIDE = 4, H = 4;
  uint4 *__restrict__ tmp = reinterpret_cast<uint4 *>(tmp_);
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
  const float *__restrict__ rt = r + (size_t)D * Nc;
  float *__restrict__ zt = z + (size_t)D * Nc;
  const double albh = albedo[col] / (double)H;  // assembled surface row: albedo/streams on every pair

  auto load_up = [&](int k) {
    TsxUpRawH u;
    const size_t c = (size_t)k * ncol + col;
#pragma unroll
    for (int t = 0; t < 8; ++t) u.row[t] = __builtin_bit_cast(tsx_h8, P[(size_t)t * Nc + c]);
#pragma unroll
    for (int t = 0; t < 8; ++t) u.r[t] = r[(size_t)t * Nc + c];
    if (GS) {
#pragma unroll
      for (int m = 0; m < 2; ++m) u.cy[m] = P[(size_t)(8 + m) * Nc + c];
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) u.zy[q] = zc[(size_t)(NTOP + NSIDE + q) * Nc + c + (tsx_inward(q) ? offS : offN)];
    }
    if (XL) {
#pragma unroll
      for (int m = 0; m < 2; ++m) u.cx[m] = P[(size_t)(10 + m) * Nc + c];
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
      float gu8 = 0.0f, gd8 = 0.0f;  // coupling sums in fp8 units (x TSX_FP8_SCALE)
      if (GS) {
        const unsigned wu = a < 2 ? (a == 0 ? u.cy[0].x : u.cy[0].z) : (a == 2 ? u.cy[1].x : u.cy[1].z);  // dst 2a
        const unsigned wd = a < 2 ? (a == 0 ? u.cy[0].y : u.cy[0].w) : (a == 2 ? u.cy[1].y : u.cy[1].w);  // dst 2a+1
        float cu4[4], cd4[4];
        tsx_fp8x4(wu, cu4);
        tsx_fp8x4(wd, cd4);
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) {
          const float zv = (tsx_inward(q) ? offS : offN) ? u.zy[q] : 0.0f;  // select: the unused slot may hold NaN
          gu8 += cu4[q] * zv;
          gd8 += cd4[q] * zv;
        }
      }
      if (XL) {
        const unsigned wu = a < 2 ? (a == 0 ? u.cx[0].x : u.cx[0].z) : (a == 2 ? u.cx[1].x : u.cx[1].z);
        const unsigned wd = a < 2 ? (a == 0 ? u.cx[0].y : u.cx[0].w) : (a == 2 ? u.cx[1].y : u.cx[1].w);
        float cu4[4], cd4[4];
        tsx_fp8x4(wu, cu4);
        tsx_fp8x4(wd, cd4);
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) {
          const float zv = (tsx_inward(q) ? offW : offE) ? u.zx[q] : 0.0f;
          gu8 += cu4[q] * zv;
          gd8 += cd4[q] * zv;
        }
      }
      ru[a] = (double)u.r[2 * a] + (one ? 0.0 : (double)gu8 * (1.0 / TSX_FP8_SCALE));
      rd[a] = (double)u.r[2 * a + 1] + (one ? 0.0 : (double)gd8 * (1.0 / TSX_FP8_SCALE));
    }
    double RA[H][H], G[H][H], GT[H][H], w[H], Gw[H], AGw[H], TA[H][H], An[H][H], Bn[H];
    SM::matmul(Rdu, A, RA);
    SM::inv_i_minus(RA, G);
    SM::matvec(Rdu, B, w);
#pragma unroll
    for (int a = 0; a < H; ++a) w[a] += rd[a];
    SM::matvec(G, w, Gw);
    SM::matmul(G, Tdd, GT);
    tmp[(size_t)0 * Nc + c] = __builtin_bit_cast(uint4, make_float4((float)Gw[0], (float)Gw[1], (float)Gw[2], (float)Gw[3]));
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2) {  // two matrix rows per record
      tmp[(size_t)(1 + a2) * Nc + c] =
          make_uint4(tsx_to_h2((float)GT[2 * a2][0], (float)GT[2 * a2][1]), tsx_to_h2((float)GT[2 * a2][2], (float)GT[2 * a2][3]),
                     tsx_to_h2((float)GT[2 * a2 + 1][0], (float)GT[2 * a2 + 1][1]), tsx_to_h2((float)GT[2 * a2 + 1][2], (float)GT[2 * a2 + 1][3]));
      tmp[(size_t)(3 + a2) * Nc + c] =
          make_uint4(tsx_to_h2((float)A[2 * a2][0], (float)A[2 * a2][1]), tsx_to_h2((float)A[2 * a2][2], (float)A[2 * a2][3]),
                     tsx_to_h2((float)A[2 * a2 + 1][0], (float)A[2 * a2 + 1][1]), tsx_to_h2((float)A[2 * a2 + 1][2], (float)A[2 * a2 + 1][3]));
    }
    tmp[(size_t)5 * Nc + c] = __builtin_bit_cast(uint4, make_float4((float)B[0], (float)B[1], (float)B[2], (float)B[3]));
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
    for (int dd = 0; dd < 8; ++dd) d.row[dd] = __builtin_bit_cast(tsx_h8, P[(size_t)(12 + dd) * Nc + c]);
#pragma unroll
    for (int q = 0; q < 6; ++q) d.t[q] = tmp[(size_t)q * Nc + c];
#pragma unroll
    for (int q = 0; q < 8; ++q) d.rs[q] = r[(size_t)(NTOP + q) * Nc + c];
    if (GS) {
#pragma unroll
      for (int m = 0; m < 2; ++m) d.cy[m] = P[(size_t)(20 + m) * Nc + c];
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) d.zy[q] = zc[(size_t)(NTOP + NSIDE + q) * Nc + c + (tsx_inward(q) ? offS : offN)];
    }
    if (XL) {
#pragma unroll
      for (int m = 0; m < 2; ++m) d.cx[m] = P[(size_t)(22 + m) * Nc + c];
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

  auto f4 = [](const uint4 &u, int i) {  // element i of a record of 4 floats
    const float4 v = __builtin_bit_cast(float4, u);
    return (double)(i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w);
  };
  auto h8 = [](const uint4 &u, int i) { return (double)__builtin_bit_cast(tsx_h8, u)[i]; };  // element i of 8 halves
  auto step_dn = [&](int k, const TsxDnRawH &d) {
    const size_t c = (size_t)k * ncol + col;
    bool one = false;
    if (HAS1D) one = l1d[k] != 0;
    double Vn[H], Un[H];
#pragma unroll
    for (int a = 0; a < H; ++a) {
      double v = f4(d.t[0], a);
#pragma unroll
      for (int b = 0; b < H; ++b) v += h8(d.t[1 + (a >> 1)], (a & 1) * 4 + b) * V[b];
      Vn[a] = v;
    }
#pragma unroll
    for (int a = 0; a < H; ++a) {  // U_{k+1} = A_{k+1} V_{k+1} + B_{k+1} (the surface closure is what the sweep started from)
      double v = f4(d.t[5], a);
#pragma unroll
      for (int b = 0; b < H; ++b) v += h8(d.t[3 + (a >> 1)], (a & 1) * 4 + b) * Vn[b];
      Un[a] = v;
    }
#pragma unroll
    for (int a = 0; a < H; ++a) {
      z[(size_t)(2 * a) * Nc + c] = (float)U[a];
      z[(size_t)(2 * a + 1) * Nc + c] = (float)Vn[a];
    }
    float zy[NSIDE], zq[NSIDE];
    if (GS) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) zy[q] = (tsx_inward(q) ? offS : offN) ? d.zy[q] : 0.0f;
    }
    if (XL) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) zq[q] = (tsx_inward(q) ? offW : offE) ? d.zx[q] : 0.0f;
    }
    const unsigned wy[8] = {d.cy[0].x, d.cy[0].y, d.cy[0].z, d.cy[0].w, d.cy[1].x, d.cy[1].y, d.cy[1].z, d.cy[1].w};
    const unsigned wx[8] = {d.cx[0].x, d.cx[0].y, d.cx[0].z, d.cx[0].w, d.cx[1].x, d.cx[1].y, d.cx[1].z, d.cx[1].w};
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) {
      double acc = 0.0;
#pragma unroll
      for (int a = 0; a < H; ++a) acc += (double)d.row[dd][2 * a] * Un[a] + (double)d.row[dd][2 * a + 1] * V[a];
      float a8 = 0.0f;
      if (GS) {
        float cq[4];
        tsx_fp8x4(wy[dd], cq);
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) a8 += cq[q] * zy[q];
      }
      if (XL) {
        float cq[4];
        tsx_fp8x4(wx[dd], cq);
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) a8 += cq[q] * zq[q];
      }
      acc += (double)a8 * (1.0 / TSX_FP8_SCALE);
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

// 8_16 in red-black order: tsx_k_pc_column_p16h with the index mapping of tsx_k_pc_column_rb (colour-split private
// arrays, natural-layout output only in the FINAL pass of a colour)
template <bool GS, bool HAS1D, bool FINAL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void tsx_k_pc_column_rbh(
    TsxGeo g, const uint4 *__restrict__ P, const uint8_t *__restrict__ l1d, const double *__restrict__ a11,
    const double *__restrict__ a12, const double *__restrict__ albedo, const float *__restrict__ r, float *__restrict__ z,
    const float *__restrict__ zc, float *__restrict__ zfin, float4 *__restrict__ tmp_, const int *__restrict__ done, int rbc) {
  constexpr int D = 16, NTOP = 8, NSIDE = 4, H = 4;
  constexpr bool XL = GS;
  const float *__restrict__ zx = zc;
  uint4 *__restrict__ tmp = reinterpret_cast<uint4 *>(tmp_);
  constexpr int PU = 2, PD = 1;
  using SM = TsxSm<H>;
  if (done && *done) return;
  const int h = g.xm >> 1;
  const int t_ = blockIdx.x * 64 + threadIdx.x;
  if (t_ >= g.ym * h) return;
  const long long Nc = g.Nc;
  const int Nz = g.Nz, ncol = g.ncol;
  const int jrow = t_ / h, qh = t_ - jrow * h;
  const int par = (jrow + rbc) & 1;
  const int icol = 2 * qh + par;
  const int ncl = jrow * g.xm + icol;          // natural column index
  const int col = jrow * g.xm + rbc * h + qh;  // colour-split column index
  const long long oc = (long long)(1 - 2 * rbc) * h;
  const int jn = jrow + 1 < g.ym ? jrow + 1 : (g.wrap_y ? 0 : -1), js = jrow > 0 ? jrow - 1 : (g.wrap_y ? g.ym - 1 : -1);
  const int qw = par ? qh : (qh > 0 ? qh - 1 : (g.wrap_x ? h - 1 : -1)), qe = par ? (qh + 1 < h ? qh + 1 : (g.wrap_x ? 0 : -1)) : qh;
  const long long offN = jn >= 0 ? (long long)(jn - jrow) * g.xm + oc : 0;
  const long long offS = js >= 0 ? (long long)(js - jrow) * g.xm + oc : 0;
  const long long offE = qe >= 0 ? oc + (qe - qh) : 0;
  const long long offW = qw >= 0 ? oc + (qw - qh) : 0;
  // FINAL (the very last pass): this lane's value and its row partner's (columns 2q, 2q+1: one of each colour; the
  // partner's final value sits at the same q in the other colour's half, offset oc) go out as one aligned float2 in the
  // Krylov layout -- contiguous stores instead of two stride-2 passes
  const int ncp = jrow * g.xm + 2 * qh;  // natural index of the pair's first column
  auto cn0 = [&](int k) { return (size_t)k * ncol + ncp; };
  auto wpair = [&](float *dst, float mine, float partner) {
    *reinterpret_cast<float2 *>(dst) = par ? make_float2(partner, mine) : make_float2(mine, partner);
  };
  const float *__restrict__ rt = r + (size_t)D * Nc;
  float *__restrict__ zt = z + (size_t)D * Nc;
  float *__restrict__ zft = FINAL ? zfin + (size_t)D * Nc : nullptr;
  const double albh = albedo[ncl] / (double)H;  // assembled surface row: albedo/streams on every pair

  auto load_up = [&](int k) {
    TsxUpRawH u;
    const size_t c = (size_t)k * ncol + col;
#pragma unroll
    for (int t = 0; t < 8; ++t) u.row[t] = __builtin_bit_cast(tsx_h8, P[(size_t)t * Nc + c]);
#pragma unroll
    for (int t = 0; t < 8; ++t) u.r[t] = r[(size_t)t * Nc + c];
    if (GS) {
#pragma unroll
      for (int m = 0; m < 2; ++m) u.cy[m] = P[(size_t)(8 + m) * Nc + c];
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) u.zy[q] = zc[(size_t)(NTOP + NSIDE + q) * Nc + c + (tsx_inward(q) ? offS : offN)];
    }
    if (XL) {
#pragma unroll
      for (int m = 0; m < 2; ++m) u.cx[m] = P[(size_t)(10 + m) * Nc + c];
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) u.zx[q] = zx[(size_t)(NTOP + q) * Nc + c + (tsx_inward(q) ? offW : offE)];
    }
    if (HAS1D) {
      u.t11 = a11[(size_t)k * ncol + ncl];
      u.t12 = a12[(size_t)k * ncol + ncl];
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
      float gu8 = 0.0f, gd8 = 0.0f;  // coupling sums in fp8 units (x TSX_FP8_SCALE)
      if (GS) {
        const unsigned wu = a < 2 ? (a == 0 ? u.cy[0].x : u.cy[0].z) : (a == 2 ? u.cy[1].x : u.cy[1].z);  // dst 2a
        const unsigned wd = a < 2 ? (a == 0 ? u.cy[0].y : u.cy[0].w) : (a == 2 ? u.cy[1].y : u.cy[1].w);  // dst 2a+1
        float cu4[4], cd4[4];
        tsx_fp8x4(wu, cu4);
        tsx_fp8x4(wd, cd4);
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) {
          const float zv = (tsx_inward(q) ? offS : offN) ? u.zy[q] : 0.0f;  // select: the unused slot may hold NaN
          gu8 += cu4[q] * zv;
          gd8 += cd4[q] * zv;
        }
      }
      if (XL) {
        const unsigned wu = a < 2 ? (a == 0 ? u.cx[0].x : u.cx[0].z) : (a == 2 ? u.cx[1].x : u.cx[1].z);
        const unsigned wd = a < 2 ? (a == 0 ? u.cx[0].y : u.cx[0].w) : (a == 2 ? u.cx[1].y : u.cx[1].w);
        float cu4[4], cd4[4];
        tsx_fp8x4(wu, cu4);
        tsx_fp8x4(wd, cd4);
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) {
          const float zv = (tsx_inward(q) ? offW : offE) ? u.zx[q] : 0.0f;
          gu8 += cu4[q] * zv;
          gd8 += cd4[q] * zv;
        }
      }
      ru[a] = (double)u.r[2 * a] + (one ? 0.0 : (double)gu8 * (1.0 / TSX_FP8_SCALE));
      rd[a] = (double)u.r[2 * a + 1] + (one ? 0.0 : (double)gd8 * (1.0 / TSX_FP8_SCALE));
    }
    double RA[H][H], G[H][H], GT[H][H], w[H], Gw[H], AGw[H], TA[H][H], An[H][H], Bn[H];
    SM::matmul(Rdu, A, RA);
    SM::inv_i_minus(RA, G);
    SM::matvec(Rdu, B, w);
#pragma unroll
    for (int a = 0; a < H; ++a) w[a] += rd[a];
    SM::matvec(G, w, Gw);
    SM::matmul(G, Tdd, GT);
    tmp[(size_t)0 * Nc + c] = __builtin_bit_cast(uint4, make_float4((float)Gw[0], (float)Gw[1], (float)Gw[2], (float)Gw[3]));
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2) {  // two matrix rows per record
      tmp[(size_t)(1 + a2) * Nc + c] =
          make_uint4(tsx_to_h2((float)GT[2 * a2][0], (float)GT[2 * a2][1]), tsx_to_h2((float)GT[2 * a2][2], (float)GT[2 * a2][3]),
                     tsx_to_h2((float)GT[2 * a2 + 1][0], (float)GT[2 * a2 + 1][1]), tsx_to_h2((float)GT[2 * a2 + 1][2], (float)GT[2 * a2 + 1][3]));
      tmp[(size_t)(3 + a2) * Nc + c] =
          make_uint4(tsx_to_h2((float)A[2 * a2][0], (float)A[2 * a2][1]), tsx_to_h2((float)A[2 * a2][2], (float)A[2 * a2][3]),
                     tsx_to_h2((float)A[2 * a2 + 1][0], (float)A[2 * a2 + 1][1]), tsx_to_h2((float)A[2 * a2 + 1][2], (float)A[2 * a2 + 1][3]));
    }
    tmp[(size_t)5 * Nc + c] = __builtin_bit_cast(uint4, make_float4((float)B[0], (float)B[1], (float)B[2], (float)B[3]));
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

  struct DnRaw : TsxDnRawH {
    float pz[16];  // FINAL: the row partner's final values, prefetched with the level
  };
  auto load_dn = [&](int k) {
    DnRaw d;
    const size_t c = (size_t)k * ncol + col;
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) d.row[dd] = __builtin_bit_cast(tsx_h8, P[(size_t)(12 + dd) * Nc + c]);
#pragma unroll
    for (int q = 0; q < 6; ++q) d.t[q] = tmp[(size_t)q * Nc + c];
#pragma unroll
    for (int q = 0; q < 8; ++q) d.rs[q] = r[(size_t)(NTOP + q) * Nc + c];
    if (FINAL) {
#pragma unroll
      for (int q = 0; q < D; ++q) d.pz[q] = z[(size_t)q * Nc + c + oc];
    }
    if (GS) {
#pragma unroll
      for (int m = 0; m < 2; ++m) d.cy[m] = P[(size_t)(20 + m) * Nc + c];
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) d.zy[q] = zc[(size_t)(NTOP + NSIDE + q) * Nc + c + (tsx_inward(q) ? offS : offN)];
    }
    if (XL) {
#pragma unroll
      for (int m = 0; m < 2; ++m) d.cx[m] = P[(size_t)(22 + m) * Nc + c];
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
    if (FINAL) wpair(zft + (size_t)(2 * a + 1) * ncol + ncp, (float)V[a], zt[(size_t)(2 * a + 1) * ncol + col + oc]);
  }
  SM::matvec(A, V, U);  // A, B hold level 0
#pragma unroll
  for (int a = 0; a < H; ++a) U[a] += B[a];

  auto f4 = [](const uint4 &u, int i) {  // element i of a record of 4 floats
    const float4 v = __builtin_bit_cast(float4, u);
    return (double)(i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w);
  };
  auto h8 = [](const uint4 &u, int i) { return (double)__builtin_bit_cast(tsx_h8, u)[i]; };  // element i of 8 halves
  auto step_dn = [&](int k, const DnRaw &d) {
    const size_t c = (size_t)k * ncol + col;
    bool one = false;
    if (HAS1D) one = l1d[k] != 0;
    double Vn[H], Un[H];
#pragma unroll
    for (int a = 0; a < H; ++a) {
      double v = f4(d.t[0], a);
#pragma unroll
      for (int b = 0; b < H; ++b) v += h8(d.t[1 + (a >> 1)], (a & 1) * 4 + b) * V[b];
      Vn[a] = v;
    }
#pragma unroll
    for (int a = 0; a < H; ++a) {  // U_{k+1} = A_{k+1} V_{k+1} + B_{k+1} (the surface closure is what the sweep started from)
      double v = f4(d.t[5], a);
#pragma unroll
      for (int b = 0; b < H; ++b) v += h8(d.t[3 + (a >> 1)], (a & 1) * 4 + b) * Vn[b];
      Un[a] = v;
    }
#pragma unroll
    for (int a = 0; a < H; ++a) {
      z[(size_t)(2 * a) * Nc + c] = (float)U[a];
      z[(size_t)(2 * a + 1) * Nc + c] = (float)Vn[a];
      if (FINAL) {
        wpair(zfin + (size_t)(2 * a) * Nc + cn0(k), (float)U[a], d.pz[2 * a]);
        wpair(zfin + (size_t)(2 * a + 1) * Nc + cn0(k), (float)Vn[a], d.pz[2 * a + 1]);
      }
    }
    float zy[NSIDE], zq[NSIDE];
    if (GS) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) zy[q] = (tsx_inward(q) ? offS : offN) ? d.zy[q] : 0.0f;
    }
    if (XL) {
#pragma unroll
      for (int q = 0; q < NSIDE; ++q) zq[q] = (tsx_inward(q) ? offW : offE) ? d.zx[q] : 0.0f;
    }
    const unsigned wy[8] = {d.cy[0].x, d.cy[0].y, d.cy[0].z, d.cy[0].w, d.cy[1].x, d.cy[1].y, d.cy[1].z, d.cy[1].w};
    const unsigned wx[8] = {d.cx[0].x, d.cx[0].y, d.cx[0].z, d.cx[0].w, d.cx[1].x, d.cx[1].y, d.cx[1].z, d.cx[1].w};
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) {
      double acc = 0.0;
#pragma unroll
      for (int a = 0; a < H; ++a) acc += (double)d.row[dd][2 * a] * Un[a] + (double)d.row[dd][2 * a + 1] * V[a];
      float a8 = 0.0f;
      if (GS) {
        float cq[4];
        tsx_fp8x4(wy[dd], cq);
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) a8 += cq[q] * zy[q];
      }
      if (XL) {
        float cq[4];
        tsx_fp8x4(wx[dd], cq);
#pragma unroll
        for (int q = 0; q < NSIDE; ++q) a8 += cq[q] * zq[q];
      }
      acc += (double)a8 * (1.0 / TSX_FP8_SCALE);
      const float zo = (float)(d.rs[dd] + (one ? 0.0 : acc));
      z[(size_t)(NTOP + dd) * Nc + c] = zo;
      if (FINAL) wpair(zfin + (size_t)(NTOP + dd) * Nc + cn0(k), zo, d.pz[NTOP + dd]);
    }
#pragma unroll
    for (int a = 0; a < H; ++a) {
      V[a] = Vn[a];
      U[a] = Un[a];
    }
  };

  // ---- downward sweep
  {
    DnRaw cd = load_dn(0);
    for (int k = 0; k < Nz; ++k) {
      const DnRaw nx = load_dn(k + 1 < Nz ? k + 1 : k);
      step_dn(k, cd);
      cd = nx;
    }
    (void)PD;
  }
#pragma unroll
  for (int a = 0; a < H; ++a) {
    zt[(size_t)(2 * a) * ncol + col] = (float)U[a];  // U_Nz
    if (FINAL) wpair(zft + (size_t)(2 * a) * ncol + ncp, (float)U[a], zt[(size_t)(2 * a) * ncol + col + oc]);
  }
#pragma unroll
  for (int d = NTOP; d < D; ++d) {
    const float v = rt[(size_t)d * ncol + col];
    zt[(size_t)d * ncol + col] = v;
    if (FINAL) wpair(zft + (size_t)d * ncol + ncp, v, zt[(size_t)d * ncol + col + oc]);
  }
}

__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_narrow(TsxGeo g, int split, const double *__restrict__ a, float *__restrict__ o) {
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < g.N; q += (long long)gridDim.x * TSX_BLOCK)
    o[split ? tsx_split_pos(q, g) : q] = (float)a[q];
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
