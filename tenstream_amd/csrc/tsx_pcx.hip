// tsx_pcx.hip -- the red-black column-block preconditioner in the reference's default arithmetic: exact blocks, fp64 iterates,
// fp64 recurrences -- as a segmented scan over the levels (round 6).
//
// The reference preconditions in `ireals` (real64 by default: PCILU / PCBJACOBI + ILU(0) on the assembled matrix,
// src/pprts.F90:4350-4371).  Until round 5 the only preconditioner of this library that rounded nothing was round 1's zebra-row
// sweep with one lane per column (tsx_k_pc_column_h1: 14 iterations, 36 M cells/s on the metric domain) -- what
// `fp32_directions = 0, pc_coeff_fp16 = 0` got, and what the retry after a failed solve ran (krylov_run_with_retry).  This is the
// same M^-1 as the mixed-precision scan kernels apply (tsx_kernels_pcs.hpp: exact two-stream solve per column, colours
// (i + j) & 1 alternately, the other colour's latest side streams on the right-hand side), evaluated the same way -- the column
// recurrences split into a matrix-only part (tsx_k_pcx_pack_col, once per coefficient set) and affine recurrences that NSEG
// segments of a column scan concurrently -- but on the operator's own blocks (fp32 as the LUT delivers them, lossless, or fp64)
// with every intermediate in fp64, directly in the Krylov vectors' dst-owned layout: a pass reads the other colour's side streams
// from z where the operator would read them from x (tsx_k_spmv_w's gather) and writes its colour's ten streams to z.  Nothing is
// packed, nothing is rounded; rank faces are treated like the reference's PCBJACOBI treats them (couplings across them dropped).
//     upward    B_k     = (ru_k + F_k rd_k) + E_k B_{k+1}       B_Nz = ru_Nz        U_k = A_k V_k + B_k
//     downward  V_{k+1} = (G_k rd_k + H_k B_{k+1}) + GT_k V_k   V_0  = rd_TOA
//     E = Tuu G, F = Tuu A_{k+1} G, G = 1 / (1 - Rdu A_{k+1}), H = G Rdu, GT = G Tdd, A_k = Rud + Tuu A_{k+1} GT, A_Nz = albedo
// The passes work on colour-split copies of v and z (within a row the xm / 2 columns of colour 0, then colour 1: tsx_split_pos) and on
// recurrence planes in the same order, so that a pass over one colour reads and writes contiguous memory -- in the Krylov vectors'
// natural order the lanes sit on every other column, every access touches twice the cache lines it uses, and the first version of
// this kernel was bound by exactly that (0.31 ms per pass, profiles/r06/exact_pc_bench.txt); two permuting copies per application.
// 3_10 only (8_16 keeps the zebra rows on this path).
#include "tsx_host.hpp"

namespace {
constexpr int PCX_CW = 16, PCX_NSEG = 16, PCX_REC = 7;

// matrix-only part, one lane per column: rec[q * Nc + c], q = E F G H GT A_{k+1} A_k (natural cell order)
template <typename CT, bool IDX>
__global__ __launch_bounds__(64) void tsx_k_pcx_pack_col(TsxGeo g, const CT *__restrict__ C, const int *__restrict__ cidx,
                                                         const uint8_t *__restrict__ l1d, const double *__restrict__ a11,
                                                         const double *__restrict__ a12, const double *__restrict__ albedo,
                                                         double *__restrict__ rec) {
  constexpr int D = 10;
  const int col = blockIdx.x * 64 + threadIdx.x;
  if (col >= g.ncol) return;
  const size_t Nc = (size_t)g.Nc;
  const size_t sp = (size_t)tsx_split_col(col % g.xm, col / g.xm, g.xm);  // the records live in colour-split order
  double A = albedo[col];
  for (int k = g.Nz - 1; k >= 0; --k) {
    const size_t c = (size_t)k * g.ncol + col;
    const size_t o = (size_t)k * g.ncol + sp;
    double tuu, rud, rdu, tdd;
    if (l1d[k]) {
      tuu = tdd = a11[c];
      rud = rdu = a12[c];
    } else if (IDX) {  // entry-major shared blocks Ce[id * D*D + dst * D + src]
      const CT *b = C + (size_t)cidx[c] * (D * D);
      tuu = (double)b[0 * D + 0];
      rud = (double)b[0 * D + 1];
      rdu = (double)b[1 * D + 0];
      tdd = (double)b[1 * D + 1];
    } else {
      tuu = (double)C[(size_t)(0 * D + 0) * Nc + c];
      rud = (double)C[(size_t)(0 * D + 1) * Nc + c];
      rdu = (double)C[(size_t)(1 * D + 0) * Nc + c];
      tdd = (double)C[(size_t)(1 * D + 1) * Nc + c];
    }
    const double G = 1.0 / (1.0 - rdu * A);
    const double GT = G * tdd;
    const double Ao = rud + tuu * A * GT;
    rec[0 * Nc + o] = tuu * G;
    rec[1 * Nc + o] = tuu * A * G;
    rec[2 * Nc + o] = G;
    rec[3 * Nc + o] = G * rdu;
    rec[4 * Nc + o] = GT;
    rec[5 * Nc + o] = A;
    rec[6 * Nc + o] = Ao;
    A = Ao;
  }
}

// o[split position] = a[natural position] (to_split) or the reverse, over the N unknowns (body planes and tail rows)
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_pcx_permute(TsxGeo g, int to_split, const double *__restrict__ a, double *__restrict__ o,
                                                               const int *__restrict__ done) {
  if (done && *done) return;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < g.N; q += (long long)gridDim.x * TSX_BLOCK) {
    const long long p = tsx_split_pos(q, g);
    if (to_split) o[p] = a[q];
    else o[q] = a[p];
  }
}

// one half-grid pass: the columns of colour rbc.  r, z, zo, rec: colour-split order; C (dense planes) and cidx: natural order.  gs: the other colour's side streams (from z) enter the right-hand side
template <typename CT, int LSEG, bool IDX, bool GS>
__global__ __launch_bounds__(PCX_CW *PCX_NSEG) void tsx_k_pcx_rb(TsxGeo g, const CT *__restrict__ C, const int *__restrict__ cidx,
                                                                 const double *__restrict__ rec, const uint8_t *__restrict__ l1d,
                                                                 const double *__restrict__ r, double *__restrict__ z,
                                                                 const double *__restrict__ zo, const int *__restrict__ done, int rbc) {
  constexpr bool gs = GS;
  // zo aliases z; it is only read at columns of the OTHER colour, which this launch never writes (a separate restrict pointer
  // lets the compiler issue those loads ahead of the stores to z)
  constexpr int D = 10, CW = PCX_CW, NSEG = PCX_NSEG;
  __shared__ double2 sS[NSEG][CW];
  if (done && *done) return;
  const int xm = g.xm, ym = g.ym, Nz = g.Nz, ncol = g.ncol;
  const size_t Nc = (size_t)g.Nc;
  const int h = xm >> 1;
  const int cl = threadIdx.x % CW, sg = threadIdx.x / CW;
  int t = blockIdx.x * CW + cl;
  const bool live = t < ym * h;
  if (!live) t = ym * h - 1;
  const int jrow = t / h, qh = t - jrow * h;
  const int par = (jrow + rbc) & 1;
  const int i = 2 * qh + par;
  const int coln = jrow * xm + i;            // natural column (dense coefficient planes, per-cell index)
  const int col = jrow * xm + rbc * h + qh;  // colour-split column (r, z, rec)
  // the neighbour a side source stream comes from (tsx_k_spmv_w's gather): inward x streams from the west, the others from the
  // east; inward y streams from the south, the others from the north -- all of the other colour, i.e. in the other half of a row
  // in split order (as tsx_pcs_rb_body addresses them).  0 = no neighbour (rank face, or gs off)
  const int oc = (1 - 2 * rbc) * h;
  const int jn = jrow + 1 < ym ? jrow + 1 : (g.wrap_y ? 0 : -1), js = jrow > 0 ? jrow - 1 : (g.wrap_y ? ym - 1 : -1);
  const int qw = par ? qh : (qh > 0 ? qh - 1 : (g.wrap_x ? h - 1 : -1)), qe = par ? (qh + 1 < h ? qh + 1 : (g.wrap_x ? 0 : -1)) : qh;
  long long offN = jn >= 0 ? (long long)(jn - jrow) * xm + oc : 0, offS = js >= 0 ? (long long)(js - jrow) * xm + oc : 0;
  long long offE = qe >= 0 ? (long long)oc + (qe - qh) : 0, offW = qw >= 0 ? (long long)oc + (qw - qh) : 0;
  if (g.pc_tile_x > 0) {
    if ((i + 1) % g.pc_tile_x == 0) offE = 0;
    if (i % g.pc_tile_x == 0) offW = 0;
  }
  if (g.pc_tile_y > 0) {
    if ((jrow + 1) % g.pc_tile_y == 0) offN = 0;
    if (jrow % g.pc_tile_y == 0) offS = 0;
  }
  if (!gs) offW = offE = offS = offN = 0;
  const long long soff[8] = {offE, offW, offE, offW, offN, offS, offN, offS};  // source stream 2 + q: tsx_inward(q) = q & 1
  const double *__restrict__ rt = r + (size_t)D * Nc;
  double *__restrict__ zt = z + (size_t)D * Nc;
  const int k0 = sg * LSEG;
  auto lev = [&](int l) { return k0 + l < Nz ? k0 + l : Nz - 1; };
  // src 2..9 of row dst (the couplings to the side streams that enter from the neighbouring columns) and src 0, 1 (the column's own
  // top streams), all loads unconditional -- a load under a branch ends a basic block and costs a full wait per level;
  // shared blocks are entry-major, a row is 40 contiguous bytes: five 8-byte loads instead of ten 4-byte ones
  auto row = [&](size_t c, int id, int dst, double(&top)[2], double(&side)[8]) {
    if constexpr (IDX && sizeof(CT) == 4) {
      const float2 *rw = reinterpret_cast<const float2 *>(C + (size_t)id * (D * D) + dst * D);
      const float2 t = rw[0];
      top[0] = (double)t.x;
      top[1] = (double)t.y;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float2 v = rw[1 + q];
        side[2 * q] = (double)v.x;
        side[2 * q + 1] = (double)v.y;
      }
    } else {
      top[0] = (double)C[(size_t)(dst * D + 0) * Nc + c];
      top[1] = (double)C[(size_t)(dst * D + 1) * Nc + c];
#pragma unroll
      for (int q = 0; q < 8; ++q) side[q] = (double)C[(size_t)(dst * D + 2 + q) * Nc + c];
    }
  };
  auto nbrs = [&](size_t c, double(&zn)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const double v = zo[(size_t)(2 + q) * Nc + c + soff[q]];  // unconditional load from a valid address, then select
      zn[q] = soff[q] ? v : 0.0;
    }
  };
  // ---- phase 1: local upward scan with zero inflow
  double beta[LSEG], rdv[LSEG], Ek[LSEG];
  {
    double Bl = 0.0, Pe = 1.0;
#pragma unroll
    for (int l = LSEG - 1; l >= 0; --l) {
      const bool act = k0 + l < Nz;
      const int k = lev(l);
      const size_t c = (size_t)k * ncol + col;
      double ru = r[0 * Nc + c], rd = r[1 * Nc + c];
      if constexpr (GS) {
        const bool one = l1d[k] != 0;
        const size_t cn = (size_t)k * ncol + coln;
        const int id = IDX ? cidx[cn] : 0;
        double zn[8], t0[2], t1[2], c0[8], c1[8];
        nbrs(c, zn);
        row(cn, id, 0, t0, c0);
        row(cn, id, 1, t1, c1);
        double gu = 0.0, gd = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          gu += c0[q] * zn[q];
          gd += c1[q] * zn[q];
        }
        ru += one ? 0.0 : gu;  // (a select, not a product: a 1-D layer's block may hold anything)
        rd += one ? 0.0 : gd;
      }
      const double E = rec[0 * Nc + c], F = rec[1 * Nc + c];
      Ek[l] = E;
      rdv[l] = rd;
      beta[l] = act ? ru + F * rd : 0.0;
      if (act) {
        Bl = beta[l] + E * Bl;
        Pe = E * Pe;
      }
    }
    sS[sg][cl] = make_double2(Bl, Pe);
  }
  __syncthreads();
  double Bin = rt[(size_t)0 * ncol + col];  // B_Nz = ru_Nz
  for (int s2 = NSEG - 1; s2 > sg; --s2) {
    const double2 m = sS[s2][cl];
    Bin = m.x + m.y * Bin;
  }
  // ---- phase 2: the true B of every level
  double Bk[LSEG];
  {
    double Bc = Bin;
#pragma unroll
    for (int l = LSEG - 1; l >= 0; --l) {
      if (k0 + l < Nz) Bc = beta[l] + Ek[l] * Bc;
      Bk[l] = Bc;
    }
  }
  __syncthreads();  // the upward summaries have been read: the buffer is free for the downward ones
  // ---- phase 3: local downward scan with zero inflow
  double gam[LSEG], GTk[LSEG];
  {
    double Vl = 0.0, Q = 1.0;
#pragma unroll
    for (int l = 0; l < LSEG; ++l) {
      const bool act = k0 + l < Nz;
      const size_t c = (size_t)lev(l) * ncol + col;
      const double Bn = l + 1 < LSEG ? Bk[l + 1 < LSEG ? l + 1 : l] : Bin;
      const double G = rec[2 * Nc + c], H = rec[3 * Nc + c], GT = rec[4 * Nc + c];
      GTk[l] = GT;
      gam[l] = act ? G * rdv[l] + H * Bn : 0.0;
      if (act) {
        Vl = gam[l] + GT * Vl;
        Q = GT * Q;
      }
    }
    sS[sg][cl] = make_double2(Vl, Q);
  }
  __syncthreads();
  double V = rt[(size_t)1 * ncol + col];  // V_0 = rd_TOA
  if (sg == 0 && live) {  // tail rows: the TOA identity row and the side dummies at level Nz
    zt[(size_t)1 * ncol + col] = V;
#pragma unroll
    for (int d = 2; d < D; ++d) zt[(size_t)d * ncol + col] = rt[(size_t)d * ncol + col];
  }
  for (int s2 = 0; s2 < sg; ++s2) {
    const double2 m = sS[s2][cl];
    V = m.x + m.y * V;
  }
  // ---- phase 4: true V, U; side streams; stores
  // (no early exit from the level loop: a branch between the levels fences their loads off from each other -- the loads of absent
  // levels go to the clamped level and only the stores are predicated)
#pragma unroll
  for (int l = 0; l < LSEG; ++l) {
    const bool act = k0 + l < Nz;
    const bool st = live && act;
    const int k = lev(l);
    const size_t c = (size_t)k * ncol + col;
    const double Bn = l + 1 < LSEG ? Bk[l + 1 < LSEG ? l + 1 : l] : Bin;
    const double Vn = gam[l] + GTk[l] * V;
    const double Un = rec[5 * Nc + c] * Vn + Bn;      // U_{k+1} = A_{k+1} V_{k+1} + B_{k+1}
    const double Uk = rec[6 * Nc + c] * V + Bk[l];    // U_k
    if (st) {
      z[0 * Nc + c] = Uk;
      z[1 * Nc + c] = Vn;
    }
    const bool one = l1d[k] != 0;
    const size_t cn = (size_t)k * ncol + coln;
    const int id = IDX ? cidx[cn] : 0;
    double zn[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if constexpr (GS) nbrs(c, zn);
#pragma unroll
    for (int d = 2; d < D; ++d) {
      double tp[2], sd[8];
      row(cn, id, d, tp, sd);
      double acc = tp[0] * Un + tp[1] * V;
      if constexpr (GS) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += sd[q] * zn[q];
      }
      acc = r[(size_t)d * Nc + c] + (one ? 0.0 : acc);
      if (st) z[(size_t)d * Nc + c] = acc;
    }
    if (k0 + l == Nz - 1 && live) zt[(size_t)0 * ncol + col] = Un;  // U_Nz = albedo V_Nz + ru_Nz: the surface row
    V = act ? Vn : V;
  }
}
}  // namespace

// exact scan passes available for this solver's grid?  (3_10, an even number of columns per row, an even number of rows where
// the rank wraps onto itself in y, at most 256 levels); TSX_PC_EXACT_SCAN=0 keeps the zebra rows (A/B)
bool tsx_pcx_eligible(const tsx_solver *s) {
  const TsxGeo &g = s->geo;
  const char *e = getenv("TSX_PC_EXACT_SCAN");
  if (e && atoi(e) == 0) return false;
  return g.ntop == 2 && g.xm % 2 == 0 && g.xm >= 2 && (!g.wrap_y || g.ym % 2 == 0) && g.Nz <= 16 * PCX_NSEG && g.Nc < (1ll << 31);
}

template <typename CT>
static int pcx_apply_t(tsx_solver *s, const CT *C, bool idx, const double *v, double *z, const int *done) {
  const TsxGeo &g = s->geo;
  if (!s->pcx_rec) HIPCHK(tsx_dev_malloc((void **)&s->pcx_rec, sizeof(double) * (size_t)PCX_REC * g.Nc));
  const int *cidx = idx ? (const int *)s->dd_cidx : (const int *)nullptr;
  if (!s->pcx_valid) {
    const int nbc = (g.ncol + 63) / 64;
    if (idx)
      hipLaunchKernelGGL((tsx_k_pcx_pack_col<CT, true>), dim3(nbc), dim3(64), 0, s->stream, g, C, cidx, s->l1d, s->a11, s->a12, s->albedo,
                         s->pcx_rec);
    else
      hipLaunchKernelGGL((tsx_k_pcx_pack_col<CT, false>), dim3(nbc), dim3(64), 0, s->stream, g, C, cidx, s->l1d, s->a11, s->a12, s->albedo,
                         s->pcx_rec);
    HIPCHK(hipGetLastError());
    s->pcx_valid = true;
  }
  const int nthr = g.ym * (g.xm / 2), nb = (nthr + PCX_CW - 1) / PCX_CW;
  const int P = s->pc_sweeps + 1;
  // colour-split copies of the right-hand side and the iterate (one allocation: [v | z])
  if (!s->pcx_vz) HIPCHK(tsx_dev_malloc((void **)&s->pcx_vz, sizeof(double) * 2 * (size_t)g.N));
  double *vs = s->pcx_vz, *zs = s->pcx_vz + (size_t)g.N;
  hipLaunchKernelGGL(tsx_k_pcx_permute, dim3(grid_for(g.N, 8192)), dim3(TSX_BLOCK), 0, s->stream, g, 1, v, vs, done);
  const double *vin = v;
  double *zout = z;
  v = vs;
  z = zs;
#define TSX_PCX_GO(L)                                                                                                                    \
  do {                                                                                                                                   \
    if (idx && pass > 0)                                                                                                                 \
      hipLaunchKernelGGL((tsx_k_pcx_rb<CT, L, true, true>), dim3(nb), dim3(PCX_CW *PCX_NSEG), 0, s->stream, g, C, cidx,                   \
                         (const double *)s->pcx_rec, s->l1d, v, z, (const double *)z, done, pass & 1);                                    \
    else if (idx)                                                                                                                        \
      hipLaunchKernelGGL((tsx_k_pcx_rb<CT, L, true, false>), dim3(nb), dim3(PCX_CW *PCX_NSEG), 0, s->stream, g, C, cidx,                  \
                         (const double *)s->pcx_rec, s->l1d, v, z, (const double *)z, done, pass & 1);                                    \
    else if (pass > 0)                                                                                                                   \
      hipLaunchKernelGGL((tsx_k_pcx_rb<CT, L, false, true>), dim3(nb), dim3(PCX_CW *PCX_NSEG), 0, s->stream, g, C, cidx,                  \
                         (const double *)s->pcx_rec, s->l1d, v, z, (const double *)z, done, pass & 1);                                    \
    else                                                                                                                                 \
      hipLaunchKernelGGL((tsx_k_pcx_rb<CT, L, false, false>), dim3(nb), dim3(PCX_CW *PCX_NSEG), 0, s->stream, g, C, cidx,                 \
                         (const double *)s->pcx_rec, s->l1d, v, z, (const double *)z, done, pass & 1);                                    \
  } while (0)
  for (int pass = 0; pass < P; ++pass) {
    if (g.Nz <= 4 * PCX_NSEG) TSX_PCX_GO(4);
    else if (g.Nz <= 8 * PCX_NSEG) TSX_PCX_GO(8);
    else TSX_PCX_GO(16);
  }
#undef TSX_PCX_GO
  (void)vin;
  hipLaunchKernelGGL(tsx_k_pcx_permute, dim3(grid_for(g.N, 8192)), dim3(TSX_BLOCK), 0, s->stream, g, 0, (const double *)zs, zout, done);
  HIPCHK(hipGetLastError());
  return TSX_OK;
}

// z = M^-1 v, both fp64 in the Krylov layout; pc_sweeps + 1 half-grid passes, colours alternately
int tsx_pcx_apply(tsx_solver *s, const double *v, double *z, const int *done) {
  if (s->dd_on && s->coef_bytes == 4) return pcx_apply_t<float>(s, (const float *)s->dd_coef_e, true, v, z, done);
  if (!s->coef_dense_valid) {
    tsx_set_error("exact scan preconditioner: neither shared blocks nor dense planes are valid (internal state error)");
    return TSX_ERR_STATE;
  }
  if (s->coef_bytes == 4) return pcx_apply_t<float>(s, (const float *)s->coef, false, v, z, done);
  return pcx_apply_t<double>(s, (const double *)s->coef, false, v, z, done);
}
