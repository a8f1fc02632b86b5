// tsx_kernels_pcs.hpp -- the red-black column-block preconditioner of 3_10 as a *segmented scan over the levels*.
//
// Same M^-1 as tsx_k_pc_column_rb (tsx_kernels_pc.hpp): per column the exact two-stream solve
//     U_k     = ru_k + Tuu U_{k+1} + Rud V_k ,   V_{k+1} = rd_k + Rdu U_{k+1} + Tdd V_k ,   V_0 = rd_TOA , U_Nz = ru_Nz + alb V_Nz
// (what the reference's ILU(0) in z-fastest ordering approximates, src/pprts.F90:4350-4371; the SOR sweep of
// src/pprts_explicit.F90:849-1015 is its scalar ancestor), then the side streams by substitution, colours (i + j) & 1
// alternately with the other colour's latest values on the right-hand side.
//
// What changed is the evaluation order.  One lane per column walking 2 x Nz dependent levels leaves 512 waves on 1024 SIMDs
// and pays a memory latency per level.  But the recurrences split into a part that depends on the matrix only and a part
// that is *affine* in the right-hand side:
//     with  A_Nz = alb,  G_k = 1 / (1 - Rdu_k A_{k+1}),  A_k = Rud_k + Tuu_k A_{k+1} G_k Tdd_k           (Moebius, matrix only)
//     E_k = Tuu_k G_k,  F_k = Tuu_k A_{k+1} G_k,  H_k = G_k Rdu_k,  GT_k = G_k Tdd_k                        (matrix only)
//     upward:    B_k     = (ru_k + F_k rd_k) + E_k B_{k+1} ,          B_Nz = ru_Nz          (U_k = A_k V_k + B_k)
//     downward:  V_{k+1} = (G_k rd_k + H_k B_{k+1}) + GT_k V_k ,      V_0  = rd_TOA
// The matrix-only part is evaluated once per coefficient set (tsx_k_pcs_pack_col) and stored with the packed blocks; the
// affine recurrences compose associatively, (e1, b1) o (e2, b2) = (e1 e2, b1 + e1 b2), so a column is cut into NSEG
// segments of LSEG levels that run concurrently: each thread scans its segment with zero inflow, the NSEG segment
// summaries (8 bytes each) are combined through LDS, and the thread corrects its levels with the true inflow.  A
// workgroup is CW columns x NSEG segments: every global load of a pass belongs to an independent (column, level) and can
// be in flight at once, there are NSEG x as many waves, and small domains (config 2: 8192 columns per pass) fill the chip
// with CW = 16 or 32.  Lanes still run along x: all accesses stay coalesced.
//
// Packed layout "S16" (8 records of 16 B per cell, colour-split order tsx_split_col), P[grp * Nc + cell]:
//   grp 0: E F G-1 H GT A_{k+1} A_k 0 (fp16)
//   grp 1: c(y_q->0) c(y_q->1), q = 0..3 | c(x_q->0) c(x_q->1), q = 0..3  (fp8 e4m3 x 64)     (y_q = src dof 6+q, x_q = 2+q)
//   grp 2: c(0 -> side d), d = 2..9 (fp16)        grp 3: c(1 -> side d) (fp16)
//   grp 4, 5: c(y_q -> side 2+dd), byte 4 dd + q (fp8)       grp 6, 7: c(x_q -> side 2+dd) (fp8)
// 1-D layers (src/pprts_shell.F90:417-427): grp 0 from a11 / a12, groups 1..7 zero -- the kernels need no 1-D branch.
// Iterates: the 8 side streams of a cell are stored as four records by the neighbour that consumes them,
//   rec 0 = dofs (2, 4) -> read by the west neighbour,  1 = (3, 5) -> east,  2 = (6, 8) -> south,  3 = (7, 9) -> north,
// bf16 pairs (4 B) in zb for the intermediate passes, float2 in z for the last two: a neighbour value costs one load per
// direction instead of one per stream.
#pragma once
#include "tsx_pack.hpp"

// ---- matrix-only part: one lane per column, once per coefficient set
template <typename CT>
__global__ __launch_bounds__(64) void tsx_k_pcs_pack_col(TsxGeo g, const CT *__restrict__ C, const uint8_t *__restrict__ l1d,
                                                         const double *__restrict__ a11, const double *__restrict__ a12,
                                                         const double *__restrict__ albedo, uint4 *__restrict__ P) {
  constexpr int D = 10;
  const int col = blockIdx.x * 64 + threadIdx.x;
  if (col >= g.ncol) return;
  const long long Nc = g.Nc;
  const long long sp = tsx_split_col(col % g.xm, col / g.xm, g.xm);
  double A = albedo[col];
  for (int k = g.Nz - 1; k >= 0; --k) {
    const size_t c = (size_t)k * g.ncol + col;
    double tuu, rud, rdu, tdd;
    if (l1d[k]) {
      tuu = tdd = a11[c];
      rud = rdu = a12[c];
    } else {
      tuu = (double)C[(size_t)(0 * D + 0) * Nc + c];
      rud = (double)C[(size_t)(0 * D + 1) * Nc + c];
      rdu = (double)C[(size_t)(1 * D + 0) * Nc + c];
      tdd = (double)C[(size_t)(1 * D + 1) * Nc + c];
    }
    const double G = 1.0 / (1.0 - rdu * A);
    const double GT = G * tdd;
    const double Ao = rud + tuu * A * GT;
    uint4 v;
    v.x = tsx_to_h2((float)(tuu * G), (float)(tuu * A * G));
    v.y = tsx_to_h2((float)(G - 1.0), (float)(G * rdu));  // G in [1, 2): its excess over 1 keeps 4x the resolution
    v.z = tsx_to_h2((float)GT, (float)A);
    v.w = tsx_to_h2((float)Ao, 0.0f);
    P[(size_t)k * g.ncol + sp] = v;
    A = Ao;
  }
}

// ---- groups 1..7: the couplings, regrouped in consumption order
template <typename CT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_pcs_pack(TsxGeo g, const CT *__restrict__ C, const uint8_t *__restrict__ l1d,
                                                            uint4 *__restrict__ P) {
  constexpr int D = 10;
  const long long Nc = g.Nc, n = Nc * 7;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * TSX_BLOCK) {
    const int grp = 1 + (int)(q / Nc);
    const long long c = q - (long long)(grp - 1) * Nc;
    const int i = (int)(c % g.xm);
    const long long t = c / g.xm;
    const int j = (int)(t % g.ym), k = (int)(t / g.ym);
    auto cf = [&](int dst, int src) { return (float)C[(size_t)(dst * D + src) * Nc + c]; };
    uint4 v = make_uint4(0, 0, 0, 0);
    if (!l1d[k]) {
      if (grp == 1) {
        v.x = tsx_to_fp8x4(cf(0, 6), cf(1, 6), cf(0, 7), cf(1, 7));
        v.y = tsx_to_fp8x4(cf(0, 8), cf(1, 8), cf(0, 9), cf(1, 9));
        v.z = tsx_to_fp8x4(cf(0, 2), cf(1, 2), cf(0, 3), cf(1, 3));
        v.w = tsx_to_fp8x4(cf(0, 4), cf(1, 4), cf(0, 5), cf(1, 5));
      } else if (grp == 2 || grp == 3) {
        const int s = grp - 2;
        v.x = tsx_to_h2(cf(2, s), cf(3, s));
        v.y = tsx_to_h2(cf(4, s), cf(5, s));
        v.z = tsx_to_h2(cf(6, s), cf(7, s));
        v.w = tsx_to_h2(cf(8, s), cf(9, s));
      } else {
        const int s0 = grp < 6 ? 6 : 2, d0 = 2 + 4 * ((grp - 4) & 1);
        v.x = tsx_to_fp8x4(cf(d0 + 0, s0), cf(d0 + 0, s0 + 1), cf(d0 + 0, s0 + 2), cf(d0 + 0, s0 + 3));
        v.y = tsx_to_fp8x4(cf(d0 + 1, s0), cf(d0 + 1, s0 + 1), cf(d0 + 1, s0 + 2), cf(d0 + 1, s0 + 3));
        v.z = tsx_to_fp8x4(cf(d0 + 2, s0), cf(d0 + 2, s0 + 1), cf(d0 + 2, s0 + 2), cf(d0 + 2, s0 + 3));
        v.w = tsx_to_fp8x4(cf(d0 + 3, s0), cf(d0 + 3, s0 + 1), cf(d0 + 3, s0 + 2), cf(d0 + 3, s0 + 3));
      }
    }
    P[(size_t)grp * Nc + (size_t)k * g.ncol + tsx_split_col(i, j, g.xm)] = v;
  }
}

// ---- the same groups 1..7 per *distinct* block (tsx_dedup.hip): PE[(grp - 1) * nent + id]; an entry whose representative
// cell lies in a 1-D layer stands for all 1-D cells: zero records
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_pcs_pack_ent(int ncol, long long nent, const float *__restrict__ Cd,
                                                                const int *__restrict__ ent_cell, const uint8_t *__restrict__ l1d,
                                                                uint4 *__restrict__ PE) {
  constexpr int D = 10;
  const long long n = nent * 7;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * TSX_BLOCK) {
    const int grp = 1 + (int)(q / nent);
    const long long id = q - (long long)(grp - 1) * nent;
    const int k = ent_cell[id] / ncol;
    auto cf = [&](int dst, int src) { return Cd[(size_t)(dst * D + src) * nent + id]; };
    uint4 v = make_uint4(0, 0, 0, 0);
    if (!l1d[k]) {
      if (grp == 1) {
        v.x = tsx_to_fp8x4(cf(0, 6), cf(1, 6), cf(0, 7), cf(1, 7));
        v.y = tsx_to_fp8x4(cf(0, 8), cf(1, 8), cf(0, 9), cf(1, 9));
        v.z = tsx_to_fp8x4(cf(0, 2), cf(1, 2), cf(0, 3), cf(1, 3));
        v.w = tsx_to_fp8x4(cf(0, 4), cf(1, 4), cf(0, 5), cf(1, 5));
      } else if (grp == 2 || grp == 3) {
        const int s = grp - 2;
        v.x = tsx_to_h2(cf(2, s), cf(3, s));
        v.y = tsx_to_h2(cf(4, s), cf(5, s));
        v.z = tsx_to_h2(cf(6, s), cf(7, s));
        v.w = tsx_to_h2(cf(8, s), cf(9, s));
      } else {
        const int s0 = grp < 6 ? 6 : 2, d0 = 2 + 4 * ((grp - 4) & 1);
        v.x = tsx_to_fp8x4(cf(d0 + 0, s0), cf(d0 + 0, s0 + 1), cf(d0 + 0, s0 + 2), cf(d0 + 0, s0 + 3));
        v.y = tsx_to_fp8x4(cf(d0 + 1, s0), cf(d0 + 1, s0 + 1), cf(d0 + 1, s0 + 2), cf(d0 + 1, s0 + 3));
        v.z = tsx_to_fp8x4(cf(d0 + 2, s0), cf(d0 + 2, s0 + 1), cf(d0 + 2, s0 + 2), cf(d0 + 2, s0 + 3));
        v.w = tsx_to_fp8x4(cf(d0 + 3, s0), cf(d0 + 3, s0 + 1), cf(d0 + 3, s0 + 2), cf(d0 + 3, s0 + 3));
      }
    }
    PE[q] = v;
  }
}

__device__ __forceinline__ unsigned tsx_bf16x2(float lo, float hi) {
  return (unsigned)tsx_to_bf16(lo) | ((unsigned)tsx_to_bf16(hi) << 16);
}

// ---- one half-grid pass.  rbc = colour of this pass.  GS: the other colour's values enter the right-hand side.
// MODE 0: intermediate pass -- only the side streams are stored, as bf16 records in zb; neighbours from zb.
// MODE 1: the last pass of the first colour -- all ten streams in fp32 to z (colour-split; side streams as float2 records);
//         neighbours from zb.
// MODE 2: the very last pass -- neighbours and the row partner's final values from z; the result of both colours goes out
//         as aligned pairs in the Krylov layout zfin.
// nonbr: run a GS kernel without neighbours (first pass of a short sequence).
// IDX: groups 1..7 are stored per distinct block: PE[(grp - 1) * nent + cidx[cell]] (cidx in colour-split order); group 0
// (the column recurrences) stays per cell.
template <int LSEG, int NSEG, int CW, bool GS, int MODE, bool IDX = false>
__global__ __launch_bounds__(CW *NSEG) void tsx_k_pcs_rb(TsxGeo g, const uint4 *__restrict__ P, const float *__restrict__ r,
                                                         float *__restrict__ z, unsigned *__restrict__ zb,
                                                         float *__restrict__ zfin, const int *__restrict__ done, int rbc,
                                                         int nonbr, const int *__restrict__ cidx, long long nent,
                                                         const uint4 *__restrict__ PE) {
  constexpr int D = 10, NTOP = 2;
  constexpr bool FINAL = MODE == 2;
  __shared__ float2 sB[NSEG][CW], sV[NSEG][CW];
  if (done && *done) return;
  const int h = g.xm >> 1;
  const int cl = threadIdx.x % CW, sg = threadIdx.x / CW;
  const int nthr = g.ym * h;
  int t_ = blockIdx.x * CW + cl;
  const bool live = t_ < nthr;  // dead lanes shadow the last column (loads stay valid, nothing is stored)
  if (!live) t_ = nthr - 1;
  const long long Nc = g.Nc;
  const int Nz = g.Nz, ncol = g.ncol;
  const int jrow = t_ / h, qh = t_ - jrow * h;
  const int par = (jrow + rbc) & 1;
  const int icol = 2 * qh + par;
  const int col = jrow * g.xm + rbc * h + qh;  // colour-split column index (P, r, z, zb)
  // neighbours (other colour) in split space; 0 = no neighbour (rank face / tile edge)
  const long long oc = (long long)(1 - 2 * rbc) * h;
  const int jn = jrow + 1 < g.ym ? jrow + 1 : (g.wrap_y ? 0 : -1), js = jrow > 0 ? jrow - 1 : (g.wrap_y ? g.ym - 1 : -1);
  const int qw = par ? qh : (qh > 0 ? qh - 1 : (g.wrap_x ? h - 1 : -1)), qe = par ? (qh + 1 < h ? qh + 1 : (g.wrap_x ? 0 : -1)) : qh;
  long long offN = jn >= 0 ? (long long)(jn - jrow) * g.xm + oc : 0;
  long long offS = js >= 0 ? (long long)(js - jrow) * g.xm + oc : 0;
  long long offE = qe >= 0 ? oc + (qe - qh) : 0;
  long long offW = qw >= 0 ? oc + (qw - qh) : 0;
  if (g.pc_tile_x > 0) {  // analysis knob: behave like a rank of pc_tile_x x pc_tile_y columns
    if ((icol + 1) % g.pc_tile_x == 0) offE = 0;
    if (icol % g.pc_tile_x == 0) offW = 0;
  }
  if (g.pc_tile_y > 0) {
    if ((jrow + 1) % g.pc_tile_y == 0) offN = 0;
    if (jrow % g.pc_tile_y == 0) offS = 0;
  }
  if (nonbr) offN = offS = offE = offW = 0;
  const int ncp = jrow * g.xm + 2 * qh;  // FINAL: natural index of the pair's first column
  auto wpair = [&](float *dst, float mine, float partner) {
    if (live) *reinterpret_cast<float2 *>(dst) = par ? make_float2(partner, mine) : make_float2(mine, partner);
  };
  const float *__restrict__ rt = r + (size_t)D * Nc;
  float *__restrict__ zt = z + (size_t)D * Nc;
  float2 *__restrict__ zr = reinterpret_cast<float2 *>(z + (size_t)2 * Nc);  // side-stream records of the fp32 iterate
  const float rsurf = rt[col], V0 = rt[(size_t)ncol + col];

  const int k0 = sg * LSEG;
  const int nl = Nz - k0 < LSEG ? (Nz - k0 > 0 ? Nz - k0 : 0) : LSEG;  // levels of this segment that exist
  auto cell = [&](int l) { return (size_t)(k0 + l < Nz ? k0 + l : Nz - 1) * ncol + col; };

  // neighbour records of one level: [E (dofs 2,4), W (3,5), N (6,8), S (7,9)]
  auto nbr_load = [&](size_t c, uint2 (&o)[4]) {
    const long long off[4] = {offE, offW, offN, offS};
    const int rec[4] = {0, 1, 2, 3};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const size_t idx = (size_t)rec[m] * Nc + c + off[m];
      if (MODE == 2) o[m] = *reinterpret_cast<const uint2 *>(zr + idx);
      else o[m] = make_uint2(zb[idx], 0u);
    }
  };
  // -> values by stream: zx[q] = stream 2+q entering through an x face, zy[q] = stream 6+q through a y face
  auto nbr_vals = [&](const uint2 (&n)[4], float (&zx)[4], float (&zy)[4]) {
    float lo[4], hi[4];
    const long long off[4] = {offE, offW, offN, offS};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float a = MODE == 2 ? __uint_as_float(n[m].x) : __uint_as_float(n[m].x << 16);
      const float b = MODE == 2 ? __uint_as_float(n[m].y) : __uint_as_float(n[m].x & 0xffff0000u);
      lo[m] = off[m] ? a : 0.0f;  // select: the unused slot may hold NaN
      hi[m] = off[m] ? b : 0.0f;
    }
    zx[0] = lo[0]; zx[2] = hi[0]; zx[1] = lo[1]; zx[3] = hi[1];
    zy[0] = lo[2]; zy[2] = hi[2]; zy[1] = lo[3]; zy[3] = hi[3];
  };

  // ---- phase 1: all loads of the upward scan (independent of each other), then the local scan with zero inflow
  uint4 r0[LSEG], r1[LSEG];
  float ru[LSEG], rd[LSEG];
  uint2 nb[LSEG][4];
  int eid[LSEG];
  auto rec = [&](int grp, size_t c, int id) { return IDX ? PE[(size_t)(grp - 1) * nent + id] : P[(size_t)grp * Nc + c]; };
#pragma unroll
  for (int l = 0; l < LSEG; ++l) {
    const size_t c = cell(l);
    eid[l] = IDX ? cidx[c] : 0;
    r0[l] = P[c];
    ru[l] = r[c];
    rd[l] = r[(size_t)Nc + c];
    if (GS) {
      r1[l] = rec(1, c, eid[l]);
      nbr_load(c, nb[l]);
    }
  }
  float Bloc[LSEG], Pcum[LSEG], rdg[LSEG];
  {
    float Bl = 0.0f, Pc = 1.0f;
#pragma unroll
    for (int l = LSEG - 1; l >= 0; --l) {
      const tsx_h8 m = __builtin_bit_cast(tsx_h8, r0[l]);
      float gu = 0.0f, gd = 0.0f;
      if (GS) {
        float zx[4], zy[4], cy0[4], cy1[4], cx0[4], cx1[4];
        nbr_vals(nb[l], zx, zy);
        tsx_fp8x4(r1[l].x, cy0);
        tsx_fp8x4(r1[l].y, cy1);
        tsx_fp8x4(r1[l].z, cx0);
        tsx_fp8x4(r1[l].w, cx1);
        float gu8 = cy0[0] * zy[0] + cy0[2] * zy[1] + cy1[0] * zy[2] + cy1[2] * zy[3];
        float gd8 = cy0[1] * zy[0] + cy0[3] * zy[1] + cy1[1] * zy[2] + cy1[3] * zy[3];
        gu8 += cx0[0] * zx[0] + cx0[2] * zx[1] + cx1[0] * zx[2] + cx1[2] * zx[3];
        gd8 += cx0[1] * zx[0] + cx0[3] * zx[1] + cx1[1] * zx[2] + cx1[3] * zx[3];
        gu = gu8 * (1.0f / TSX_FP8_SCALE);
        gd = gd8 * (1.0f / TSX_FP8_SCALE);
      }
      const bool act = l < nl;
      const float E = act ? (float)m[0] : 1.0f;
      const float rdl = rd[l] + gd;
      const float beta = act ? (ru[l] + gu) + (float)m[1] * rdl : 0.0f;
      Bl = beta + E * Bl;
      Pc *= E;
      Bloc[l] = Bl;
      Pcum[l] = Pc;
      rdg[l] = rdl;
    }
    sB[sg][cl] = make_float2(Bl, Pc);
  }
  __syncthreads();
  float Bin = rsurf;  // B at the level below this segment
  for (int s2 = NSEG - 1; s2 > sg; --s2) {
    const float2 v = sB[s2][cl];
    Bin = v.x + v.y * Bin;
  }
  // ---- phase 2: true B; local downward scan with zero inflow
  float Bk[LSEG], Vloc[LSEG], Qcum[LSEG];
#pragma unroll
  for (int l = 0; l < LSEG; ++l) Bk[l] = Bloc[l] + Pcum[l] * Bin;
  {
    float Vl = 0.0f, Qc = 1.0f;
#pragma unroll
    for (int l = 0; l < LSEG; ++l) {
      const tsx_h8 m = __builtin_bit_cast(tsx_h8, r0[l]);
      const bool act = l < nl;
      const float Bn = l + 1 < LSEG ? Bk[l + 1] : Bin;
      const float GT = act ? (float)m[4] : 1.0f;
      const float gam = act ? (rdg[l] + (float)m[2] * rdg[l]) + (float)m[3] * Bn : 0.0f;
      Vl = gam + GT * Vl;
      Qc *= GT;
      Vloc[l] = Vl;
      Qcum[l] = Qc;
    }
    sV[sg][cl] = make_float2(Vl, Qc);
  }
  __syncthreads();
  float Vin = V0;  // V at the top level of this segment
  for (int s2 = 0; s2 < sg; ++s2) {
    const float2 v = sV[s2][cl];
    Vin = v.x + v.y * Vin;
  }
  // ---- phase 3: true V, U; side streams; stores
  if (sg == 0) {  // tail rows: TOA Edn (identity row) and the side dummies at level Nz
    if (MODE == 1 && live) zt[(size_t)ncol + col] = V0;
    if (FINAL) wpair(zfin + (size_t)D * Nc + (size_t)ncol + ncp, V0, zt[(size_t)ncol + col + oc]);
#pragma unroll
    for (int d = NTOP; d < D; ++d) {
      const float v = rt[(size_t)d * ncol + col];
      if (MODE == 1 && live) zt[(size_t)d * ncol + col] = v;
      if (FINAL) wpair(zfin + (size_t)D * Nc + (size_t)d * ncol + ncp, v, zt[(size_t)d * ncol + col + oc]);
    }
  }
  // (loads are unconditional -- a load under a branch costs a full wait -- and only the stores are predicated)
  float V = Vin;
#pragma unroll
  for (int l = 0; l < LSEG; ++l) {
    const bool st = live && l < nl;
    auto wpair2 = [&](float *dst, float mine, float partner) {
      if (st) *reinterpret_cast<float2 *>(dst) = par ? make_float2(partner, mine) : make_float2(mine, partner);
    };
    const size_t c = cell(l);
    const size_t cn = (size_t)(k0 + l < Nz ? k0 + l : Nz - 1) * ncol + ncp;
    const tsx_h8 m = __builtin_bit_cast(tsx_h8, r0[l]);
    const uint4 wcu = rec(2, c, eid[l]), wcv = rec(3, c, eid[l]);
    uint4 wy[2], wx[2];
    if (GS) {
      wy[0] = rec(4, c, eid[l]);
      wy[1] = rec(5, c, eid[l]);
      wx[0] = rec(6, c, eid[l]);
      wx[1] = rec(7, c, eid[l]);
    }
    float rs[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) rs[q] = r[(size_t)(NTOP + q) * Nc + c];
    float pt[2];
    float2 ps[4];
    if (FINAL) {
      pt[0] = z[c + oc];
      pt[1] = z[(size_t)Nc + c + oc];
#pragma unroll
      for (int m2 = 0; m2 < 4; ++m2) ps[m2] = zr[(size_t)m2 * Nc + c + oc];
    }
    const float Bn = l + 1 < LSEG ? Bk[l + 1] : Bin;
    const float Vn = Vloc[l] + Qcum[l] * Vin;
    const float Un = (float)m[5] * Vn + Bn;
    const float U = (float)m[6] * V + Bk[l];
    if (MODE == 1 && st) {
      z[c] = U;
      z[(size_t)Nc + c] = Vn;
    }
    if (FINAL) {
      wpair2(zfin + cn, U, pt[0]);
      wpair2(zfin + (size_t)Nc + cn, Vn, pt[1]);
    }
    float zx[4], zy[4];
    if (GS) nbr_vals(nb[l], zx, zy);
    const tsx_h8 hcu = __builtin_bit_cast(tsx_h8, wcu), hcv = __builtin_bit_cast(tsx_h8, wcv);
    const unsigned uy[8] = {wy[0].x, wy[0].y, wy[0].z, wy[0].w, wy[1].x, wy[1].y, wy[1].z, wy[1].w};
    const unsigned ux[8] = {wx[0].x, wx[0].y, wx[0].z, wx[0].w, wx[1].x, wx[1].y, wx[1].z, wx[1].w};
    float zo[8];
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) {
      float acc = (float)hcu[dd] * Un + (float)hcv[dd] * V;
      if (GS) {
        float cq[4], cp[4];
        tsx_fp8x4(uy[dd], cq);
        tsx_fp8x4(ux[dd], cp);
        float a8 = cq[0] * zy[0] + cq[1] * zy[1] + cq[2] * zy[2] + cq[3] * zy[3];
        a8 += cp[0] * zx[0] + cp[1] * zx[1] + cp[2] * zx[2] + cp[3] * zx[3];
        acc += a8 * (1.0f / TSX_FP8_SCALE);
      }
      zo[dd] = rs[dd] + acc;
    }
    // records: (2,4) (3,5) (6,8) (7,9)  = zo[0,2] zo[1,3] zo[4,6] zo[5,7]
    if (MODE == 0 && st) {
      zb[(size_t)0 * Nc + c] = tsx_bf16x2(zo[0], zo[2]);
      zb[(size_t)1 * Nc + c] = tsx_bf16x2(zo[1], zo[3]);
      zb[(size_t)2 * Nc + c] = tsx_bf16x2(zo[4], zo[6]);
      zb[(size_t)3 * Nc + c] = tsx_bf16x2(zo[5], zo[7]);
    }
    if (MODE == 1 && st) {
      zr[(size_t)0 * Nc + c] = make_float2(zo[0], zo[2]);
      zr[(size_t)1 * Nc + c] = make_float2(zo[1], zo[3]);
      zr[(size_t)2 * Nc + c] = make_float2(zo[4], zo[6]);
      zr[(size_t)3 * Nc + c] = make_float2(zo[5], zo[7]);
    }
    if (FINAL) {
      wpair2(zfin + (size_t)2 * Nc + cn, zo[0], ps[0].x);
      wpair2(zfin + (size_t)4 * Nc + cn, zo[2], ps[0].y);
      wpair2(zfin + (size_t)3 * Nc + cn, zo[1], ps[1].x);
      wpair2(zfin + (size_t)5 * Nc + cn, zo[3], ps[1].y);
      wpair2(zfin + (size_t)6 * Nc + cn, zo[4], ps[2].x);
      wpair2(zfin + (size_t)8 * Nc + cn, zo[6], ps[2].y);
      wpair2(zfin + (size_t)7 * Nc + cn, zo[5], ps[3].x);
      wpair2(zfin + (size_t)9 * Nc + cn, zo[7], ps[3].y);
    }
    if (k0 + l == Nz - 1) {  // U_Nz = alb V_Nz + ru_Nz: the surface row
      if (MODE == 1 && live) zt[col] = Un;
      if (FINAL) wpair(zfin + (size_t)D * Nc + ncp, Un, zt[col + oc]);
    }
    V = Vn;
  }
}
