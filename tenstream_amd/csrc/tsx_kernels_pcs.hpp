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
#include <type_traits>

#include "tsx_pack.hpp"
#include "tsx_peer_dev.hpp"

// ---- matrix-only part: one lane per column, once per coefficient set
template <typename CT>
__global__ __launch_bounds__(64) void tsx_k_pcs_pack_col(TsxGeo g, const CT *__restrict__ C, const uint8_t *__restrict__ l1d,
                                                         const double *__restrict__ a11, const double *__restrict__ a12,
                                                         const double *__restrict__ albedo, uint4 *__restrict__ P,
                                                         const int *__restrict__ cidx = nullptr, long long nent = 0) {
  // cidx != null: C is the shared storage of bit-identical blocks, plane-major C[q * nent + cidx[cell]] (the LUT path may
  // have skipped the dense planes altogether, tsx_dedup_from_coords); same values, so the same records
  constexpr int D = 10;
  const int col = blockIdx.x * 64 + threadIdx.x;
  if (col >= g.ncol) return;
  const long long Nc = cidx ? nent : g.Nc;
  const long long sp = tsx_split_col(col % g.xm, col / g.xm, g.xm);
  double A = albedo[col];
  for (int k = g.Nz - 1; k >= 0; --k) {
    const size_t c = (size_t)k * g.ncol + col;
    double tuu, rud, rdu, tdd;
    if (l1d[k]) {
      tuu = tdd = a11[c];
      rud = rdu = a12[c];
    } else {
      const size_t e = cidx ? (size_t)cidx[c] : c;
      tuu = (double)C[(size_t)(0 * D + 0) * Nc + e];
      rud = (double)C[(size_t)(0 * D + 1) * Nc + e];
      rdu = (double)C[(size_t)(1 * D + 0) * Nc + e];
      tdd = (double)C[(size_t)(1 * D + 1) * Nc + e];
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
static __global__ __launch_bounds__(TSX_BLOCK) void tsx_k_pcs_pack_ent(int ncol, long long nent, const float *__restrict__ Cd,
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

// ---- side -> top couplings in fp16 (C16).  tests/studies/quant_study.py: of everything the preconditioner rounds, only the
// precision of the couplings from the neighbouring columns' side streams INTO the column's top streams (record 1) costs
// iterations -- they feed the exact column solve, which spreads their error through the whole column: 3 iterations with
// them exact or fp16 against 4 with fp8 e4m3 on 64 x 64 x 32 (the device: 4), 6 -> 5 on 256 x 256 x 64; the 64 side -> side
// couplings (records 4..7) can stay fp8.  Record 1 therefore becomes two fp16 records:
//   1a: c(y_q -> 0) c(y_q -> 1), q = 0..3 (8 halfs)      1b: the same for x_q
// per cell:  1a in group 1's place, 1b as group 8:  P[8 * Nc + cell]                      (9 records per cell)
// per block: PE[slot * nent + id], slot 0 = 1a, 1 = 1b, slot g = record g for g = 2..7   (8 records per distinct block)
__device__ __forceinline__ uint4 tsx_pcs_rec1_h(int half, float (&c0)[4], float (&c1)[4]) {  // c0[q] = c(src_q -> 0), c1: -> 1
  (void)half;
  return make_uint4(tsx_to_h2(c0[0], c1[0]), tsx_to_h2(c0[1], c1[1]), tsx_to_h2(c0[2], c1[2]), tsx_to_h2(c0[3], c1[3]));
}
template <typename CT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_pcs_pack_rec1h(TsxGeo g, const CT *__restrict__ C, const uint8_t *__restrict__ l1d,
                                                                  uint4 *__restrict__ P) {
  constexpr int D = 10;
  const long long Nc = g.Nc;
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const int i = (int)(c % g.xm);
    const long long t = c / g.xm;
    const int j = (int)(t % g.ym), k = (int)(t / g.ym);
    uint4 va = make_uint4(0, 0, 0, 0), vb = va;
    if (!l1d[k]) {
      float y0[4], y1[4], x0[4], x1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        y0[q] = (float)C[(size_t)(0 * D + 6 + q) * Nc + c];
        y1[q] = (float)C[(size_t)(1 * D + 6 + q) * Nc + c];
        x0[q] = (float)C[(size_t)(0 * D + 2 + q) * Nc + c];
        x1[q] = (float)C[(size_t)(1 * D + 2 + q) * Nc + c];
      }
      va = tsx_pcs_rec1_h(0, y0, y1);
      vb = tsx_pcs_rec1_h(1, x0, x1);
    }
    const size_t o = (size_t)k * g.ncol + tsx_split_col(i, j, g.xm);
    P[(size_t)1 * Nc + o] = va;
    P[(size_t)8 * Nc + o] = vb;
  }
}
constexpr int TSX_PCS_ENT16_SLOTS = 8;
// entry_major: PE[id * 8 + slot] -- an entry's eight records are one aligned 128-byte line.  For the groups of NEAR-identical
// blocks the lanes of a wave hold unrelated ids (a group's members lie anywhere in the domain), so with PE[slot * nent + id]
// every one of a level's eight gathers touches up to 64 different lines and uses 16 bytes of each; entry-major, the first
// gather brings the lane's line and the other seven hit it.  Bit-identical sharing keeps the slot-major order: there a
// wave's ids are equal (clear sky: a broadcast) or consecutive (cloud cells in cell order: coalesced).
static __global__ __launch_bounds__(TSX_BLOCK) void tsx_k_pcs_pack_ent16(int ncol, long long nent, const float *__restrict__ Cd,
                                                                  const int *__restrict__ ent_cell, const uint8_t *__restrict__ l1d,
                                                                  uint4 *__restrict__ PE, int entry_major) {
  constexpr int D = 10;
  const long long n = nent * TSX_PCS_ENT16_SLOTS;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * TSX_BLOCK) {
    const int slot = (int)(q / nent);
    const long long id = q - (long long)slot * nent;
    const int k = ent_cell[id] / ncol;
    auto cf = [&](int dst, int src) { return Cd[(size_t)(dst * D + src) * nent + id]; };
    uint4 v = make_uint4(0, 0, 0, 0);
    if (!l1d[k]) {
      if (slot < 2) {
        const int s0 = slot == 0 ? 6 : 2;
        v.x = tsx_to_h2(cf(0, s0), cf(1, s0));
        v.y = tsx_to_h2(cf(0, s0 + 1), cf(1, s0 + 1));
        v.z = tsx_to_h2(cf(0, s0 + 2), cf(1, s0 + 2));
        v.w = tsx_to_h2(cf(0, s0 + 3), cf(1, s0 + 3));
      } else if (slot < 4) {
        const int sc = slot - 2;
        v.x = tsx_to_h2(cf(2, sc), cf(3, sc));
        v.y = tsx_to_h2(cf(4, sc), cf(5, sc));
        v.z = tsx_to_h2(cf(6, sc), cf(7, sc));
        v.w = tsx_to_h2(cf(8, sc), cf(9, sc));
      } else {
        const int s0 = slot < 6 ? 6 : 2, d0 = 2 + 4 * (slot & 1);
        v.x = tsx_to_fp8x4(cf(d0 + 0, s0), cf(d0 + 0, s0 + 1), cf(d0 + 0, s0 + 2), cf(d0 + 0, s0 + 3));
        v.y = tsx_to_fp8x4(cf(d0 + 1, s0), cf(d0 + 1, s0 + 1), cf(d0 + 1, s0 + 2), cf(d0 + 1, s0 + 3));
        v.z = tsx_to_fp8x4(cf(d0 + 2, s0), cf(d0 + 2, s0 + 1), cf(d0 + 2, s0 + 2), cf(d0 + 2, s0 + 3));
        v.w = tsx_to_fp8x4(cf(d0 + 3, s0), cf(d0 + 3, s0 + 1), cf(d0 + 3, s0 + 2), cf(d0 + 3, s0 + 3));
      }
    }
    PE[entry_major ? (size_t)id * TSX_PCS_ENT16_SLOTS + slot : (size_t)q] = v;
  }
}

// ---- record 0 for the intermediate passes where the blocks are shared: they never use A_k (the last fp16 of record 0), so a
// second copy carries the cell's block index in that word -- one 16-byte load instead of record 0 + index (20 B)
static __global__ __launch_bounds__(TSX_BLOCK) void tsx_k_pcs_pack_r0g(long long Nc, const uint4 *__restrict__ P0,
                                                                const int *__restrict__ cidx_split, uint4 *__restrict__ P0G) {
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    uint4 v = P0[c];
    v.w = (unsigned)cidx_split[c];
    P0G[c] = v;
  }
}

typedef float tsx_f2 __attribute__((ext_vector_type(2)));
// (lo, hi) -> two bf16 in one word, round to nearest even: gfx950's v_cvt_pk_bf16_f32 (one instruction; the integer spelling of
// tsx_to_bf16 costs six per word, and an intermediate pass is bound by its vector instructions)
__device__ __forceinline__ unsigned tsx_bf16x2(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const tsx_f2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}
// Element i of an array whose base is the same for the whole wave, i a 32-bit lane offset: a raw buffer access
// (`buffer_load_dword v, v_off, s[rsrc:rsrc+3], 0 offen`).  The base lives in scalar registers and the byte offset is formed in
// 32 bits, so an access costs no vector instruction of address arithmetic -- with plain pointers an intermediate pass of
// tsx_k_pcs_rb spent one v_lshl_add_u64 per load and store (140 of its 1770 vector instructions), and the pass is bound by its
// vector instructions (scripts/fold_probe.sh).  Needs i * sizeof(T) < 2^32, which pcs_config checks (Nc < 2^26).  The
// descriptor: stride 0, no range limit, gfx9 dword 3 (data format 32).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tsx_rsrc(const void *base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, -1, 0x00020000);
}
// AUX: the instruction's cache-policy bits; 16 = sc1 (agent scope: the load bypasses this CU's L1, the store is written through
// the XCD's L2) -- the accesses of data that another workgroup of the SAME launch writes or reads (tsx_k_pcs_flow)
template <typename T, int AUX = 0>
__device__ __forceinline__ T tsx_ldu(const T *base, unsigned i) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8 || sizeof(T) == 16, "dword, dwordx2 or dwordx4");
  const int off = (int)(i * (unsigned)sizeof(T));
  if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(tsx_rsrc(base), off, 0, AUX));
  else if constexpr (sizeof(T) == 8) return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(tsx_rsrc(base), off, 0, AUX));
  else return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b128(tsx_rsrc(base), off, 0, AUX));
}
// the same with a wave-uniform plane offset of `plane` elements carried in the instruction's scalar offset: planes of one array
// share one descriptor (four scalar registers each -- tsx_k_pcsh_rb walks 50 planes).  Needs plane * sizeof(T) < 2^32 (pcs_config).
template <typename T, int AUX = 0>
__device__ __forceinline__ T tsx_ldo(const T *base, size_t plane, unsigned i) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8 || sizeof(T) == 16, "dword, dwordx2 or dwordx4");
  const int off = (int)(i * (unsigned)sizeof(T)), so = (int)(unsigned)(plane * sizeof(T));
  if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(tsx_rsrc(base), off, so, AUX));
  else if constexpr (sizeof(T) == 8) return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(tsx_rsrc(base), off, so, AUX));
  else return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b128(tsx_rsrc(base), off, so, AUX));
}
template <typename T, int AUX = 0>
__device__ __forceinline__ void tsx_sto(T *base, size_t plane, unsigned i, T v) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "dword or dwordx2");
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const int off = (int)(i * (unsigned)sizeof(T)), so = (int)(unsigned)(plane * sizeof(T));
  if constexpr (sizeof(T) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), tsx_rsrc(base), off, so, AUX);
  else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), tsx_rsrc(base), off, so, AUX);
}
template <typename T, int AUX = 0>
__device__ __forceinline__ void tsx_stu(T *base, unsigned i, T v) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "dword or dwordx2");
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const int off = (int)(i * (unsigned)sizeof(T));
  if constexpr (sizeof(T) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), tsx_rsrc(base), off, 0, AUX);
  else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), tsx_rsrc(base), off, 0, AUX);
}

// ---- several ranks: the other colour's boundary columns live on the neighbouring rank.  After every pass the records a
// neighbour consumes are packed per face (tsx_k_pcs_halo_pack), exchanged like the operator's halo
// (exchange_diffuse_boundary's pattern, src/pprts_explicit.F90:769-843) and read by the next pass at the rank faces --
// without it the preconditioner would drop those couplings (block-Jacobi over ranks like the reference's PCBJACOBI) and need
// 20-40 % more iterations.  Buffers: bf16-pair records [j][k] (W / E faces) and [i][k] (S / N faces); null = no exchange.
// message layout: one run of nzp = Nz rounded up to 4 words per boundary column, [j][k] (W / E faces) resp. [i][k] (S / N): a
// thread's consecutive levels are consecutive words (16-byte stores / adjacent loads when the mailbox is accessed in place)
__host__ __device__ __forceinline__ int tsx_pcs_halo_nzp(int Nz) { return (Nz + 3) & ~3; }
struct TsxPcHalo {
  const unsigned *W, *E, *S, *N;
  TsxPeerWait wait;  // peer transport (wait.mine != null): W .. N are mailbox slots, valid once the neighbour has published them
};
// my W face sends rec 0 (the -x streams of my columns i = 0: the west rank's E input), E face rec 1 of i = xm-1, S face rec 2
// of j = 0, N face rec 3 of j = ym-1; zb: bf16 records, or zr (float2 records of the fp32 pass) when from_f32
static __global__ __launch_bounds__(TSX_BLOCK) void tsx_k_pcs_halo_pack(TsxGeo g, const unsigned *__restrict__ zb,
                                                                 const float2 *__restrict__ zr, int from_f32,
                                                                 unsigned *__restrict__ sW, unsigned *__restrict__ sE,
                                                                 unsigned *__restrict__ sS, unsigned *__restrict__ sN,
                                                                 const int *__restrict__ done) {
  if (done && *done) return;
  const int xm = g.xm, ym = g.ym, Nz = g.Nz;
  const long long Nc = g.Nc;
  const int nzp = tsx_pcs_halo_nzp(Nz);
  const long long nx = g.wrap_x ? 0 : (long long)nzp * ym, ny = g.wrap_y ? 0 : (long long)nzp * xm;
  auto rec = [&](int m, int k, int i, int j) {
    const size_t idx = (size_t)m * Nc + (size_t)k * g.ncol + tsx_split_col(i, j, xm);
    if (!from_f32) return zb[idx];
    const float2 v = zr[idx];
    return (unsigned)tsx_to_bf16(v.x) | ((unsigned)tsx_to_bf16(v.y) << 16);
  };
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < nx + ny; q += (long long)gridDim.x * TSX_BLOCK) {
    if (q < nx) {
      const int k = (int)(q % nzp), j = (int)(q / nzp);
      sW[q] = k < Nz ? rec(0, k, 0, j) : 0u;
      sE[q] = k < Nz ? rec(1, k, xm - 1, j) : 0u;
    } else {
      const long long p = q - nx;
      const int k = (int)(p % nzp), i = (int)(p / nzp);
      sS[p] = k < Nz ? rec(2, k, i, 0) : 0u;
      sN[p] = k < Nz ? rec(3, k, i, ym - 1) : 0u;
    }
  }
}

// The same records stored straight into the neighbours' mailboxes (peer transport, tsx_peer_dev.hpp): pack and send in one
// kernel.  My W-face records are the west rank's E input: they land in its slot of face E (q ^ 1), and so on.
static __global__ __launch_bounds__(TSX_BLOCK) void tsx_k_pcs_halo_send(TsxGeo g, const unsigned *__restrict__ zb,
                                                                 const float2 *__restrict__ zr, int from_f32, TsxPeerXArgs a) {
  if (!tsx_peer_send_begin(a)) return;
  const int xm = g.xm, ym = g.ym, Nz = g.Nz;
  const long long Nc = g.Nc;
  const int nzp = tsx_pcs_halo_nzp(Nz);
  const long long nx = a.bytes[0] ? (long long)nzp * ym : 0, ny = a.bytes[2] ? (long long)nzp * xm : 0;
  unsigned *sW = reinterpret_cast<unsigned *>(tsx_peer_data(a.remote[0], a.data_off, a.cap, 1, (int)(a.n[0] & 1)));
  unsigned *sE = reinterpret_cast<unsigned *>(tsx_peer_data(a.remote[1], a.data_off, a.cap, 0, (int)(a.n[1] & 1)));
  unsigned *sS = reinterpret_cast<unsigned *>(tsx_peer_data(a.remote[2], a.data_off, a.cap, 3, (int)(a.n[2] & 1)));
  unsigned *sN = reinterpret_cast<unsigned *>(tsx_peer_data(a.remote[3], a.data_off, a.cap, 2, (int)(a.n[3] & 1)));
  auto rec = [&](int m, int k, int i, int j) {
    const size_t idx = (size_t)m * Nc + (size_t)k * g.ncol + tsx_split_col(i, j, xm);
    if (!from_f32) return zb[idx];
    const float2 v = zr[idx];
    return (unsigned)tsx_to_bf16(v.x) | ((unsigned)tsx_to_bf16(v.y) << 16);
  };
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < nx + ny; q += (long long)gridDim.x * TSX_BLOCK) {
    if (q < nx) {
      const int k = (int)(q % nzp), j = (int)(q / nzp);
      sW[q] = k < Nz ? rec(0, k, 0, j) : 0u;
      sE[q] = k < Nz ? rec(1, k, xm - 1, j) : 0u;
    } else {
      const long long p = q - nx;
      const int k = (int)(p % nzp), i = (int)(p / nzp);
      sS[p] = k < Nz ? rec(2, k, i, 0) : 0u;
      sN[p] = k < Nz ? rec(3, k, i, ym - 1) : 0u;
    }
  }
  tsx_peer_send_end(a, a.blkctr, gridDim.x);
}
// ---- parts of a pass on several ranks (overlap of the boundary-record exchange with the pass, tsx_pcs_apply):
// part 0 = every column of the colour; 1 = the same launch with the columns on a rank face idle (their neighbours' records
// are still in flight); 2 = only the columns on a rank face.  Frame columns of colour rbc, enumerated: the rows j = 0 and
// j = ym - 1 entirely (if the rank does not wrap onto itself in y), then of every other row the one column at i = 0 or
// i = xm - 1 that has the colour (if it does not wrap in x; xm is even, so exactly one of the two has it).
__device__ __forceinline__ int tsx_pcs_nframe(const TsxGeo &g) {
  const int h = g.xm >> 1;
  const int rows = g.wrap_y ? 0 : (g.ym >= 2 ? 2 : 1);
  return rows * h + (g.wrap_x ? 0 : g.ym - rows);
}
__device__ __forceinline__ int tsx_pcs_frame_thread(const TsxGeo &g, int rbc, int f) {  // -> jrow * h + qh
  const int h = g.xm >> 1;
  const int rows = g.wrap_y ? 0 : (g.ym >= 2 ? 2 : 1);
  if (f < rows * h) return (f < h ? 0 : g.ym - 1) * h + (f % h);
  const int jrow = (f - rows * h) + (rows ? 1 : 0);
  return jrow * h + (((jrow + rbc) & 1) ? h - 1 : 0);
}
__device__ __forceinline__ bool tsx_pcs_on_frame(const TsxGeo &g, int jrow, int icol) {
  return (!g.wrap_y && (jrow == 0 || jrow == g.ym - 1)) || (!g.wrap_x && (icol == 0 || icol == g.xm - 1));
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
// RQ (MODE 0 only): what an intermediate pass produces is bf16 anyway, so from a colour's second visit on it reads its
// right-hand side as five bf16-pair words per cell, rb[w * Nc + cell] = (ru, rd), (rs0, rs1) .. (rs6, rs7): 20 B instead of
// 40 B.  RQ 1: the colour's first visit reads fp32 and leaves those words; RQ 2: reads them; RQ 0: fp32 only.  (Measured:
// rounding the intermediate passes' right-hand side to bf16 changes no iteration count, 8 / 13 at rtol 1e-5 / 1e-8.)
// C16: the side -> top couplings (record 1) are two fp16 records (tsx_k_pcs_pack_rec1h / tsx_k_pcs_pack_ent16).
// PEER (MODE 0 / 1, peer transport): the columns on a rank face store the records their neighbour rank consumes straight into
// that rank's mailbox slot (what tsx_k_pcs_halo_pack + tsx_k_peer_send would do after the pass), the workgroup that finishes
// last publishes the sequence numbers -- an exchange without a kernel of its own (tsx_peer_dev.hpp; snd: this pass's messages).
// Analysis builds only (scripts/fold_probe.sh; never the shipped library): -DTSX_PCS_FOLD=2^n wraps the per-cell streams of an
// intermediate pass (right-hand side words, neighbour records, stored records) onto the first 2^n cells of their planes, so that they
// are served by L2 and what remains of the pass's time is its table gathers and arithmetic; -DTSX_PCS_FOLD_IDX wraps the record
// index too (the gathers then hit L1).  The results are meaningless.
#ifndef TSX_PCS_PROBE
#define TSX_PCS_PROBE 0  // analysis builds (scripts/pass_parts.sh): bit 0 no side-stream right-hand side loads in phase 3, 1 no stores of
#endif                   // the intermediate passes, 2 no neighbour loads, 3 every lane the block entry 0, 4 every lane record 0 of PT, 5 one store per thread
#ifndef TSX_PCS_COOP
#define TSX_PCS_COOP 1  // 0: entry-major per-block records gathered lane by lane also in phase 3 (A/B builds)
#endif
#ifndef TSX_PCS_STAGED
#define TSX_PCS_STAGED 0  // 1: the staged order of phase 1's loads for every layout (A/B builds)
#endif
#ifdef TSX_PCS_FOLD
#define TSX_FOLDC(x) ((size_t)(x) & (size_t)(TSX_PCS_FOLD - 1))
#else
#define TSX_FOLDC(x) (x)
#endif
#ifdef TSX_PCS_FOLD_IDX
#define TSX_FOLDI(x) TSX_FOLDC(x)
#else
#define TSX_FOLDI(x) (x)
#endif
// -DTSX_FLOW_TRACE (analysis builds, scripts/flow_trace.sh): thread 0 of a flow workgroup leaves wall-clock stamps of the stages of
// every work item in tsx_flow_tl
#ifdef TSX_FLOW_TRACE
#define TSX_FLOW_TL_N 32768
__device__ unsigned long long tsx_flow_tl[TSX_FLOW_TL_N][12];
#define TSX_TL(k)                                                  \
  do {                                                             \
    if (FLOW && tl && threadIdx.x == 0) tl[k] = wall_clock64();    \
  } while (0)
#else
#define TSX_TL(k) \
  do {            \
  } while (0)
#endif
// FLOW (tsx_k_pcs_flow below: the intermediate passes of one application inside ONE launch, a workgroup per (pass, tile) work
// item): the iterate records and right-hand side words another workgroup of the same launch wrote or will read travel as sc1
// (write-through) stores and sc1 loads -- a CU's L1 is never refreshed by another CU's stores and the XCDs' L2s are not coherent
// with each other; `tile` replaces blockIdx.x.
// HOIST (the flow kernel on small domains, where a pass is a chain of memory latencies and not bytes): every load that does not
// depend on the neighbours -- record 0, the per-block records of phase 1 AND phase 3, the right-hand side words, for all LSEG
// levels -- is issued at the top; then `wait_nbrs()` (the flow kernel: poll of the neighbour tiles' progress words + barrier);
// then the neighbour records.  About 190 registers of loads in flight: the kernel runs at two waves per SIMD.  Same arithmetic.
struct TsxNoWait {
  __device__ __forceinline__ void operator()() const {}
};
// GRAN (with HOIST): the iterate records travel as 8-byte granules {bf16 pair, tag}, tag = epoch + index of the pass that wrote
// them, each written by ONE sc1 store of one lane: the record is its own flag.  The consumer re-loads the sixteen granules of its
// four levels (sc1) until every tag is the one of the pass before -- no drain of the producer's stores, no barrier, no progress
// word, no poll of it, no separate load behind the poll: the hand-off costs one store-to-load latency (MI355X guide, R2 /
// handoff-1to1: 0.8-1.0 us against 1.7-1.9 x that for payload + flag).  A neighbour cannot overwrite a granule this tile still
// needs: its next pass needs THIS pass's granules of this tile first.  first: the neighbours' records come from the launch before
// (plain words in zb); last: the pass after this launch reads plain words -- stored beside the granules.
struct TsxGran {
  uint2 *zb8;        // [4][Nc] granules
  unsigned need;     // tag of the neighbours' granules this item reads
  unsigned tag;      // tag of the granules this item writes
  int first, last;
  unsigned long long ticks;
  int *err;
};
template <int LSEG, int NSEG, int CW, bool GS, int MODE, bool IDX, int RQ, bool C16, bool PEER, bool FLOW, bool HOIST = false,
          typename WaitF = TsxNoWait, bool GRAN = false>
__device__ __forceinline__ void tsx_pcs_rb_body(const TsxGeo &g, const uint4 *__restrict__ P, const float *__restrict__ r,
                                                float *__restrict__ z, unsigned *__restrict__ zb, float *__restrict__ zfin, int rbc,
                                                int nonbr, const int *__restrict__ cidx, long long nent,
                                                const uint4 *__restrict__ PE, const TsxPcHalo &hal, unsigned *__restrict__ rb,
                                                int part, const int *__restrict__ pidx, const uint4 *__restrict__ PT,
                                                const TsxPeerXArgs &snd, int pe_si, int tile,
                                                unsigned long long *tl = nullptr, WaitF wait_nbrs = WaitF(),
                                                const TsxGran *gr = nullptr) {
  (void)tl;
  static_assert(!GRAN || HOIST, "granules: the flow kernel's fat body");
  static_assert(!HOIST || (FLOW && GS && MODE == 0 && RQ == 2 && C16), "HOIST: an intermediate pass of the flow kernel");
  static_assert(!PEER || MODE != 2, "the last pass sends nothing");
  static_assert(!FLOW || MODE == 0, "the flow kernel runs intermediate passes");
  // (round 6: rank faces inside the flow kernel with the lean body too -- shards whose passes are not resident at once; it sends
  // after the scan like the 32-column pass kernel, the fat body from the level loop)
  constexpr int XA = FLOW ? 16 : 0;  // aux of the accesses another workgroup of the launch is on the other end of: sc1
  // per-block records: PE[slot * pe_ss + id * pe_si]; pe_si = 1: slot-major planes of nent entries, pe_si = 8 (C16 only):
  // entry-major, an entry's eight records in one 128-byte line (tsx_k_pcs_pack_ent16)
  const size_t pe_ss = pe_si == 1 ? (size_t)nent : (size_t)1;
  // pidx != null (intermediate passes with shared blocks): the cell's record 0 (with its block index) is entry pidx[cell] of
  // the table PT of distinct records (tsx_records_share)
  static_assert(RQ == 0 || MODE == 0, "bf16 right-hand side only in the intermediate passes");
  constexpr int D = 10, NTOP = 2;
  constexpr bool FINAL = MODE == 2;
  constexpr float CSC1 = C16 ? 1.0f : 1.0f / TSX_FP8_SCALE;  // the fp8 couplings are stored times TSX_FP8_SCALE
  __shared__ float2 sB[NSEG][CW], sV[NSEG][CW];
  // COOP (entry-major per-block records, pe_si = 8: every lane another 128-byte entry -- the near-identical grouping of a field
  // without identical blocks): the six records of phase 3 are fetched by groups of eight lanes, lane j of a group the record 2 + j
  // of the entry of the group's t-th lane, t = 0..7 -- an instruction touches 8 lines instead of 64 (the texture addresser is
  // 80 % busy there, one tag look-up per lane and gather) -- and handed to their lanes through LDS (rows of 7 records: no bank
  // conflicts on the way out).  One wave's rows are private to it: LDS executes a wave's instructions in order, no barrier.
  constexpr bool COOP = TSX_PCS_COOP && IDX && C16 && GS && !PEER && !HOIST && (CW == 32 || CW == 16) && (CW * NSEG) % 64 == 0;
  __shared__ uint4 sE[COOP ? CW * NSEG / 64 : 1][COOP ? 64 * 7 : 1];
  // Lane offsets are 32-bit (tsx_ldu / tsx_stu), plane bases 64-bit and wave-uniform.
  const int h = g.xm >> 1;
  const int cl = threadIdx.x % CW, sg = threadIdx.x / CW;
  const int nthr = part == 2 ? tsx_pcs_nframe(g) : g.ym * h;
  int t_ = tile * CW + cl;
  bool live = t_ < nthr;  // dead lanes shadow the last column (loads stay valid, nothing is stored)
  if (!live) t_ = nthr - 1;
  if (part == 2) t_ = tsx_pcs_frame_thread(g, rbc, t_);
  const size_t Nc = (size_t)g.Nc;
  const int Nz = g.Nz;
  const unsigned ncol = (unsigned)g.ncol;
  const int jrow = t_ / h, qh = t_ - jrow * h;
  const int par = (jrow + rbc) & 1;
  const int icol = 2 * qh + par;
  if (part == 1 && tsx_pcs_on_frame(g, jrow, icol)) live = false;
  const unsigned col = (unsigned)(jrow * g.xm + rbc * h + qh);  // colour-split column index (P, r, z, zb)
  // neighbours (other colour) in split space; 0 = no neighbour (rank face / tile edge)
  const int oc = (1 - 2 * rbc) * h;
  const int jn = jrow + 1 < g.ym ? jrow + 1 : (g.wrap_y ? 0 : -1), js = jrow > 0 ? jrow - 1 : (g.wrap_y ? g.ym - 1 : -1);
  const int qw = par ? qh : (qh > 0 ? qh - 1 : (g.wrap_x ? h - 1 : -1)), qe = par ? (qh + 1 < h ? qh + 1 : (g.wrap_x ? 0 : -1)) : qh;
  int offN = jn >= 0 ? (jn - jrow) * g.xm + oc : 0;
  int offS = js >= 0 ? (js - jrow) * g.xm + oc : 0;
  int offE = qe >= 0 ? oc + (qe - qh) : 0;
  int offW = qw >= 0 ? oc + (qw - qh) : 0;
  if (g.pc_tile_x > 0) {  // analysis knob: behave like a rank of pc_tile_x x pc_tile_y columns
    if ((icol + 1) % g.pc_tile_x == 0) offE = 0;
    if (icol % g.pc_tile_x == 0) offW = 0;
  }
  if (g.pc_tile_y > 0) {
    if ((jrow + 1) % g.pc_tile_y == 0) offN = 0;
    if (jrow % g.pc_tile_y == 0) offS = 0;
  }
  if (nonbr) offN = offS = offE = offW = 0;
  const unsigned ncp = (unsigned)(jrow * g.xm + 2 * qh);  // FINAL: natural index of the pair's first column
  auto wpair = [&](float *base, unsigned i, float mine, float partner) {
    if (live) tsx_stu(reinterpret_cast<float2 *>(base), i >> 1, par ? make_float2(partner, mine) : make_float2(mine, partner));
  };
  const float *__restrict__ rt = r + (size_t)D * Nc;
  float *__restrict__ zt = z + (size_t)D * Nc;
  float2 *__restrict__ zr = reinterpret_cast<float2 *>(z + (size_t)2 * Nc);  // side-stream records of the fp32 iterate
  const float rsurf = tsx_ldu(rt, col), V0 = tsx_ldu(rt + ncol, col);

  const int k0 = sg * LSEG;
  const int nl = Nz - k0 < LSEG ? (Nz - k0 > 0 ? Nz - k0 : 0) : LSEG;  // levels of this segment that exist
  auto level = [&](int l) { return k0 + l < Nz ? k0 + l : Nz - 1; };  // (clamped: the loads of absent levels stay valid)
  auto cell = [&](int l) { return (unsigned)level(l) * ncol + col; };

  // neighbour records of one level: [E (dofs 2,4), W (3,5), N (6,8), S (7,9)]
  // rank faces: the neighbour's records come from the exchanged buffers (bf16 pairs), [k][j] resp. [k][i]
  const bool anyface = (hal.E || hal.W || hal.N || hal.S) && !nonbr;  // wave-uniform
  const bool face[4] = {hal.E && !nonbr && qe < 0, hal.W && !nonbr && qw < 0, hal.N && !nonbr && jn < 0, hal.S && !nonbr && js < 0};
  if (!PEER && GS && hal.wait.mine) tsx_peer_wait_faces(hal.wait, face[1], face[0], face[3], face[2]);  // the records are read in place
  // PEER: which faces this column sends through (bits W, E, S, N); the slots must be free before the first store.  Only the
  // mask stays live (the kernel sits at the register count that allows four waves per SIMD)
  int sendmask = 0;
  if constexpr (PEER) {
    if (live) sendmask = (snd.bytes[0] && icol == 0 ? 1 : 0) | (snd.bytes[1] && icol == g.xm - 1 ? 2 : 0) |
                         (snd.bytes[2] && jrow == 0 ? 4 : 0) | (snd.bytes[3] && jrow == g.ym - 1 ? 8 : 0);
    if constexpr (!FLOW) {  // (the flow kernel waits per work item, for sequence numbers and progress words side by side)
      TsxPeerWait w = hal.wait;
      if (!GS) w.mine = nullptr;
      tsx_peer_begin_both(w, face[0] || face[1] || face[2] || face[3], true, snd, sendmask != 0);
    }
  }
  // word of level k in the run of this column in the slot of face f (my W records land in the west rank's slot of face E ...)
  auto send_word = [&](int f, int k, unsigned w) {
    unsigned *base = reinterpret_cast<unsigned *>(tsx_peer_data(snd.remote[f], snd.data_off, snd.cap, f ^ 1, (int)(snd.n[f] & 1)));
    base[(size_t)(f < 2 ? jrow : icol) * tsx_pcs_halo_nzp(Nz) + k] = w;
  };
  // c = cell(l).  Lanes without a neighbour in a direction get zero words (the slot they would read may hold NaN); at a rank
  // face the neighbour's record comes from the exchanged buffers (nbr_halo, a second stage: no branch between the loads of
  // the levels, so that all of them are in flight together)
  auto nbr_load = [&](unsigned c, uint2 (&o)[4]) {
    const int off[4] = {offE, offW, offN, offS};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const unsigned ci = (unsigned)((int)c + off[m]);
      if (MODE == 2) o[m] = tsx_ldo(reinterpret_cast<const uint2 *>(zr), (size_t)m * Nc, ci);
      else o[m] = make_uint2((TSX_PCS_PROBE & 4) ? 0x3f803f80u : tsx_ldo<unsigned, XA>(zb, (size_t)m * Nc, (unsigned)TSX_FOLDC(ci)), 0u);
    }
  };
  auto nbr_halo = [&](int k, unsigned (&hv)[4]) {  // unconditional loads from valid addresses
    const unsigned *hp[4] = {hal.E, hal.W, hal.N, hal.S};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const size_t hidx = (size_t)(m < 2 ? jrow : icol) * tsx_pcs_halo_nzp(Nz) + k;  // [j][k] resp. [i][k]
      const unsigned *hq = face[m] ? hp[m] + hidx : zb;
      // inside the flow kernel a slot is read again two passes later by a workgroup that may sit on the same CU: the load must not
      // be served by a line that CU's L1 still holds from then (a launch per pass starts with an invalidated L1)
      if constexpr (FLOW) hv[m] = __hip_atomic_load(hq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      else hv[m] = *hq;
    }
  };
  auto nbr_select = [&](uint2 (&o)[4], const unsigned (&hv)[4], bool halo) {
    const int off[4] = {offE, offW, offN, offS};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const bool has = off[m] != 0;
      o[m].x = has ? o[m].x : 0u;
      if (MODE == 2) o[m].y = has ? o[m].y : 0u;
      if (halo && face[m]) o[m].x = hv[m];
    }
  };
  // -> values by stream: zx[q] = stream 2+q entering through an x face, zy[q] = stream 6+q through a y face
  auto nbr_vals = [&](const uint2 (&n)[4], float (&zx)[4], float (&zy)[4]) {
    float lo[4], hi[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const bool f32 = MODE == 2 && !face[m];
      lo[m] = f32 ? __uint_as_float(n[m].x) : __uint_as_float(n[m].x << 16);
      hi[m] = f32 ? __uint_as_float(n[m].y) : __uint_as_float(n[m].x & 0xffff0000u);
    }
    zx[0] = lo[0]; zx[2] = hi[0]; zx[1] = lo[1]; zx[3] = hi[1];
    zy[0] = lo[2]; zy[2] = hi[2]; zy[1] = lo[3]; zy[3] = hi[3];
  };

  // ---- phase 1: all loads of the upward scan (independent of each other), then the local scan with zero inflow
  uint4 r0[LSEG], r1[LSEG], r1x[C16 ? LSEG : 1];
  float ru[LSEG], rd[LSEG];
  uint2 nb[LSEG][4];
  unsigned eid[LSEG];  // (times pe_si: the lane offset into a slot of the per-block records)
  // group g = 1..7 of the layout: per cell P[g * Nc + cell]; per block PE[(g - 1) * nent + id], or with C16 PE[g * nent + id]
  // (slots 0 and 1 hold record 1's two fp16 halves)
  auto rec = [&](int grp, unsigned c, unsigned ei) {
    const int slot = C16 ? grp : grp - 1;  // planes in fours per descriptor, the plane within the four as scalar offset
    return IDX ? tsx_ldo(PE + (size_t)(slot & ~3) * pe_ss, (size_t)(slot & 3) * pe_ss, ei) : tsx_ldo(P + (size_t)(grp & ~3) * Nc, (size_t)(grp & 3) * Nc, c);
  };
  // four side -> top couplings as floats: fp8 word w, or the fp16 pair of words (a, b)
  auto dec4 = [&](unsigned w, unsigned a, unsigned b, float (&o)[4]) {
    if (C16) {
      const tsx_h4 h = __builtin_bit_cast(tsx_h4, make_uint2(a, b));
      o[0] = (float)h[0]; o[1] = (float)h[1]; o[2] = (float)h[2]; o[3] = (float)h[3];
    } else {
      tsx_fp8x4(w, o);
    }
  };
  // The loads are written stage by stage over the levels (indices -> independent words -> records behind the indices), with
  // the wave-uniform decisions outside the loops: a branch per level fences the levels' loads off from each other, and a
  // wave walks the chain index -> record -> block records four times in a row (it did: 33.4 us per pass against 30.1 us)
  uint4 hw[HOIST ? LSEG : 1][6];     // HOIST: records 2..7 of every level
  unsigned hrs[HOIST ? LSEG : 1][4];  // ... and the four side-stream right-hand side words
  if constexpr (HOIST) {
    unsigned pi[LSEG];
    if (IDX && pidx) {
#pragma unroll
      for (int l = 0; l < LSEG; ++l) pi[l] = (unsigned)tsx_ldu(pidx, cell(l));
    }
#pragma unroll
    for (int l = 0; l < LSEG; ++l) {
      const unsigned c = cell(l);
      const unsigned w = tsx_ldu(rb, c);
      ru[l] = __uint_as_float(w << 16);
      rd[l] = __uint_as_float(w & 0xffff0000u);
#pragma unroll
      for (int q = 0; q < 4; ++q) hrs[l][q] = tsx_ldo(rb, (size_t)(1 + q) * Nc, c);
    }
#pragma unroll
    for (int l = 0; l < LSEG; ++l) {
      if (IDX) r0[l] = pidx ? tsx_ldu(PT, pi[l]) : tsx_ldo(P + (size_t)4 * Nc, (size_t)3 * Nc, cell(l));
      else r0[l] = tsx_ldu(P, cell(l));
      eid[l] = IDX ? r0[l].w * (unsigned)pe_si : 0u;
    }
#pragma unroll
    for (int l = 0; l < LSEG; ++l) {
      const unsigned c = cell(l);
      r1[l] = IDX ? tsx_ldu(PE, eid[l]) : tsx_ldo(P, Nc, c);
      r1x[l] = IDX ? tsx_ldo(PE, pe_ss, eid[l]) : tsx_ldu(P + (size_t)8 * Nc, c);
    }
#pragma unroll
    for (int l = 0; l < LSEG; ++l) {
#pragma unroll
      for (int q = 0; q < 6; ++q) hw[l][q] = rec(2 + q, cell(l), eid[l]);
    }
    wait_nbrs();  // the neighbour tiles have published the pass before this one
    if (GRAN && !gr->first) {
      // (polling one granule per direction and fetching the other twelve behind it was measured slower at every size: a second
      // latency on the chain -- 64 x 64 columns 1.70 -> 1.95 ms per solve, 128 x 64 2.64 -> 2.86)
      const int off[4] = {offE, offW, offN, offS};
      const unsigned long long t0 = wall_clock64();
      bool gaveup = false;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int l = 0; l < LSEG; ++l) {
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const uint2 v = tsx_ldo<uint2, 16>(gr->zb8, (size_t)m * Nc, (unsigned)((int)cell(l) + off[m]));
            nb[l][m] = make_uint2(v.x, 0u);
            ok = ok && v.y == gr->need;
          }
        }
        if (__all(ok) || gaveup) break;
        if (wall_clock64() - t0 > gr->ticks || __hip_atomic_load(gr->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          __hip_atomic_store(gr->err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          gaveup = true;  // (one more round, then on with whatever the records hold)
          continue;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    } else {
#pragma unroll
      for (int l = 0; l < LSEG; ++l) nbr_load(cell(l), nb[l]);
    }
    if (anyface) {  // rank faces: the neighbour rank's records, in place in this rank's mailbox (uncached memory)
#pragma unroll
      for (int l = 0; l < LSEG; ++l) {
        unsigned hv[4];
        nbr_halo(level(l), hv);
        nbr_select(nb[l], hv, true);
      }
    } else {
      const unsigned none[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int l = 0; l < LSEG; ++l) nbr_select(nb[l], none, false);
    }
  } else
  if (TSX_PCS_STAGED || pe_si == 1) {
  unsigned pi[LSEG];
  if (IDX && MODE == 0) {
    if (pidx) {
#pragma unroll
      for (int l = 0; l < LSEG; ++l) pi[l] = (TSX_PCS_PROBE & 16) ? 0u : (unsigned)tsx_ldu(pidx, (unsigned)TSX_FOLDI(cell(l)));
    }
  } else if (IDX) {
#pragma unroll
    for (int l = 0; l < LSEG; ++l) pi[l] = (unsigned)tsx_ldu(cidx, cell(l));
  }
#pragma unroll
  for (int l = 0; l < LSEG; ++l) {
    const unsigned c = cell(l);
    if (RQ == 2) {
      const unsigned w = tsx_ldu<unsigned, XA>(rb, (unsigned)TSX_FOLDC(c));
      ru[l] = __uint_as_float(w << 16);
      rd[l] = __uint_as_float(w & 0xffff0000u);
    } else {
      ru[l] = tsx_ldu(r, c);
      rd[l] = tsx_ldo(r, Nc, c);
    }
    if (GS) nbr_load(c, nb[l]);
  }
  if (IDX && MODE == 0) {  // record 0 with the block index in place of A_k (tsx_k_pcs_pack_r0g)
    if (pidx) {
#pragma unroll
      for (int l = 0; l < LSEG; ++l) r0[l] = tsx_ldu(PT, pi[l]);
    } else {
#pragma unroll
      for (int l = 0; l < LSEG; ++l) r0[l] = tsx_ldo(P + (size_t)4 * Nc, (size_t)3 * Nc, cell(l));
    }
#pragma unroll
    for (int l = 0; l < LSEG; ++l) eid[l] = (TSX_PCS_PROBE & 8) ? 0u : r0[l].w * (unsigned)pe_si;
  } else {
#pragma unroll
    for (int l = 0; l < LSEG; ++l) {
      eid[l] = IDX ? pi[l] * (unsigned)pe_si : 0u;
      r0[l] = tsx_ldu(P, cell(l));
    }
  }
  if (GS) {
#pragma unroll
    for (int l = 0; l < LSEG; ++l) {
      const unsigned c = cell(l);
      if (C16) {
        r1[l] = IDX ? tsx_ldu(PE, eid[l]) : tsx_ldo(P, Nc, c);
        r1x[l] = IDX ? tsx_ldo(PE, pe_ss, eid[l]) : tsx_ldu(P + (size_t)8 * Nc, c);
      } else {
        r1[l] = rec(1, c, eid[l]);
      }
    }
    if (anyface) {
#pragma unroll
      for (int l = 0; l < LSEG; ++l) {
        unsigned hv[4];
        nbr_halo(level(l), hv);
        nbr_select(nb[l], hv, true);
      }
    } else {
      const unsigned none[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int l = 0; l < LSEG; ++l) nbr_select(nb[l], none, false);
    }
  }
  } else {
    // entry-major per-block records (every lane another entry: the gathers of one level already occupy the L1): level by level
#pragma unroll
    for (int l = 0; l < LSEG; ++l) {
      const unsigned c = cell(l);
      if (IDX && MODE == 0) {
        r0[l] = pidx ? tsx_ldu(PT, (unsigned)tsx_ldu(pidx, (unsigned)TSX_FOLDI(c))) : tsx_ldo(P + (size_t)4 * Nc, (size_t)3 * Nc, c);
        eid[l] = r0[l].w * (unsigned)pe_si;
      } else {
        eid[l] = IDX ? (unsigned)tsx_ldu(cidx, c) * (unsigned)pe_si : 0u;
        r0[l] = tsx_ldu(P, c);
      }
      if (RQ == 2) {
        const unsigned w = tsx_ldu<unsigned, XA>(rb, (unsigned)TSX_FOLDC(c));
        ru[l] = __uint_as_float(w << 16);
        rd[l] = __uint_as_float(w & 0xffff0000u);
      } else {
        ru[l] = tsx_ldu(r, c);
        rd[l] = tsx_ldo(r, Nc, c);
      }
      if (GS) {
        if (C16) {
          r1[l] = IDX ? tsx_ldu(PE, eid[l]) : tsx_ldo(P, Nc, c);
          r1x[l] = IDX ? tsx_ldo(PE, pe_ss, eid[l]) : tsx_ldu(P + (size_t)8 * Nc, c);
        } else {
          r1[l] = rec(1, c, eid[l]);
        }
        nbr_load(c, nb[l]);
        unsigned hv[4] = {0u, 0u, 0u, 0u};
        if (anyface) nbr_halo(level(l), hv);
        nbr_select(nb[l], hv, anyface);
      }
    }
  }
  if (RQ == 1) {
#pragma unroll
    for (int l = 0; l < LSEG; ++l)
      if (live && l < nl) tsx_stu<unsigned, XA>(rb, cell(l), tsx_bf16x2(ru[l], rd[l]));
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
        const uint4 ry = r1[l], rx = C16 ? r1x[l] : r1[l];
        dec4(ry.x, ry.x, ry.y, cy0);
        dec4(ry.y, ry.z, ry.w, cy1);
        dec4(rx.z, rx.x, rx.y, cx0);
        dec4(rx.w, rx.z, rx.w, cx1);
        float gu8 = cy0[0] * zy[0] + cy0[2] * zy[1] + cy1[0] * zy[2] + cy1[2] * zy[3];
        float gd8 = cy0[1] * zy[0] + cy0[3] * zy[1] + cy1[1] * zy[2] + cy1[3] * zy[3];
        gu8 += cx0[0] * zx[0] + cx0[2] * zx[1] + cx1[0] * zx[2] + cx1[2] * zx[3];
        gd8 += cx0[1] * zx[0] + cx0[3] * zx[1] + cx1[1] * zx[2] + cx1[3] * zx[3];
        gu = gu8 * CSC1;
        gd = gd8 * CSC1;
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
  TSX_TL(3);
  __syncthreads();
  TSX_TL(4);
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
  TSX_TL(5);
  float Vin = V0;  // V at the top level of this segment
  for (int s2 = 0; s2 < sg; ++s2) {
    const float2 v = sV[s2][cl];
    Vin = v.x + v.y * Vin;
  }
  // ---- phase 3: true V, U; side streams; stores
  if (sg == 0) {  // tail rows: TOA Edn (identity row) and the side dummies at level Nz
    if (MODE == 1 && live) zt[(size_t)ncol + col] = V0;
    if (FINAL) wpair(zfin + (size_t)D * Nc + (size_t)ncol, ncp, V0, zt[(size_t)ncol + col + oc]);
#pragma unroll
    for (int d = NTOP; d < D; ++d) {
      const float v = rt[(size_t)d * ncol + col];
      if (MODE == 1 && live) zt[(size_t)d * ncol + col] = v;
      if (FINAL) wpair(zfin + (size_t)D * Nc + (size_t)d * ncol, ncp, v, zt[(size_t)d * ncol + col + oc]);
    }
  }
  // (only the stores are predicated.  Measured and dropped, scripts/ab_lib.sh on one box: a second copy of this loop with
  // unconditional stores behind a wave-uniform "every lane stores every level" test, so that the scheduler may start a level's
  // loads under the arithmetic of the one before: pass 30.1 -> 30.8 us; the four right-hand side words per level fetched at the
  // top of the kernel straight into LDS (`buffer_load_dword ... lds`): 30.1 -> 32.7 us -- the pass is bound by the texture
  // addresser's cycles (TA busy 65-74 %, scripts/pass_pmc.sh), not by the latency of this chain, and each DMA is one more
  // vector-memory instruction)
  float probe_acc = 0.0f;
  float V = Vin;
#pragma unroll
  for (int l = 0; l < LSEG; ++l) {
    const bool st = live && l < nl;
    const unsigned c = cell(l);
    const unsigned cn = (unsigned)level(l) * ncol + ncp;
    auto wpair2 = [&](float *base, float mine, float partner) {
      if (st) tsx_stu(reinterpret_cast<float2 *>(base), cn >> 1, par ? make_float2(partner, mine) : make_float2(mine, partner));
    };
    const tsx_h8 m = __builtin_bit_cast(tsx_h8, r0[l]);
    uint4 wcu, wcv, wy[2], wx[2];
    if constexpr (HOIST) {
      wcu = hw[l][0];
      wcv = hw[l][1];
      wy[0] = hw[l][2];
      wy[1] = hw[l][3];
      wx[0] = hw[l][4];
      wx[1] = hw[l][5];
    } else
    if (COOP && pe_si == TSX_PCS_ENT16_SLOTS) {
      const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6), g8 = lane & 56, j = lane & 7;
      const unsigned jj = (unsigned)(j < 6 ? j : 5);  // (lanes 6 and 7 of a group repeat record 7: no branch around the loads)
      uint4 got[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const unsigned et = (unsigned)__builtin_amdgcn_ds_bpermute((g8 + t) << 2, (int)eid[l]);
        got[t] = tsx_ldu(PE + 2, et + jj);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (j < 6) sE[wv][(g8 + t) * 7 + j] = got[t];
      __builtin_amdgcn_wave_barrier();
      wcu = sE[wv][lane * 7 + 0];
      wcv = sE[wv][lane * 7 + 1];
      wy[0] = sE[wv][lane * 7 + 2];
      wy[1] = sE[wv][lane * 7 + 3];
      wx[0] = sE[wv][lane * 7 + 4];
      wx[1] = sE[wv][lane * 7 + 5];
      __builtin_amdgcn_wave_barrier();
    } else {
      wcu = rec(2, c, eid[l]);
      wcv = rec(3, c, eid[l]);
      if (GS) {
        wy[0] = rec(4, c, eid[l]);
        wy[1] = rec(5, c, eid[l]);
        wx[0] = rec(6, c, eid[l]);
        wx[1] = rec(7, c, eid[l]);
      }
    }
    float rs[8];
    if (RQ == 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned w = HOIST ? hrs[HOIST ? l : 0][q] : (TSX_PCS_PROBE & 1) ? 0x3f803f80u : tsx_ldo<unsigned, XA>(rb, (size_t)(1 + q) * Nc, (unsigned)TSX_FOLDC(c));
        rs[2 * q] = __uint_as_float(w << 16);
        rs[2 * q + 1] = __uint_as_float(w & 0xffff0000u);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) rs[q] = tsx_ldo(r, (size_t)(NTOP + q) * Nc, c);
      if (RQ == 1 && st) {
#pragma unroll
        for (int q = 0; q < 4; ++q) tsx_sto<unsigned, XA>(rb, (size_t)(1 + q) * Nc, c, tsx_bf16x2(rs[2 * q], rs[2 * q + 1]));
      }
    }
    float pt[2];
    float2 ps[4];
    if (FINAL) {
      const unsigned co = (unsigned)((int)c + oc);
      pt[0] = tsx_ldu(z, co);
      pt[1] = tsx_ldo(z, Nc, co);
#pragma unroll
      for (int m2 = 0; m2 < 4; ++m2) ps[m2] = tsx_ldo(zr, (size_t)m2 * Nc, co);
    }
    const float Bn = l + 1 < LSEG ? Bk[l + 1] : Bin;
    const float Vn = Vloc[l] + Qcum[l] * Vin;
    const float Un = (float)m[5] * Vn + Bn;
    const float U = (float)m[6] * V + Bk[l];
    if (MODE == 1 && st) {
      tsx_stu(z, c, U);
      tsx_sto(z, Nc, c, Vn);
    }
    if (FINAL) {
      wpair2(zfin, U, pt[0]);
      wpair2(zfin + Nc, Vn, pt[1]);
    }
    float zx[4], zy[4];
    if (GS) nbr_vals(nb[l], zx, zy);
    const tsx_f2 zy01 = {zy[0], zy[1]}, zy23 = {zy[2], zy[3]}, zx01 = {zx[0], zx[1]}, zx23 = {zx[2], zx[3]};
    const tsx_h8 hcu = __builtin_bit_cast(tsx_h8, wcu), hcv = __builtin_bit_cast(tsx_h8, wcv);
    const unsigned uy[8] = {wy[0].x, wy[0].y, wy[0].z, wy[0].w, wy[1].x, wy[1].y, wy[1].z, wy[1].w};
    const unsigned ux[8] = {wx[0].x, wx[0].y, wx[0].z, wx[0].w, wx[1].x, wx[1].y, wx[1].z, wx[1].w};
    float zo[8];
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) {
      float acc = (float)hcu[dd] * Un + (float)hcv[dd] * V;
      if (GS) {
        // eight fp8 couplings against the eight entering streams as four packed fp32 FMAs (v_pk_fma_f32: the pairs that
        // v_cvt_pk_f32_fp8 delivers times the pairs of neighbour values), then one horizontal add
        tsx_f2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)uy[dd], false) * zy01;
        a = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)uy[dd], true), zy23, a);
        a = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)ux[dd], false), zx01, a);
        a = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)ux[dd], true), zx23, a);
        acc += (a.x + a.y) * (1.0f / TSX_FP8_SCALE);
      }
      zo[dd] = rs[dd] + acc;
    }
    // records: (2,4) (3,5) (6,8) (7,9)  = zo[0,2] zo[1,3] zo[4,6] zo[5,7]
    if (MODE == 0 && (TSX_PCS_PROBE & 32)) {  // one store per thread instead of four per level: what the stores cost
      probe_acc += zo[0] + zo[1] + zo[2] + zo[3] + zo[4] + zo[5] + zo[6] + zo[7];
      if (l == LSEG - 1 && st) tsx_sto(zb, (size_t)0, (unsigned)TSX_FOLDC(c), __float_as_uint(probe_acc));
    } else
    if (GRAN) {
      if (st) {
        const unsigned w4[4] = {tsx_bf16x2(zo[0], zo[2]), tsx_bf16x2(zo[1], zo[3]), tsx_bf16x2(zo[4], zo[6]), tsx_bf16x2(zo[5], zo[7])};
#pragma unroll
        for (int m2 = 0; m2 < 4; ++m2) tsx_sto<uint2, 16>(gr->zb8, (size_t)m2 * Nc, c, make_uint2(w4[m2], gr->tag));
        if (gr->last) {
#pragma unroll
          for (int m2 = 0; m2 < 4; ++m2) tsx_sto<unsigned, 0>(zb, (size_t)m2 * Nc, c, w4[m2]);
        }
      }
    } else
    if (MODE == 0 && st && !(TSX_PCS_PROBE & 2)) {
      tsx_sto<unsigned, XA>(zb, (size_t)0 * Nc, (unsigned)TSX_FOLDC(c), tsx_bf16x2(zo[0], zo[2]));
      tsx_sto<unsigned, XA>(zb, (size_t)1 * Nc, (unsigned)TSX_FOLDC(c), tsx_bf16x2(zo[1], zo[3]));
      tsx_sto<unsigned, XA>(zb, (size_t)2 * Nc, (unsigned)TSX_FOLDC(c), tsx_bf16x2(zo[4], zo[6]));
      tsx_sto<unsigned, XA>(zb, (size_t)3 * Nc, (unsigned)TSX_FOLDC(c), tsx_bf16x2(zo[5], zo[7]));
    }
    if constexpr (PEER && (FLOW ? HOIST : CW < 32)) {
      // small passes (16-column workgroups: at most one workgroup per CU's worth of columns, latency-bound, registers to spare):
      // the boundary columns send from the loop, so that the stores' acknowledgements arrive under the remaining levels;
      // the 32-column kernel sends after the scan (below)
      if (sendmask && st) {
        if (sendmask & 1) send_word(0, k0 + l, tsx_bf16x2(zo[0], zo[2]));
        if (sendmask & 2) send_word(1, k0 + l, tsx_bf16x2(zo[1], zo[3]));
        if (sendmask & 4) send_word(2, k0 + l, tsx_bf16x2(zo[4], zo[6]));
        if (sendmask & 8) send_word(3, k0 + l, tsx_bf16x2(zo[5], zo[7]));
      }
    }
    if (MODE == 1 && st) {
      tsx_sto(zr, (size_t)0 * Nc, c, make_float2(zo[0], zo[2]));
      tsx_sto(zr, (size_t)1 * Nc, c, make_float2(zo[1], zo[3]));
      tsx_sto(zr, (size_t)2 * Nc, c, make_float2(zo[4], zo[6]));
      tsx_sto(zr, (size_t)3 * Nc, c, make_float2(zo[5], zo[7]));
    }
    if (FINAL) {
      wpair2(zfin + (size_t)2 * Nc, zo[0], ps[0].x);
      wpair2(zfin + (size_t)4 * Nc, zo[2], ps[0].y);
      wpair2(zfin + (size_t)3 * Nc, zo[1], ps[1].x);
      wpair2(zfin + (size_t)5 * Nc, zo[3], ps[1].y);
      wpair2(zfin + (size_t)6 * Nc, zo[4], ps[2].x);
      wpair2(zfin + (size_t)8 * Nc, zo[6], ps[2].y);
      wpair2(zfin + (size_t)7 * Nc, zo[5], ps[3].x);
      wpair2(zfin + (size_t)9 * Nc, zo[7], ps[3].y);
    }
    if (k0 + l == Nz - 1) {  // U_Nz = alb V_Nz + ru_Nz: the surface row
      if (MODE == 1 && live) zt[col] = Un;
      if (FINAL) wpair(zfin + (size_t)D * Nc, ncp, Un, zt[(size_t)((int)col + oc)]);
    }
    V = Vn;
  }
  if constexpr (PEER && FLOW && !HOIST) {
    // the flow kernel's lean body (four waves per SIMD: no registers for stores inside the level loop): the face columns re-read
    // the records they have just stored -- their own sc1 stores, in program order -- and store them into the neighbours' slots; the
    // flow kernel's drain and tags follow (tsx_k_pcs_flow)
    if (sendmask) {
#pragma unroll
      for (int l = 0; l < LSEG; ++l) {
        if (l >= nl) continue;
        const unsigned c = cell(l);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          if (!(sendmask & (1 << f))) continue;
          send_word(f, k0 + l, tsx_ldo<unsigned, XA>(zb, (size_t)f * Nc, c));
        }
      }
    }
  }
  if constexpr (PEER && !FLOW) {
    // the columns on a rank face: the records just stored (rec 0 of i = 0 westwards, 1 of i = xm - 1, 2 of j = 0, 3 of j = ym - 1,
    // like tsx_k_pcs_halo_pack) go into the neighbours' slots.  Re-read here, after the scan, where few registers are live: the
    // stores inside the level loop cost 35 registers = a wave per SIMD (163 against 128 VGPRs)
    if (CW >= 32 && sendmask) {
#pragma unroll
      for (int l = 0; l < LSEG; ++l) {
        if (l >= nl) continue;
        const unsigned c = cell(l);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          if (!(sendmask & (1 << f))) continue;
          unsigned w;
          if (MODE == 0) {
            w = zb[(size_t)f * Nc + c];
          } else {
            const float2 v = zr[(size_t)f * Nc + c];
            w = tsx_bf16x2(v.x, v.y);
          }
          send_word(f, k0 + l, w);
        }
      }
    }
    tsx_peer_send_end(snd, snd.blkctr, gridDim.x);
  }
}

// one pass per launch: a workgroup per CW columns of the colour
template <int LSEG, int NSEG, int CW, bool GS, int MODE, bool IDX = false, int RQ = 0, bool C16 = false, bool PEER = false>
__global__ __launch_bounds__(CW *NSEG, PEER && MODE == 0 && CW >= 32 ? 4 : 1) void tsx_k_pcs_rb(TsxGeo g, const uint4 *__restrict__ P, const float *__restrict__ r,
                                                         float *__restrict__ z, unsigned *__restrict__ zb,
                                                         float *__restrict__ zfin, const int *__restrict__ done, int rbc,
                                                         int nonbr, const int *__restrict__ cidx, long long nent,
                                                         const uint4 *__restrict__ PE, TsxPcHalo hal,
                                                         unsigned *__restrict__ rb, int part, const int *__restrict__ pidx,
                                                         const uint4 *__restrict__ PT, TsxPeerXArgs snd, int pe_si) {
  if (done && *done) return;
  tsx_pcs_rb_body<LSEG, NSEG, CW, GS, MODE, IDX, RQ, C16, PEER, false>(g, P, r, z, zb, zfin, rbc, nonbr, cidx, nent, PE, hal, rb, part,
                                                                       pidx, PT, snd, pe_si, (int)blockIdx.x);
}

// ---- the intermediate passes of one application as ONE launch ("flow" kernel, round 5).
// A pass of a small domain is latency, not bytes: 128 x 64 columns: 10.9 us per launch for 2.9 us of bytes (launch, prologue, the
// chain index -> record -> block records, two barriers, the drain of the stores; profiles/r04/shard_study.txt), and at
// 256 x 256 still 8.2 of 30.4 us (profiles/r04/pass_parts.txt) -- times 26 intermediate passes per application.  What a pass
// needs from the one before is local: a tile of CW columns of one row reads the other colour's records of the same tile
// position, of the next tile of its row and of the rows above and below -- four tiles.  So the work items (pass p, tile t) of
// passes [p0, p1) are handed out by ONE ticket counter in the order (p, t) to the workgroups of one launch, and an item waits for
// the progress words of its four neighbour tiles instead of a kernel boundary: a tile of pass p + 1 starts as soon as ITS
// neighbours have finished pass p, no prologue, no fill and drain of the chip per pass.
//  * Deadlock-free whatever the residency: an item waits only for items with SMALLER tickets, and a ticket is taken by a
//    workgroup that is running -- by induction over the ticket order every taken item completes.  No co-residency assumption,
//    no cooperative launch; several such launches (config 4's instances on their own streams) cannot starve each other.
//  * The neighbour relation is symmetric (the tiles whose records I read are the tiles that read mine), so waiting for them to
//    finish pass p - 1 orders both the reads of their new records and the overwrite of mine that they were still reading.
//  * Visibility (MI355X: a CU's L1 is never refreshed by other CUs' stores, the XCDs' L2s are not coherent with each other):
//    the iterate records and progress words are written by sc1 (write-through) stores, every storing wave drains its stores
//    (s_waitcnt vmcnt(0)) before the workgroup's barrier, then ONE lane publishes the progress word with an agent-scope store;
//    the consumer polls with agent-scope loads from lanes 0..3, joins the workgroup's barrier, and every load of records is an
//    sc1 load (tsx_pcs_rb_body<..., FLOW>).  Results are bit-identical to the launch-per-pass path (same arithmetic per cell):
//    tests/test_gpu_parity.py::test_flow_kernel_is_bit_identical_to_launch_per_pass.
//  * Every wait is bounded (ticks of the 100 MHz wall clock): on expiry the state's error word is set, the kernel goes on with
//    whatever the records hold and the host reports TSX_ERR_HIP at its next look at the stop flag (krylov_run).
struct TsxFlowState {  // device memory, zero at allocation, one per solver
  unsigned ticket;     // next work item of the running launch; reset by the workgroup that leaves last
  unsigned exited;     // workgroups that have left the running launch
  unsigned epoch;      // progress words written by earlier launches are <= epoch; advanced by the workgroup that leaves last
  unsigned pad;
};
struct TsxFlowArgs {
  TsxFlowState *st;
  unsigned *prog;      // [2][ntiles]: last finished item of the tile, as epoch + (p - p0) + 1
  int p0, p1;          // passes [p0, p1) of the application (colour = p & 1)
  int ntiles, R;       // tiles per pass, tiles per row of columns
  unsigned long long ticks;
  int *err;            // set to 1 when a bounded wait expires (TsxScalars::flow_err: the host sees it with the stop flag)
  uint2 *zb8;          // FAT: the iterate records as granules (TsxGran), [4][Nc]
  // several ranks (FPEER, peer transport): the tiles on a rank face store the records the neighbour rank consumes into its
  // mailbox slot like tsx_k_pcs_rb<..., PEER> does, a face's last tile of a pass publishes that message's sequence number, the
  // tiles of the next pass wait for the neighbour's.  Message numbers through face q: the pass p0 + pp reads R0[q] + pp (R0: the
  // message of the launch before, already counted) and sends S0[q] + pp + 1.  No acknowledgements inside the launch: the
  // neighbour can only overwrite a slot (message n + 2) after it has seen this rank's message n + 1, which this rank sends
  // after its reads of n.
  const TsxFlowPeer *prp;           // device-resident: what does not change from launch to launch (the arguments live in scalar registers)
  unsigned long long R0[4], S0[4];  // messages received / sent through the faces before this launch
};
template <int LSEG, int NSEG, int CW, bool IDX, bool C16, bool FAT, bool GRANV = false, bool FPEER = false>
__global__ __launch_bounds__(CW *NSEG, FAT ? 2 : 4) void tsx_k_pcs_flow(TsxGeo g, const uint4 *__restrict__ P,
                                                                        const float *__restrict__ r, unsigned *__restrict__ zb,
                                                                        const int *__restrict__ done, const int *__restrict__ cidx,
                                                                        long long nent, const uint4 *__restrict__ PE,
                                                                        unsigned *__restrict__ rb, const int *__restrict__ pidx,
                                                                        const uint4 *__restrict__ PT, int pe_si, TsxFlowArgs f) {
  if (done && *done) return;
  __shared__ unsigned s_tk;
  const unsigned epoch = __hip_atomic_load(&f.st->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  static_assert(!FPEER || !GRANV, "rank faces: progress words (fat or lean body)");
  TsxPcHalo hal;
  hal.W = hal.E = hal.S = hal.N = nullptr;
  hal.wait.mine = nullptr;
  TsxPeerXArgs snd;
  snd.bytes[0] = snd.bytes[1] = snd.bytes[2] = snd.bytes[3] = 0;
  TsxFlowPeer pr;
  if constexpr (FPEER) pr = *f.prp;
  else memset((void *)&pr, 0, sizeof(pr));
  if constexpr (FPEER) {
    snd.mine = pr.mine;
    snd.cap = pr.cap;
    snd.data_off = pr.data_off;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      snd.remote[q] = pr.remote[q];
      snd.bytes[q] = pr.remote[q] ? 1 : 0;
    }
  }
  const unsigned ntiles = (unsigned)f.ntiles, nitems = (unsigned)(f.p1 - f.p0) * ntiles;
  if (threadIdx.x == 0) s_tk = atomicAdd(&f.st->ticket, 1u);
  __syncthreads();
  unsigned tk = s_tk;
  while (tk < nitems) {
    const unsigned pp = tk / ntiles, t = tk - pp * ntiles;
    const int rbc = (f.p0 + (int)pp) & 1;
    [[maybe_unused]] constexpr bool FLOW = true;
#ifdef TSX_FLOW_TRACE
    unsigned long long *tl = tsx_flow_tl[tk % TSX_FLOW_TL_N];
    if (threadIdx.x == 0) {
      tl[0] = wall_clock64();
      tl[10] = ((unsigned long long)pp << 32) | t;
      tl[11] = (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) | ((unsigned long long)blockIdx.x << 8);  // XCC_ID
    }
#else
    unsigned long long *tl = nullptr;
#endif
    unsigned nxt = 0;
    if (threadIdx.x == 0) nxt = atomicAdd(&f.st->ticket, 1u);  // the next item's ticket travels under this item's work
    // wait for the four tiles of the other colour whose records this tile reads (and which read this tile's): same position, the
    // next (previous) tile of the row where the columns' parity shifts them east (west), the rows above and below.  Lanes 0..3
    // poll one progress word each; the barrier releases the workgroup
    // which rank faces this tile's columns touch (they read the neighbour rank's records through them and send theirs): W / E where
    // the row's parity puts this colour's column at i = 0 / i = xm - 1, S / N in the first / last row
    [[maybe_unused]] int tface = 0;
    if constexpr (FPEER) {
      const int R = f.R, jrow = (int)t / R, tq = (int)t - jrow * R, parb = (jrow + rbc) & 1;
      tface = (pr.remote[0] && tq == 0 && parb == 0 ? 1 : 0) | (pr.remote[1] && tq == R - 1 && parb == 1 ? 2 : 0) |
              (pr.remote[2] && jrow == 0 ? 4 : 0) | (pr.remote[3] && jrow == g.ym - 1 ? 8 : 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) snd.n[q] = f.S0[q] + pp + 1ull;
      // the neighbour's message of the pass before, in place: slot of face q, parity of its number
      const unsigned nzpw = (unsigned)tsx_pcs_halo_nzp(g.Nz);
      (void)nzpw;
      hal.W = pr.remote[0] ? reinterpret_cast<const unsigned *>(tsx_peer_data(pr.mine, pr.data_off, pr.cap, 0, (int)((f.R0[0] + pp) & 1))) : nullptr;
      hal.E = pr.remote[1] ? reinterpret_cast<const unsigned *>(tsx_peer_data(pr.mine, pr.data_off, pr.cap, 1, (int)((f.R0[1] + pp) & 1))) : nullptr;
      hal.S = pr.remote[2] ? reinterpret_cast<const unsigned *>(tsx_peer_data(pr.mine, pr.data_off, pr.cap, 2, (int)((f.R0[2] + pp) & 1))) : nullptr;
      hal.N = pr.remote[3] ? reinterpret_cast<const unsigned *>(tsx_peer_data(pr.mine, pr.data_off, pr.cap, 3, (int)((f.R0[3] + pp) & 1))) : nullptr;
    }
    auto wait_nbrs = [&]() {
      if constexpr (FPEER) {
        if (pp == 0) {  // the neighbours' messages of the launch before: their sequence numbers, as any consumer in place
          if (threadIdx.x >= 4 && threadIdx.x < 8 && (tface & (1 << (threadIdx.x - 4)))) {
            const int q = (int)threadIdx.x - 4;
            unsigned long long have = 0;
            if (!tsx_peer_wait_ge(&reinterpret_cast<const TsxPeerHdr *>(pr.mine)->seq[q], f.R0[q], pr.ticks, &have, pr.heavy, pr.mine)) {
              tsx_peer_fail(pr.mine, 2, q, f.R0[q], have);
              __hip_atomic_store(f.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        } else {
          // inside the launch a face is as fine-grained as the tiles: in the mailbox's tag area (its own: the payload slots serve
          // every kind of exchange) one word per face, parity and row (W, E) resp. column (S, N) -- the number of the message whose
          // records for that row / column are complete, stored by the neighbour's tile after its drain.  Lanes 4 / 5 poll this row's W / E tag, lanes 32.. this tile's columns' S or N tags
          const int lane = (int)threadIdx.x;
          int q = -1;
          unsigned idx = 0;
          if (lane == 4 && (tface & 1)) q = 0, idx = (unsigned)((int)t / f.R);
          else if (lane == 5 && (tface & 2)) q = 1, idx = (unsigned)((int)t / f.R);
          else if (lane >= 32 && lane < 32 + CW && (tface & 12)) {
            const int jrow = (int)t / f.R, tq = (int)t - jrow * f.R;
            q = (tface & 4) ? 2 : 3;
            idx = (unsigned)(2 * (tq * CW + (lane - 32)) + ((jrow + rbc) & 1));
          }
          if (q >= 0) {
            const unsigned long long n = f.R0[q] + pp;
            const unsigned *tag = reinterpret_cast<const unsigned *>(pr.mine + pr.tag_off) + ((size_t)q * 2 + (size_t)(n & 1)) * pr.tag_edge + idx;
            const unsigned long long t0 = wall_clock64();
            for (;;) {
              const unsigned v = __hip_atomic_load(tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              if ((int)(v - (unsigned)n) >= 0) break;
              if (wall_clock64() - t0 > pr.ticks || __hip_atomic_load(f.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                tsx_peer_fail(pr.mine, 2, q, n, v);
                __hip_atomic_store(f.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
              }
              __builtin_amdgcn_s_sleep(1);
            }
            if (pr.heavy) __threadfence_system();  // (a platform whose peer mapping is cached: full fences, as tsx_peer_peek does)
          }
        }
      }
      if (pp > 0 && threadIdx.x < 4) {
        const int R = f.R, jrow = (int)t / R, tq = (int)t - jrow * R;
        int dep;
        if (threadIdx.x == 0) {
          dep = (int)t;
        } else if (threadIdx.x == 1) {
          int nq = ((jrow + rbc) & 1) ? tq + 1 : tq - 1;
          if (nq < 0) nq = g.wrap_x ? R - 1 : -1;
          else if (nq >= R) nq = g.wrap_x ? 0 : -1;
          dep = nq < 0 ? -1 : jrow * R + nq;
        } else {
          int jn = threadIdx.x == 2 ? jrow + 1 : jrow - 1;
          if (jn < 0) jn = g.wrap_y ? g.ym - 1 : -1;
          else if (jn >= g.ym) jn = g.wrap_y ? 0 : -1;
          dep = jn < 0 ? -1 : jn * R + tq;
        }
        if (dep >= 0) {
          const unsigned *w = f.prog + (size_t)(1 - rbc) * ntiles + dep;
          const unsigned need = epoch + pp;  // the neighbour has finished pass p - 1
          const unsigned long long t0 = wall_clock64();
          while ((int)(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - need) < 0) {
            if (wall_clock64() - t0 > f.ticks || __hip_atomic_load(f.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
              __hip_atomic_store(f.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
        }
      }
      TSX_TL(1);
      __syncthreads();
      TSX_TL(2);
    };
    if constexpr (FAT && GRANV) {
      // the records are their own flags (TsxGran): nothing to wait for in front of them, nothing to publish behind them
      auto stamps = [&]() {
        TSX_TL(1);
        TSX_TL(2);
      };
      TsxGran gr;
      gr.zb8 = f.zb8;
      gr.need = epoch + pp;
      gr.tag = epoch + pp + 1u;
      gr.first = pp == 0;
      gr.last = (int)pp == f.p1 - f.p0 - 1;
      gr.ticks = f.ticks;
      gr.err = f.err;
      tsx_pcs_rb_body<LSEG, NSEG, CW, true, 0, IDX, 2, C16, false, true, true, decltype(stamps), true>(
          g, P, r, nullptr, zb, nullptr, rbc, 0, cidx, nent, PE, hal, rb, 0, pidx, PT, snd, pe_si, (int)t, tl, stamps, &gr);
      TSX_TL(6);
      TSX_TL(7);
      if (threadIdx.x == 0) s_tk = nxt;
      __syncthreads();
      TSX_TL(8);
      tk = s_tk;
      continue;
    } else if constexpr (FAT) {
      tsx_pcs_rb_body<LSEG, NSEG, CW, true, 0, IDX, 2, C16, FPEER, true, true, decltype(wait_nbrs)>(
          g, P, r, nullptr, zb, nullptr, rbc, 0, cidx, nent, PE, hal, rb, 0, pidx, PT, snd, pe_si, (int)t, tl, wait_nbrs);
    } else {
      wait_nbrs();
      tsx_pcs_rb_body<LSEG, NSEG, CW, true, 0, IDX, 2, C16, FPEER, true>(g, P, r, nullptr, zb, nullptr, rbc, 0, cidx, nent, PE, hal, rb, 0,
                                                                         pidx, PT, snd, pe_si, (int)t, tl);
    }
    TSX_TL(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: its records have left the XCD's L2
    if constexpr (FPEER) {
      if (pr.heavy) __threadfence_system();
    }
    TSX_TL(7);
    if (threadIdx.x == 0) s_tk = nxt;
    __syncthreads();
    TSX_TL(8);
    if (threadIdx.x == 0)
      __hip_atomic_store(f.prog + (size_t)rbc * ntiles + t, epoch + pp + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if constexpr (FPEER) {
      // every wave's stores into the neighbours' slots have been acknowledged (the drain above covers them): the tags of this
      // tile's rows / columns in those slots say so
      {
        const int lane = (int)threadIdx.x;
        const int jrow = (int)t / f.R, tq = (int)t - jrow * f.R;
        int q = -1;
        unsigned idx = 0;
        if (lane == 4 && (tface & 1)) q = 0, idx = (unsigned)jrow;
        else if (lane == 5 && (tface & 2)) q = 1, idx = (unsigned)jrow;
        else if (lane >= 32 && lane < 32 + CW && (tface & 12)) {
          q = (tface & 4) ? 2 : 3;
          idx = (unsigned)(2 * (tq * CW + (lane - 32)) + ((jrow + rbc) & 1));
        }
        if (q >= 0) {
          const unsigned long long n = f.S0[q] + pp + 1ull;
          unsigned *tag = reinterpret_cast<unsigned *>(pr.remote[q] + pr.tag_off) + ((size_t)(q ^ 1) * 2 + (size_t)(n & 1)) * pr.tag_edge + idx;
          if (pr.heavy) __threadfence_system();
          __hip_atomic_store(tag, (unsigned)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    tk = s_tk;
  }
  if (threadIdx.x == 0) {
    const unsigned prev = atomicAdd(&f.st->exited, 1u);
    if (prev + 1 == gridDim.x) {  // nobody reads the ticket or the epoch any more: ready for the next launch
      if constexpr (FPEER) {
        // every message of the launch but its neighbours' last one has been consumed: acknowledge them, so that the senders after
        // this launch (which wait for the acknowledgement of their message n - 2) find their slots free
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (pr.remote[q]) {
            tsx_peer_post(&reinterpret_cast<TsxPeerHdr *>(pr.remote[q])->ack[q ^ 1], f.R0[q] + (unsigned long long)(f.p1 - f.p0) - 1ull, pr.heavy);
            // ... and every message of this rank is complete: the consumer of the last one is a kernel of its own, which waits for
            // the face's sequence number
            tsx_peer_post(&reinterpret_cast<TsxPeerHdr *>(pr.remote[q])->seq[q ^ 1], f.S0[q] + (unsigned long long)(f.p1 - f.p0), pr.heavy);
          }
      }
      __hip_atomic_store(&f.st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&f.st->exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&f.st->epoch, epoch + (unsigned)(f.p1 - f.p0) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// =====================================================================================================================
// 8_16: the same scan with H = 4 up/down pairs.  U, V, B are 4-vectors, the matrix-only quantities 4 x 4 blocks:
//     G = (I - Rdu A_{k+1})^-1,  GT = G Tdd,  H = G Rdu,  F = Tuu A_{k+1} G,  E = Tuu + F Rdu,  A_k = Rud + F Tdd
//     upward    B_k     = (ru + F rd) + E B_{k+1}            downward  V_{k+1} = (G rd + H B_{k+1}) + GT V_k
// A segment's summary is (4-vector, 4 x 4 product); the threads re-run their levels with the true inflow (cheaper than
// keeping a cumulative product per level).  Packed layout "S16H", 16-byte records:
//   per cell, P[grp * Nc + cell] (fp16 4 x 4 row-major, two records each): 0,1 E | 2,3 F | 4,5 G - I | 6,7 H | 8,9 GT |
//     10,11 A_{k+1} | 12,13 A_k
//   per block, PB[grp * stride + (cell | entry)]: 0,1 c(y_q -> t), byte 4 t + q (fp8) | 2,3 c(x_q -> t) |
//     4..11 c(src 0..7 -> side dst 8 + dd) (fp16) | 12,13 c(y_q -> 8 + dd), byte 4 dd + q | 14,15 c(x_q -> 8 + dd)
//   (y_q = src 12 + q, x_q = src 8 + q; t = top dst 0..7).  1-D layers: matrices from a11 / a12, block records zero.
#ifndef TSX_PCS_C16
#define TSX_PCS_C16 1  // side -> top couplings in fp16 (see tsx_k_pcs_pack_rec1h); 0: fp8 like the side -> side couplings
#endif
#ifndef TSX_PCSH_DEFER_ST
#define TSX_PCSH_DEFER_ST 0  // 1 (A/B builds): the 8_16 intermediate passes store their four records per level behind phase 4's level loop -- the scheduler then hoists the levels' loads until 105 registers spill (256 VGPRs); off
#endif
// With TSX_PCS_C16 the four fp8 records 0..3 of the per-block part become eight fp16 records (two top dsts per record:
// halfs 4 (t & 1) + q of record t >> 1 for the y sources, of record 4 + (t >> 1) for the x sources) and the others move up by four
constexpr int TSX_S16H_CELL = 14, TSX_S16H_BLOCK = TSX_PCS_C16 ? 20 : 16, TSX_S16H_BO = TSX_PCS_C16 ? 4 : 0;

__device__ __forceinline__ uint4 tsx_pack_rows2(const double (&M)[4][4], int r0, double sub_diag) {
  auto v = [&](int a, int b) { return (float)(M[a][b] - (a == b ? sub_diag : 0.0)); };
  return make_uint4(tsx_to_h2(v(r0, 0), v(r0, 1)), tsx_to_h2(v(r0, 2), v(r0, 3)), tsx_to_h2(v(r0 + 1, 0), v(r0 + 1, 1)),
                    tsx_to_h2(v(r0 + 1, 2), v(r0 + 1, 3)));
}

template <typename CT>
__global__ __launch_bounds__(64) void tsx_k_pcsh_pack_col(TsxGeo g, const CT *__restrict__ C, const uint8_t *__restrict__ l1d,
                                                          const double *__restrict__ a11, const double *__restrict__ a12,
                                                          const double *__restrict__ albedo, uint4 *__restrict__ P) {
  constexpr int D = 16, H = 4;
  using SM = TsxSm<H>;
  const int col = blockIdx.x * 64 + threadIdx.x;
  if (col >= g.ncol) return;
  const long long Nc = g.Nc;
  const long long sp = tsx_split_col(col % g.xm, col / g.xm, g.xm);
  double A[H][H];
#pragma unroll
  for (int a = 0; a < H; ++a)
#pragma unroll
    for (int b = 0; b < H; ++b) A[a][b] = albedo[col] / (double)H;  // assembled surface row: albedo / streams on every pair
  for (int k = g.Nz - 1; k >= 0; --k) {
    const size_t c = (size_t)k * g.ncol + col;
    double Tuu[H][H], Rud[H][H], Rdu[H][H], Tdd[H][H];
    const bool one = l1d[k] != 0;
    const double t11 = one ? a11[c] : 0.0, t12 = one ? a12[c] : 0.0;
#pragma unroll
    for (int a = 0; a < H; ++a)
#pragma unroll
      for (int b = 0; b < H; ++b) {
        const double dg = a == b ? 1.0 : 0.0;
        Tuu[a][b] = one ? dg * t11 : (double)C[(size_t)((2 * a) * D + 2 * b) * Nc + c];
        Rud[a][b] = one ? dg * t12 : (double)C[(size_t)((2 * a) * D + 2 * b + 1) * Nc + c];
        Rdu[a][b] = one ? dg * t12 : (double)C[(size_t)((2 * a + 1) * D + 2 * b) * Nc + c];
        Tdd[a][b] = one ? dg * t11 : (double)C[(size_t)((2 * a + 1) * D + 2 * b + 1) * Nc + c];
      }
    double RA[H][H], G[H][H], GT[H][H], Hm[H][H], TA[H][H], F[H][H], FR[H][H], E[H][H], FT[H][H], Ao[H][H];
    SM::matmul(Rdu, A, RA);
    SM::inv_i_minus(RA, G);
    SM::matmul(G, Tdd, GT);
    SM::matmul(G, Rdu, Hm);
    SM::matmul(Tuu, A, TA);
    SM::matmul(TA, G, F);
    SM::matmul(F, Rdu, FR);
    SM::matmul(F, Tdd, FT);
#pragma unroll
    for (int a = 0; a < H; ++a)
#pragma unroll
      for (int b = 0; b < H; ++b) {
        E[a][b] = Tuu[a][b] + FR[a][b];
        Ao[a][b] = Rud[a][b] + FT[a][b];
      }
    const size_t o = (size_t)k * g.ncol + sp;
    P[(size_t)0 * Nc + o] = tsx_pack_rows2(E, 0, 0.0);
    P[(size_t)1 * Nc + o] = tsx_pack_rows2(E, 2, 0.0);
    P[(size_t)2 * Nc + o] = tsx_pack_rows2(F, 0, 0.0);
    P[(size_t)3 * Nc + o] = tsx_pack_rows2(F, 2, 0.0);
    P[(size_t)4 * Nc + o] = tsx_pack_rows2(G, 0, 1.0);
    P[(size_t)5 * Nc + o] = tsx_pack_rows2(G, 2, 1.0);
    P[(size_t)6 * Nc + o] = tsx_pack_rows2(Hm, 0, 0.0);
    P[(size_t)7 * Nc + o] = tsx_pack_rows2(Hm, 2, 0.0);
    P[(size_t)8 * Nc + o] = tsx_pack_rows2(GT, 0, 0.0);
    P[(size_t)9 * Nc + o] = tsx_pack_rows2(GT, 2, 0.0);
    P[(size_t)10 * Nc + o] = tsx_pack_rows2(A, 0, 0.0);
    P[(size_t)11 * Nc + o] = tsx_pack_rows2(A, 2, 0.0);
    P[(size_t)12 * Nc + o] = tsx_pack_rows2(Ao, 0, 0.0);
    P[(size_t)13 * Nc + o] = tsx_pack_rows2(Ao, 2, 0.0);
#pragma unroll
    for (int a = 0; a < H; ++a)
#pragma unroll
      for (int b = 0; b < H; ++b) A[a][b] = Ao[a][b];
  }
}

// block records from planes over `n` blocks (cells in colour-split order when split_xm > 0, or distinct-block entries);
// kof: layer of block q (cell: q / ncol; entry: its representative cell's layer)
template <typename CT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_pcsh_pack_block(TsxGeo g, long long n, const CT *__restrict__ C,
                                                                   const int *__restrict__ ent_cell, const uint8_t *__restrict__ l1d,
                                                                   uint4 *__restrict__ PB) {
  constexpr int D = 16;
  const long long tot = n * TSX_S16H_BLOCK;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < tot; q += (long long)gridDim.x * TSX_BLOCK) {
    const int grp = (int)(q / n);
    const long long e = q - (long long)grp * n;
    const long long cell = ent_cell ? ent_cell[e] : e;
    const int k = (int)(cell / g.ncol);
    auto cf = [&](int dst, int src) { return (float)C[(size_t)(dst * D + src) * n + e]; };
    uint4 v = make_uint4(0, 0, 0, 0);
    if (!l1d[k]) {
      if (TSX_PCS_C16 && grp < 8) {  // couplings into the top streams in fp16: record (t >> 1) [+ 4 for the x sources]
        const int s0 = grp < 4 ? 12 : 8, t0 = 2 * (grp & 3);
        v.x = tsx_to_h2(cf(t0, s0), cf(t0, s0 + 1));
        v.y = tsx_to_h2(cf(t0, s0 + 2), cf(t0, s0 + 3));
        v.z = tsx_to_h2(cf(t0 + 1, s0), cf(t0 + 1, s0 + 1));
        v.w = tsx_to_h2(cf(t0 + 1, s0 + 2), cf(t0 + 1, s0 + 3));
      } else if (!TSX_PCS_C16 && grp < 4) {  // couplings into the top streams: word t = the four side sources of top dst t
        const int s0 = grp < 2 ? 12 : 8, t0 = 4 * (grp & 1);
        v.x = tsx_to_fp8x4(cf(t0 + 0, s0), cf(t0 + 0, s0 + 1), cf(t0 + 0, s0 + 2), cf(t0 + 0, s0 + 3));
        v.y = tsx_to_fp8x4(cf(t0 + 1, s0), cf(t0 + 1, s0 + 1), cf(t0 + 1, s0 + 2), cf(t0 + 1, s0 + 3));
        v.z = tsx_to_fp8x4(cf(t0 + 2, s0), cf(t0 + 2, s0 + 1), cf(t0 + 2, s0 + 2), cf(t0 + 2, s0 + 3));
        v.w = tsx_to_fp8x4(cf(t0 + 3, s0), cf(t0 + 3, s0 + 1), cf(t0 + 3, s0 + 2), cf(t0 + 3, s0 + 3));
      } else if (grp < 12 + TSX_S16H_BO) {  // side dst 8 + dd from the eight top sources
        const int d = 8 + (grp - 4 - TSX_S16H_BO);
        v.x = tsx_to_h2(cf(d, 0), cf(d, 1));
        v.y = tsx_to_h2(cf(d, 2), cf(d, 3));
        v.z = tsx_to_h2(cf(d, 4), cf(d, 5));
        v.w = tsx_to_h2(cf(d, 6), cf(d, 7));
      } else {  // side dst from the side sources of the neighbouring columns
        const int s0 = grp < 14 + TSX_S16H_BO ? 12 : 8, d0 = 8 + 4 * (grp & 1);
        v.x = tsx_to_fp8x4(cf(d0 + 0, s0), cf(d0 + 0, s0 + 1), cf(d0 + 0, s0 + 2), cf(d0 + 0, s0 + 3));
        v.y = tsx_to_fp8x4(cf(d0 + 1, s0), cf(d0 + 1, s0 + 1), cf(d0 + 1, s0 + 2), cf(d0 + 1, s0 + 3));
        v.z = tsx_to_fp8x4(cf(d0 + 2, s0), cf(d0 + 2, s0 + 1), cf(d0 + 2, s0 + 2), cf(d0 + 2, s0 + 3));
        v.w = tsx_to_fp8x4(cf(d0 + 3, s0), cf(d0 + 3, s0 + 1), cf(d0 + 3, s0 + 2), cf(d0 + 3, s0 + 3));
      }
    }
    long long o = e;
    if (!ent_cell) {  // per-cell records live in colour-split order
      const int i = (int)(e % g.xm);
      const long long t = e / g.xm;
      const int j = (int)(t % g.ym), kk = (int)(t / g.ym);
      o = (long long)kk * g.ncol + tsx_split_col(i, j, g.xm);
    }
    PB[(size_t)grp * n + o] = v;
  }
}

struct TsxM4 {  // a 4 x 4 block in registers
  float m[4][4];
};
// A block as loaded: fp16, rows 0 and 1 in lo, rows 2 and 3 in hi (tsx_pack_rows2).  The products read the halves in place
// (v_fma_mix_f32): converting the 16 elements first cost 750 of the pass's 4500 vector instructions and 8 registers per block
struct TsxH4 {
  tsx_h8 lo, hi;
  __device__ __forceinline__ float e(int a, int b) const { return a < 2 ? (float)lo[4 * a + b] : (float)hi[4 * (a - 2) + b]; }
};
__device__ __forceinline__ TsxH4 tsx_h4m(const uint4 &lo, const uint4 &hi) {
  TsxH4 M;
  M.lo = __builtin_bit_cast(tsx_h8, lo);
  M.hi = __builtin_bit_cast(tsx_h8, hi);
  return M;
}
// o = M v, or (M + I) v -- the layout stores G - I
template <bool PLUS_I = false>
__device__ __forceinline__ void tsx_mv4(const TsxH4 &M, const float (&v)[4], float (&o)[4]) {
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    float acc = PLUS_I ? __builtin_fmaf(M.e(a, 0), v[0], v[a]) : M.e(a, 0) * v[0];
    acc = __builtin_fmaf(M.e(a, 1), v[1], acc);
    acc = __builtin_fmaf(M.e(a, 2), v[2], acc);
    o[a] = __builtin_fmaf(M.e(a, 3), v[3], acc);
  }
}
__device__ __forceinline__ TsxM4 tsx_mm4(const TsxH4 &X, const TsxM4 &Y) {
  TsxM4 O;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float acc = X.e(a, 0) * Y.m[0][b];
      acc = __builtin_fmaf(X.e(a, 1), Y.m[1][b], acc);
      acc = __builtin_fmaf(X.e(a, 2), Y.m[2][b], acc);
      O.m[a][b] = __builtin_fmaf(X.e(a, 3), Y.m[3][b], acc);
    }
  return O;
}

#ifndef TSX_PCSH_KEEP_E
#define TSX_PCSH_KEEP_E 1  // E of a thread's levels stays in registers between the two upward phases (0: loaded again)
#endif
#ifndef TSX_PCSH_KEEP_GT
#define TSX_PCSH_KEEP_GT 1  // likewise GT between the two downward phases
#endif
#ifndef TSX_PCSH_WAVES
#define TSX_PCSH_WAVES 2
#endif
template <int LSEG, int NSEG, int CW, bool GS, int MODE, bool IDX = false, int RQ = 0>
__global__ __launch_bounds__(CW *NSEG) __attribute__((amdgpu_waves_per_eu(LSEG == 2 ? 3 : TSX_PCSH_WAVES, LSEG == 2 ? 3 : TSX_PCSH_WAVES))) void tsx_k_pcsh_rb(TsxGeo g, const uint4 *__restrict__ P, const uint4 *__restrict__ PB,
                                                          long long bstride, const int *__restrict__ cidx,
                                                          const float *__restrict__ r, float *__restrict__ z,
                                                          unsigned *__restrict__ zb, float *__restrict__ zfin,
                                                          const int *__restrict__ done, int rbc, int nonbr, TsxPcHalo hal,
                                                          unsigned *__restrict__ rb, int part, const int *__restrict__ pidx,
                                                          long long pstride) {
  // pidx != null: the 14 recurrence records are shared between cells with identical ones (tsx_records_share): P is the table
  // P[grp * pstride + pidx[cell]]; else P[grp * Nc + cell], pstride = Nc
  // RQ as in tsx_k_pcs_rb: eight bf16-pair words per cell, rb[w * Nc + cell] = (ru_a, rd_a), a = 0..3, then (rs_2q, rs_2q+1)
  static_assert(RQ == 0 || MODE == 0, "bf16 right-hand side only in the intermediate passes");
  constexpr int D = 16, NTOP = 8;
  constexpr bool FINAL = MODE == 2;
  __shared__ float4 sS[NSEG][5][CW];  // a segment's summary: 4-vector + 4 x 4 product (reused by both scans)
  if (done && *done) return;
  const int h = g.xm >> 1;
  const int cl = threadIdx.x % CW, sg = threadIdx.x / CW;
  const int nthr = part == 2 ? tsx_pcs_nframe(g) : g.ym * h;
  int t_ = blockIdx.x * CW + cl;
  bool live = t_ < nthr;
  if (!live) t_ = nthr - 1;
  if (part == 2) t_ = tsx_pcs_frame_thread(g, rbc, t_);
  // lane offsets are 32-bit (tsx_ldu / tsx_stu), plane bases 64-bit and wave-uniform, as in tsx_k_pcs_rb
  const size_t Nc = (size_t)g.Nc;
  const int Nz = g.Nz;
  const unsigned ncol = (unsigned)g.ncol;
  const int jrow = t_ / h, qh = t_ - jrow * h;
  const int par = (jrow + rbc) & 1;
  const int icol = 2 * qh + par;
  if (part == 1 && tsx_pcs_on_frame(g, jrow, icol)) live = false;
  const unsigned col = (unsigned)(jrow * g.xm + rbc * h + qh);
  const int oc = (1 - 2 * rbc) * h;
  const int jn = jrow + 1 < g.ym ? jrow + 1 : (g.wrap_y ? 0 : -1), js = jrow > 0 ? jrow - 1 : (g.wrap_y ? g.ym - 1 : -1);
  const int qw = par ? qh : (qh > 0 ? qh - 1 : (g.wrap_x ? h - 1 : -1)), qe = par ? (qh + 1 < h ? qh + 1 : (g.wrap_x ? 0 : -1)) : qh;
  int offN = jn >= 0 ? (jn - jrow) * g.xm + oc : 0;
  int offS = js >= 0 ? (js - jrow) * g.xm + oc : 0;
  int offE = qe >= 0 ? oc + (qe - qh) : 0;
  int offW = qw >= 0 ? oc + (qw - qh) : 0;
  if (g.pc_tile_x > 0) {
    if ((icol + 1) % g.pc_tile_x == 0) offE = 0;
    if (icol % g.pc_tile_x == 0) offW = 0;
  }
  if (g.pc_tile_y > 0) {
    if ((jrow + 1) % g.pc_tile_y == 0) offN = 0;
    if (jrow % g.pc_tile_y == 0) offS = 0;
  }
  if (nonbr) offN = offS = offE = offW = 0;
  const unsigned ncp = (unsigned)(jrow * g.xm + 2 * qh);
  auto wpair_if = [&](bool on, float *base, unsigned i, float mine, float partner) {
    if (on) tsx_stu(reinterpret_cast<float2 *>(base), i >> 1, par ? make_float2(partner, mine) : make_float2(mine, partner));
  };
  const float *__restrict__ rt = r + (size_t)D * Nc;
  float *__restrict__ zt = z + (size_t)D * Nc;
  float2 *__restrict__ zr = reinterpret_cast<float2 *>(z + (size_t)NTOP * Nc);
  const int k0 = sg * LSEG;
  const int nl = Nz - k0 < LSEG ? (Nz - k0 > 0 ? Nz - k0 : 0) : LSEG;
  auto level = [&](int l) { return k0 + l < Nz ? k0 + l : Nz - 1; };
  auto cell = [&](int l) { return (unsigned)level(l) * ncol + col; };
  // record planes in fours per descriptor (the plane within the four as scalar offset: 3 * stride * 16 B < 2^32, pcs_config)
  auto brec = [&](int grp, unsigned c, unsigned id) {
    return tsx_ldo(PB + (size_t)(grp & ~3) * (size_t)bstride, (size_t)(grp & 3) * (size_t)bstride, IDX ? id : c);
  };
  auto prec = [&](int grp, unsigned pr) { return tsx_ldo(P + (size_t)(grp & ~3) * (size_t)pstride, (size_t)(grp & 3) * (size_t)pstride, pr); };
  auto mat = [&](int grp, unsigned pr) { return tsx_h4m(prec(grp, pr), prec(grp + 1, pr)); };
  // rank faces: the neighbour's records come from the exchanged buffers (bf16 pairs), [k][j] resp. [k][i]
  const bool face[4] = {hal.E && !nonbr && qe < 0, hal.W && !nonbr && qw < 0, hal.N && !nonbr && jn < 0, hal.S && !nonbr && js < 0};
  if (GS && hal.wait.mine) tsx_peer_wait_faces(hal.wait, face[1], face[0], face[3], face[2]);  // the records are read in place
  const bool anyface = (hal.E || hal.W || hal.N || hal.S) && !nonbr;  // wave-uniform
  // k = level(l), c = cell(l).  Lanes without a neighbour in a direction get zero words (the slot they would read may hold NaN)
  auto nbr_load = [&](auto halo, int k, unsigned c, uint2(&o)[4]) {
    const int off[4] = {offE, offW, offN, offS};
    const unsigned *hp[4] = {hal.E, hal.W, hal.N, hal.S};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const unsigned ci = (unsigned)((int)c + off[m]);
      const bool has = off[m] != 0 || face[m];
      if (MODE == 2) o[m] = tsx_ldo(reinterpret_cast<const uint2 *>(zr), (size_t)m * Nc, ci);
      else o[m] = make_uint2(tsx_ldo(zb, (size_t)m * Nc, ci), 0u);
      if constexpr (decltype(halo)::value) {  // unconditional load from a valid address, then select
        const size_t hidx = (size_t)(m < 2 ? jrow : icol) * tsx_pcs_halo_nzp(Nz) + k;  // [j][k] resp. [i][k]
        const unsigned hv = *(face[m] ? hp[m] + hidx : zb);
        if (face[m]) o[m].x = hv;
      }
      o[m].x = has ? o[m].x : 0u;
      if (MODE == 2) o[m].y = has ? o[m].y : 0u;
    }
  };
  auto nbr_vals = [&](const uint2(&n)[4], float(&zx)[4], float(&zy)[4]) {
    float lo[4], hi[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const bool f32 = MODE == 2 && !face[m];
      lo[m] = f32 ? __uint_as_float(n[m].x) : __uint_as_float(n[m].x << 16);
      hi[m] = f32 ? __uint_as_float(n[m].y) : __uint_as_float(n[m].x & 0xffff0000u);
    }
    zx[0] = lo[0]; zx[2] = hi[0]; zx[1] = lo[1]; zx[3] = hi[1];
    zy[0] = lo[2]; zy[2] = hi[2]; zy[1] = lo[3]; zy[3] = hi[3];
  };
  auto put_summary = [&](const float(&v)[4], const TsxM4 &M) {
    sS[sg][0][cl] = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
    for (int a = 0; a < 4; ++a) sS[sg][1 + a][cl] = make_float4(M.m[a][0], M.m[a][1], M.m[a][2], M.m[a][3]);
  };
  auto chain = [&](int s2, float(&x)[4]) {  // x <- v(s2) + M(s2) x
    const float4 v = sS[s2][0][cl];
    float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float4 m = sS[s2][1 + a][cl];
      o[a] = __builtin_fmaf(m.w, x[3], __builtin_fmaf(m.z, x[2], __builtin_fmaf(m.y, x[1], __builtin_fmaf(m.x, x[0], o[a]))));
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) x[a] = o[a];
  };

  // ---- phase 1: local upward scan with zero inflow; keeps beta, rd + couplings and E of its levels
  // (E and GT of a level are loaded again where the re-run needs them -- L2 hits -- instead of being held: 32 registers
  // per level would halve the occupancy)
  float beta[LSEG][4], rdg[LSEG][4];
  uint2 nb[LSEG][4];
  unsigned eid[LSEG];
  unsigned pr[LSEG];  // row of the level's recurrence records (the cell, or its entry of the shared table)
  TsxH4 Ekeep[TSX_PCSH_KEEP_E ? LSEG : 1];  // E of the levels, held for the re-run of phase 2 (8 registers per level as fp16)
  // (the levels' indices first and the wave-uniform decisions -- shared records or not, rank faces or not -- outside the level
  // loop: a branch per level fences the levels' loads off from each other)
  if (pidx) {
#pragma unroll
    for (int l = 0; l < LSEG; ++l) pr[l] = (unsigned)tsx_ldu(pidx, cell(l));
  } else {
#pragma unroll
    for (int l = 0; l < LSEG; ++l) pr[l] = cell(l);
  }
#pragma unroll
  for (int l = 0; l < LSEG; ++l) eid[l] = IDX ? (unsigned)tsx_ldu(cidx, cell(l)) : 0u;
  auto phase1 = [&](auto halo) {
    float Bl[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    TsxM4 Pc;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) Pc.m[a][b] = a == b ? 1.0f : 0.0f;
#pragma unroll
    for (int l = LSEG - 1; l >= 0; --l) {
      const unsigned c = cell(l);
      const bool act = l < nl;
      const TsxH4 F = mat(2, pr[l]);
      float ru[4], rd[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (RQ == 2) {
          const unsigned w = tsx_ldo(rb, (size_t)a * Nc, c);
          ru[a] = __uint_as_float(w << 16);
          rd[a] = __uint_as_float(w & 0xffff0000u);
        } else {
          ru[a] = tsx_ldo(r, (size_t)(2 * a) * Nc, c);
          rd[a] = tsx_ldo(r, (size_t)(2 * a + 1) * Nc, c);
          if (RQ == 1 && live && act) tsx_sto(rb, (size_t)a * Nc, c, tsx_bf16x2(ru[a], rd[a]));
        }
      }
      if (GS) {
        nbr_load(halo, level(l), c, nb[l]);
        float zx[4], zy[4];
        nbr_vals(nb[l], zx, zy);
        if (TSX_PCS_C16) {  // fp16: two top dsts per record; the y sources first, then the x sources (four records live at a time)
#pragma unroll
          for (int ax = 0; ax < 2; ++ax) {
            const float(&zz)[4] = ax == 0 ? zy : zx;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const tsx_h8 hh = __builtin_bit_cast(tsx_h8, brec(4 * ax + m, c, eid[l]));
#pragma unroll
              for (int q = 0; q < 4; ++q) {  // top dst t = 2 m (not inward: up), t = 2 m + 1 (down)
                ru[m] = __builtin_fmaf((float)hh[q], zz[q], ru[m]);
                rd[m] = __builtin_fmaf((float)hh[4 + q], zz[q], rd[m]);
              }
            }
          }
        } else {
        const uint4 cy0 = brec(0, c, eid[l]), cy1 = brec(1, c, eid[l]), cx0 = brec(2, c, eid[l]), cx1 = brec(3, c, eid[l]);
        const unsigned wy[8] = {cy0.x, cy0.y, cy0.z, cy0.w, cy1.x, cy1.y, cy1.z, cy1.w};
        const unsigned wx[8] = {cx0.x, cx0.y, cx0.z, cx0.w, cx1.x, cx1.y, cx1.z, cx1.w};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          float cq[4], cp[4];
          tsx_fp8x4(wy[t], cq);
          tsx_fp8x4(wx[t], cp);
          const float s8 = cq[0] * zy[0] + cq[1] * zy[1] + cq[2] * zy[2] + cq[3] * zy[3] + cp[0] * zx[0] + cp[1] * zx[1] +
                           cp[2] * zx[2] + cp[3] * zx[3];
          if (t & 1) rd[t >> 1] += s8 * (1.0f / TSX_FP8_SCALE);
          else ru[t >> 1] += s8 * (1.0f / TSX_FP8_SCALE);
        }
        }
      }
      float Fr[4];
      tsx_mv4(F, rd, Fr);
      const TsxH4 E = mat(0, pr[l]);
      if (TSX_PCSH_KEEP_E) Ekeep[l] = E;
      float EB[4];
      tsx_mv4(E, Bl, EB);
      const TsxM4 EP = tsx_mm4(E, Pc);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        beta[l][a] = act ? ru[a] + Fr[a] : 0.0f;
        rdg[l][a] = rd[a];
        Bl[a] = act ? beta[l][a] + EB[a] : Bl[a];
#pragma unroll
        for (int b = 0; b < 4; ++b) Pc.m[a][b] = act ? EP.m[a][b] : Pc.m[a][b];
      }
    }
    put_summary(Bl, Pc);
  };
  if (anyface) phase1(std::true_type{});
  else phase1(std::false_type{});
  __syncthreads();
  float Bin[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) Bin[a] = tsx_ldu(rt + (size_t)(2 * a) * ncol, col);  // B_Nz = ru_Nz
  for (int s2 = NSEG - 1; s2 > sg; --s2) chain(s2, Bin);
  // ---- phase 2: the true B of every level (re-run with the true inflow)
  float Bk[LSEG][4];
  {
    float Bc[4] = {Bin[0], Bin[1], Bin[2], Bin[3]};
#pragma unroll
    for (int l = LSEG - 1; l >= 0; --l) {
      const TsxH4 E = TSX_PCSH_KEEP_E ? Ekeep[l] : mat(0, pr[l]);
      float EB[4];
      tsx_mv4(E, Bc, EB);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        Bc[a] = l < nl ? beta[l][a] + EB[a] : Bc[a];
        Bk[l][a] = Bc[a];
      }
    }
  }
  __syncthreads();  // everybody has read the upward summaries: the buffer is free for the downward ones
  // ---- phase 3: local downward scan with zero inflow; keeps gamma and GT of its levels
  float gam[LSEG][4];
  TsxH4 GTkeep[TSX_PCSH_KEEP_GT ? LSEG : 1];
  {
    float Vl[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    TsxM4 Qc;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) Qc.m[a][b] = a == b ? 1.0f : 0.0f;
#pragma unroll
    for (int l = 0; l < LSEG; ++l) {
      const bool act = l < nl;
      const TsxH4 G = mat(4, pr[l]), Hm = mat(6, pr[l]);  // (G stored minus the identity)
      const TsxH4 GT = mat(8, pr[l]);
      if (TSX_PCSH_KEEP_GT) GTkeep[l] = GT;
      float Bn[4], Gr[4], HB[4], GV[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) Bn[a] = l + 1 < LSEG ? Bk[l + 1 < LSEG ? l + 1 : l][a] : Bin[a];
      tsx_mv4<true>(G, rdg[l], Gr);
      tsx_mv4(Hm, Bn, HB);
      tsx_mv4(GT, Vl, GV);
      const TsxM4 GQ = tsx_mm4(GT, Qc);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        gam[l][a] = act ? Gr[a] + HB[a] : 0.0f;
        Vl[a] = act ? gam[l][a] + GV[a] : Vl[a];
#pragma unroll
        for (int b = 0; b < 4; ++b) Qc.m[a][b] = act ? GQ.m[a][b] : Qc.m[a][b];
      }
    }
    put_summary(Vl, Qc);
  }
  __syncthreads();
  float V[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) V[a] = tsx_ldu(rt + (size_t)(2 * a + 1) * ncol, col);  // V_0 = rd_TOA
  if (sg == 0) {  // tail rows: the TOA identity rows and the side dummies at level Nz
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (MODE == 1 && live) zt[(size_t)(2 * a + 1) * ncol + col] = V[a];
      if (FINAL) wpair_if(live, zfin + (size_t)D * Nc + (size_t)(2 * a + 1) * ncol, ncp, V[a], zt[(size_t)(2 * a + 1) * ncol + (size_t)((int)col + oc)]);
    }
#pragma unroll
    for (int d = NTOP; d < D; ++d) {
      const float v = rt[(size_t)d * ncol + col];
      if (MODE == 1 && live) zt[(size_t)d * ncol + col] = v;
      if (FINAL) wpair_if(live, zfin + (size_t)D * Nc + (size_t)d * ncol, ncp, v, zt[(size_t)d * ncol + (size_t)((int)col + oc)]);
    }
  }
  for (int s2 = 0; s2 < sg; ++s2) chain(s2, V);
  // ---- phase 4: true V, U; side streams; stores
  unsigned zw[(MODE == 0 && TSX_PCSH_DEFER_ST) ? LSEG : 1][4];
#pragma unroll
  for (int l = 0; l < LSEG; ++l) {
    const bool st = live && l < nl;
    const unsigned c = cell(l);
    const unsigned cn = (unsigned)level(l) * ncol + ncp;
    const TsxH4 GT = TSX_PCSH_KEEP_GT ? GTkeep[l] : mat(8, pr[l]), An = mat(10, pr[l]);
    float Bn[4], GV[4], Vn[4], AV[4], Un[4], U[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) Bn[a] = l + 1 < LSEG ? Bk[l + 1 < LSEG ? l + 1 : l][a] : Bin[a];
    tsx_mv4(GT, V, GV);
#pragma unroll
    for (int a = 0; a < 4; ++a) Vn[a] = gam[l][a] + GV[a];
    tsx_mv4(An, Vn, AV);
#pragma unroll
    for (int a = 0; a < 4; ++a) Un[a] = AV[a] + Bn[a];
    if (MODE != 0) {
      const TsxH4 Ao = mat(12, pr[l]);
      float AoV[4];
      tsx_mv4(Ao, V, AoV);
#pragma unroll
      for (int a = 0; a < 4; ++a) U[a] = AoV[a] + Bk[l][a];
    }
    float pt[8];
    float2 ps[4];
    if (FINAL) {
#pragma unroll
      for (int q = 0; q < 8; ++q) pt[q] = tsx_ldo(z, (size_t)q * Nc, (unsigned)((int)c + oc));
#pragma unroll
      for (int m2 = 0; m2 < 4; ++m2) ps[m2] = tsx_ldo(zr, (size_t)m2 * Nc, (unsigned)((int)c + oc));
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (MODE == 1 && st) {
        tsx_sto(z, (size_t)(2 * a) * Nc, c, U[a]);
        tsx_sto(z, (size_t)(2 * a + 1) * Nc, c, Vn[a]);
      }
      if (FINAL) {
        wpair_if(st, zfin + (size_t)(2 * a) * Nc, cn, U[a], pt[2 * a]);
        wpair_if(st, zfin + (size_t)(2 * a + 1) * Nc, cn, Vn[a], pt[2 * a + 1]);
      }
    }
    float zx[4], zy[4];
    uint4 sy[2], sx[2];
    if (GS) {
      nbr_vals(nb[l], zx, zy);
      sy[0] = brec(12 + TSX_S16H_BO, c, eid[l]);
      sy[1] = brec(13 + TSX_S16H_BO, c, eid[l]);
      sx[0] = brec(14 + TSX_S16H_BO, c, eid[l]);
      sx[1] = brec(15 + TSX_S16H_BO, c, eid[l]);
    }
    const unsigned uy[8] = {sy[0].x, sy[0].y, sy[0].z, sy[0].w, sy[1].x, sy[1].y, sy[1].z, sy[1].w};
    const unsigned ux[8] = {sx[0].x, sx[0].y, sx[0].z, sx[0].w, sx[1].x, sx[1].y, sx[1].z, sx[1].w};
    float rs[8];
    if (RQ == 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned w = tsx_ldo(rb, (size_t)(4 + q) * Nc, c);
        rs[2 * q] = __uint_as_float(w << 16);
        rs[2 * q + 1] = __uint_as_float(w & 0xffff0000u);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) rs[q] = tsx_ldo(r, (size_t)(NTOP + q) * Nc, c);
      if (RQ == 1 && st) {
#pragma unroll
        for (int q = 0; q < 4; ++q) tsx_sto(rb, (size_t)(4 + q) * Nc, c, tsx_bf16x2(rs[2 * q], rs[2 * q + 1]));
      }
    }
    const tsx_f2 zy01 = {zy[0], zy[1]}, zy23 = {zy[2], zy[3]}, zx01 = {zx[0], zx[1]}, zx23 = {zx[2], zx[3]};
    float zo[8];
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) {
      const tsx_h8 row = __builtin_bit_cast(tsx_h8, brec(4 + TSX_S16H_BO + dd, c, eid[l]));
      float acc = rs[dd];
#pragma unroll
      for (int a = 0; a < 4; ++a) acc = __builtin_fmaf((float)row[2 * a + 1], V[a], __builtin_fmaf((float)row[2 * a], Un[a], acc));
      if (GS) {  // packed fp32 FMAs on the pairs that v_cvt_pk_f32_fp8 delivers, as in tsx_k_pcs_rb
        tsx_f2 a2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)uy[dd], false) * zy01;
        a2 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)uy[dd], true), zy23, a2);
        a2 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)ux[dd], false), zx01, a2);
        a2 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)ux[dd], true), zx23, a2);
        acc = __builtin_fmaf(a2.x + a2.y, 1.0f / TSX_FP8_SCALE, acc);
      }
      zo[dd] = acc;
    }
    // records by consumer: side dofs (8,10) (9,11) (12,14) (13,15) = zo[0,2] zo[1,3] zo[4,6] zo[5,7]
    if (MODE == 0) {
      if (TSX_PCSH_DEFER_ST) {  // stored behind the level loop: no (predicated) store between the levels' loads
        zw[l][0] = tsx_bf16x2(zo[0], zo[2]);
        zw[l][1] = tsx_bf16x2(zo[1], zo[3]);
        zw[l][2] = tsx_bf16x2(zo[4], zo[6]);
        zw[l][3] = tsx_bf16x2(zo[5], zo[7]);
      } else if (st) {
        tsx_sto(zb, (size_t)0 * Nc, c, tsx_bf16x2(zo[0], zo[2]));
        tsx_sto(zb, (size_t)1 * Nc, c, tsx_bf16x2(zo[1], zo[3]));
        tsx_sto(zb, (size_t)2 * Nc, c, tsx_bf16x2(zo[4], zo[6]));
        tsx_sto(zb, (size_t)3 * Nc, c, tsx_bf16x2(zo[5], zo[7]));
      }
    }
    if (MODE == 1 && st) {
      tsx_sto(zr, (size_t)0 * Nc, c, make_float2(zo[0], zo[2]));
      tsx_sto(zr, (size_t)1 * Nc, c, make_float2(zo[1], zo[3]));
      tsx_sto(zr, (size_t)2 * Nc, c, make_float2(zo[4], zo[6]));
      tsx_sto(zr, (size_t)3 * Nc, c, make_float2(zo[5], zo[7]));
    }
    if (FINAL) {
      const int dofs[8] = {8, 10, 9, 11, 12, 14, 13, 15};
      const float mine[8] = {zo[0], zo[2], zo[1], zo[3], zo[4], zo[6], zo[5], zo[7]};
      const float part[8] = {ps[0].x, ps[0].y, ps[1].x, ps[1].y, ps[2].x, ps[2].y, ps[3].x, ps[3].y};
#pragma unroll
      for (int q = 0; q < 8; ++q) wpair_if(st, zfin + (size_t)dofs[q] * Nc, cn, mine[q], part[q]);
    }
    if (k0 + l == Nz - 1) {  // U_Nz = (albedo / streams) sum V_Nz + ru_Nz: the surface rows
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (MODE == 1 && live) zt[(size_t)(2 * a) * ncol + col] = Un[a];
        if (FINAL) wpair_if(live, zfin + (size_t)D * Nc + (size_t)(2 * a) * ncol, ncp, Un[a], zt[(size_t)(2 * a) * ncol + (size_t)((int)col + oc)]);
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) V[a] = l < nl ? Vn[a] : V[a];
  }
  if (MODE == 0 && TSX_PCSH_DEFER_ST) {
#pragma unroll
    for (int l = 0; l < LSEG; ++l) {
      if (!(live && l < nl)) continue;
      const unsigned c = cell(l);
#pragma unroll
      for (int m = 0; m < 4; ++m) tsx_sto(zb, (size_t)m * Nc, c, zw[l][m]);
    }
  }
}
