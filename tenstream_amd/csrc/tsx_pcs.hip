// tsx_pcs.hip -- host side of the segmented-scan red-black preconditioner (kernels: tsx_kernels_pcs.hpp)
#include <stdio.h>
#include <string.h>

#include "tsx_host.hpp"
#include "tsx_peer.hpp"
#include "tsx_peer_dev.hpp"
#include "tsx_kernels_pcs.hpp"

struct PcsCfg {
  int lseg, nseg, cw;
};

// (LSEG, NSEG) pairs that are instantiated; CW in {64, 32, 16}
static const int k_pairs[][2] = {{4, 16}, {8, 8}, {8, 16}, {16, 16}, {2, 32}};  // (2, 32): 8_16 only, A/B (round 6)

static PcsCfg pcs_config(const tsx_solver *s) {
  const TsxGeo &g = s->geo;
  PcsCfg c = {0, 0, 0};
  // the scan kernels address their planes through buffer descriptors: 32-bit lane offsets of 16-byte records (tsx_ldu) and
  // 32-bit scalar offsets of up to fifteen 4-byte planes / three 16-byte planes (tsx_ldo)
  if (g.Nc >= (1ll << 26)) return c;
  int e_l = 0, e_s = 0, e_cw = 0;  // TSX_PCS_CFG=lseg,nseg,cw (A/B knob; read per call so that tests can switch it)
  if (const char *e = getenv("TSX_PCS_CFG")) sscanf(e, "%d,%d,%d", &e_l, &e_s, &e_cw);
  if (e_l > 0 && e_l * e_s >= g.Nz) {
    for (const auto &p : k_pairs)
      if (p[0] == e_l && p[1] == e_s) {
        c.lseg = e_l;
        c.nseg = e_s;
      }
  }
  const long long nthr = (long long)g.ym * (g.xm / 2);  // columns per pass
  if (g.ntop == 8) {  // 8_16: 4 x 4 blocks per level -- registers allow 4 levels per thread (8 for deep columns)
    if (!c.lseg) {
      if (g.Nz <= 64) c.lseg = 4, c.nseg = 16;
      else if (g.Nz <= 128) c.lseg = 8, c.nseg = 16;
    }
    if (c.lseg == 16 || (c.lseg == 8 && c.nseg == 8)) c.lseg = c.nseg = 0;  // not instantiated for 8_16
    if (!c.lseg) return c;
    // two levels per thread (TSX_PCS_CFG=2,32,8|16): half the per-level state in registers -- three workgroups of 8 columns x 32
    // segments per CU instead of one of 32 x 16 (round 6 A/B, profiles/r06)
    if (c.lseg == 2) c.cw = e_cw == 16 ? 16 : 8;
    else c.cw = (e_cw == 32 || e_cw == 16) ? e_cw : (nthr >= 8192 ? 32 : 16);
    return c;
  }
  if (c.lseg == 2) c.lseg = c.nseg = 0;  // 3_10: not instantiated
  if (!c.lseg) {
    // measured (scripts/pcsbench.py): 8 levels x 8 segments on large passes (>= 16 K columns: fewer, fatter threads),
    // 4 x 16 on small ones (more waves); deeper columns take the smallest pair that holds them
    if (g.Nz <= 64) {
      const bool fat = nthr >= 16384 && !(s->dd_on || s->dd_pc);  // with shared blocks a pass moves half the bytes: more waves win again
      c.lseg = fat ? 8 : 4;
      c.nseg = fat ? 8 : 16;
    } else if (g.Nz <= 128) {
      c.lseg = 8;
      c.nseg = 16;
    } else if (g.Nz <= 256) {
      c.lseg = c.nseg = 16;
    }
  }
  if (!c.lseg) return c;  // Nz > 256: not eligible
  // 32 columns per workgroup measured best on 128^2 and 256^2 columns; 16 keeps >= 256 workgroups on small domains
  c.cw = (e_cw == 64 || e_cw == 32 || e_cw == 16) ? e_cw : (nthr >= 8192 && c.lseg < 16 ? 32 : 16);
  return c;
}

// side -> top couplings (record 1) as two fp16 records instead of one fp8 record (tsx_kernels_pcs.hpp "C16"): a build-time
// choice (both variants of every pass kernel would double the compile time); measured 6 -> 5 iterations on the metric domain
#ifndef TSX_PCS_C16
#define TSX_PCS_C16 1
#endif
static constexpr bool pcs_c16() { return TSX_PCS_C16 != 0; }

bool tsx_pcs_eligible(const tsx_solver *s) {
  const char *e = getenv("TSX_PC_SCAN");  // TSX_PC_SCAN=0: the one-lane-per-column kernels (A/B knob)
  const int on = e ? atoi(e) : 1;
  return on && pcs_config(s).lseg > 0;
}

static int pcsh_pack(tsx_solver *s) {  // 8_16: 14 matrix records per cell, then 16 / 20 block records per cell or per entry
  const TsxGeo &g = s->geo;
  s->coef_h_c16 = pcs_c16();
  uint4 *P = (uint4 *)s->coef_h, *PB = P + (size_t)TSX_S16H_CELL * g.Nc;
  s->coef_h_dd = false;
  const int nbc = (g.ncol + 63) / 64;
  if (s->coef_bytes == 4) {
    hipLaunchKernelGGL((tsx_k_pcsh_pack_col<float>), dim3(nbc), dim3(64), 0, s->stream, g, (const float *)s->coef, s->l1d, s->a11,
                       s->a12, s->albedo, P);
    if (s->dd_on || s->dd_pc) {
      hipLaunchKernelGGL((tsx_k_pcsh_pack_block<float>), dim3(grid_for((long long)TSX_S16H_BLOCK * s->pc_nent)), dim3(TSX_BLOCK), 0,
                         s->stream, g, (long long)s->pc_nent, (const float *)s->pc_coef, (const int *)s->pc_ent_cell, s->l1d, PB);
      s->coef_h_dd = true;
      // where blocks repeat, so do the column recurrences below the lowest cloud of a column: share the 14 records too
      int rc = tsx_records_share(s, TSX_S16H_CELL, P);
      if (rc) return rc;
    } else {
      hipLaunchKernelGGL((tsx_k_pcsh_pack_block<float>), dim3(grid_for((long long)TSX_S16H_BLOCK * g.Nc)), dim3(TSX_BLOCK), 0,
                         s->stream, g, g.Nc, (const float *)s->coef, (const int *)nullptr, s->l1d, PB);
    }
  } else {
    hipLaunchKernelGGL((tsx_k_pcsh_pack_col<double>), dim3(nbc), dim3(64), 0, s->stream, g, (const double *)s->coef, s->l1d,
                       s->a11, s->a12, s->albedo, P);
    hipLaunchKernelGGL((tsx_k_pcsh_pack_block<double>), dim3(grid_for((long long)TSX_S16H_BLOCK * g.Nc)), dim3(TSX_BLOCK), 0,
                       s->stream, g, g.Nc, (const double *)s->coef, (const int *)nullptr, s->l1d, PB);
  }
  HIPCHK(hipGetLastError());
  return TSX_OK;
}

int tsx_pcs_pack(tsx_solver *s) {
  const TsxGeo &g = s->geo;
  s->pcr_on = false;
  if (g.ntop == 8) return pcsh_pack(s);
  uint4 *P = (uint4 *)s->coef_h;
  s->coef_h_dd = false;
  if ((s->dd_on || s->dd_pc) && s->coef_bytes == 4) {
    // group 0 per cell, groups 1..7 per distinct block behind it (7 * nent <= 7 * Nc records: the same buffer holds them)
    if (s->dd_on && !s->coef_dense_valid)  // the LUT path left the blocks in the shared storage only: read them through the index
      hipLaunchKernelGGL((tsx_k_pcs_pack_col<float>), dim3((g.ncol + 63) / 64), dim3(64), 0, s->stream, g, (const float *)s->dd_coef,
                         s->l1d, s->a11, s->a12, s->albedo, P, (const int *)s->dd_cidx, (long long)s->dd_nent);
    else
      hipLaunchKernelGGL((tsx_k_pcs_pack_col<float>), dim3((g.ncol + 63) / 64), dim3(64), 0, s->stream, g, (const float *)s->coef,
                         s->l1d, s->a11, s->a12, s->albedo, P, (const int *)nullptr, 0ll);
    s->coef_h_c16 = pcs_c16();
    // near-identical grouping (dd_pc without dd_on): a wave's ids are unrelated -> an entry's records in one line (TSX_PC_ENTRY_MAJOR=0 / 1 overrides)
    {
      const char *e = getenv("TSX_PC_ENTRY_MAJOR");
      s->pe_entry_major = pcs_c16() && (e ? atoi(e) != 0 : (s->dd_pc && !s->dd_on));
    }
    if (s->coef_h_c16)  // side -> top couplings in fp16: 8 records per distinct block
      hipLaunchKernelGGL(tsx_k_pcs_pack_ent16, dim3(grid_for((long long)TSX_PCS_ENT16_SLOTS * s->pc_nent)), dim3(TSX_BLOCK), 0,
                         s->stream, g.ncol, (long long)s->pc_nent, (const float *)s->pc_coef, (const int *)s->pc_ent_cell, s->l1d,
                         P + g.Nc, s->pe_entry_major ? 1 : 0);
    else
      hipLaunchKernelGGL(tsx_k_pcs_pack_ent, dim3(grid_for(7ll * s->pc_nent)), dim3(TSX_BLOCK), 0, s->stream, g.ncol,
                         (long long)s->pc_nent, (const float *)s->pc_coef, (const int *)s->pc_ent_cell, s->l1d, P + g.Nc);
    // the intermediate passes' copy of record 0 (block index in the word of A_k) in the last group's slot: the entries
    // fill at most 3.5 of the 7 groups behind record 0 (sharing is on only where 2 * nent <= Nc)
    hipLaunchKernelGGL(tsx_k_pcs_pack_r0g, dim3(grid_for(g.Nc)), dim3(TSX_BLOCK), 0, s->stream, (long long)g.Nc, (const uint4 *)P,
                       (const int *)s->pc_cidx_split, P + (size_t)7 * g.Nc);
    HIPCHK(hipGetLastError());
    s->coef_h_dd = true;
    // below the lowest cloud of a column -- and in every clear column -- that record is the same for all columns of a level.
    // It costs one more dependent load per level (index -> record -> block records): worth it where a pass is bound by bytes
    // (512 x 256 columns: 85.8 -> 78.0 us, 256 x 256: 35.2 -> 32.8 us), not where it is bound by latency (128 x 128: 12.0 -> 12.5 us)
    if ((long long)g.ym * (g.xm / 2) < 16384) return TSX_OK;
    return tsx_records_share(s, 1, P + (size_t)7 * g.Nc);
  }
  s->coef_h_c16 = pcs_c16();
  if (s->coef_bytes == 4) {
    hipLaunchKernelGGL((tsx_k_pcs_pack_col<float>), dim3((g.ncol + 63) / 64), dim3(64), 0, s->stream, g, (const float *)s->coef,
                       s->l1d, s->a11, s->a12, s->albedo, P);
    hipLaunchKernelGGL((tsx_k_pcs_pack<float>), dim3(grid_for(7 * g.Nc)), dim3(TSX_BLOCK), 0, s->stream, g, (const float *)s->coef,
                       s->l1d, P);
    if (s->coef_h_c16)
      hipLaunchKernelGGL((tsx_k_pcs_pack_rec1h<float>), dim3(grid_for(g.Nc)), dim3(TSX_BLOCK), 0, s->stream, g,
                         (const float *)s->coef, s->l1d, P);
  } else {
    hipLaunchKernelGGL((tsx_k_pcs_pack_col<double>), dim3((g.ncol + 63) / 64), dim3(64), 0, s->stream, g, (const double *)s->coef,
                       s->l1d, s->a11, s->a12, s->albedo, P);
    hipLaunchKernelGGL((tsx_k_pcs_pack<double>), dim3(grid_for(7 * g.Nc)), dim3(TSX_BLOCK), 0, s->stream, g,
                       (const double *)s->coef, s->l1d, P);
    if (s->coef_h_c16)
      hipLaunchKernelGGL((tsx_k_pcs_pack_rec1h<double>), dim3(grid_for(g.Nc)), dim3(TSX_BLOCK), 0, s->stream, g,
                         (const double *)s->coef, s->l1d, P);
  }
  HIPCHK(hipGetLastError());
  return TSX_OK;
}

static int pcs_inplace_env() {
  // TSX_PEER_INPLACE (peer transport): 0 pack + tsx_k_peer_send + tsx_k_peer_recv like any exchange; 1 fused pack + send, the
  // next pass reads in place; 2 (default) the pass stores its boundary records into the neighbours' mailboxes itself and the next
  // pass reads them in place (tsx_k_pcs_rb<..., PEER>; 1 where no such kernel is instantiated).  Measured and dropped (one rank
  // with itself as its neighbours, profiles/NEGATIVE_RESULTS.md): pack, send and receive in ONE kernel with the pass reading its
  // cached receive buffers (no faster than three kernels).
  const char *e = getenv("TSX_PEER_INPLACE");  // read per call: tests switch it
  return e ? atoi(e) : 2;
}
static bool pcs_peer_inkernel(const tsx_solver *s) {
  if (pcs_inplace_env() != 2 || !tsx_peer_ready(s) || s->geo.ntop != 2) return false;
  const PcsCfg c = pcs_config(s);
  return c.lseg == 4 && c.nseg == 16 && (c.cw == 16 || c.cw == 32);
}

// ---- preconditioner halo on several ranks (tsx_k_pcs_halo_pack): on when some face of the rank is a real neighbour
// (or force_halo), both local extents are even and TSX_PC_HALO != 0 -- on EVERY rank.  The exchange is a matched send / recv
// with the four neighbours, so one rank deciding differently (an odd local extent from an uneven split such as
// xs = (xi * Nx) / nxp with Nx = 7, or zebra rows after the local red-black fallback) would leave its neighbours' messages
// unmatched or paired with the operator's halo messages.  pcs_halo_local is this rank's vote, tsx_pc_global_agree
// (called from prepare_ksp / tsx_diff_pc_apply, collective) the all-reduce.  All extents even on all ranks also means
// xs and ys are even everywhere (sums of even extents), so the local colour (i + j) & 1 is the global colour.
static bool pcs_halo_local(const tsx_solver *s) {
  const TsxGeo &g = s->geo;
  const char *e = getenv("TSX_PC_HALO");
  if (e && atoi(e) == 0) return false;
  return !(g.wrap_x && g.wrap_y) && g.xm % 2 == 0 && g.ym % 2 == 0 && g.pc_tile_x == 0 && g.pc_tile_y == 0;
}
static bool pcs_halo_on(const tsx_solver *s) {
  if (!pcs_halo_local(s)) return false;
  return s->grid.nranks <= 1 || s->pcg_halo_ok;  // one rank with force_halo: its own vote is everybody's
}
int tsx_pc_global_agree(tsx_solver *s) {
  if (s->grid.nranks <= 1) return TSX_OK;
  const int key = (s->pc << 4) | (s->mixed ? 2 : 0) | (s->pc_split ? 1 : 0);
  if (key == s->pcg_key) return TSX_OK;
  const bool mine = s->pc == TSX_PC_REDBLACK && s->pc_split && tsx_pcs_eligible(s) && pcs_halo_local(s) && s->grid.xs % 2 == 0 &&
                    s->grid.ys % 2 == 0;
  // ... and the intermediate passes as one launch with the faces inside it (tsx_k_pcs_flow FPEER): its messages need no
  // acknowledgements, a neighbour that runs a launch per pass waits for them -- so everywhere or nowhere, too
  bool mine_flow = false;
  if (mine) {
    const PcsCfg c = pcs_config(s);
    const int fcw = (!getenv("TSX_PCS_CFG") && c.lseg == 4 && c.nseg == 16)
                        ? (((long long)s->geo.ym * (s->geo.xm / 2) >= 4096 && (s->geo.xm / 2) % 32 == 0) ? 32 : 16)
                        : c.cw;
    mine_flow = pcs_peer_inkernel(s) && tsx_pcs_flow_ok(s, c.lseg, c.nseg, fcw, true);
  }
  double v[2] = {mine ? 0.0 : 1.0, mine_flow ? 0.0 : 1.0};  // number of ranks that cannot take part
  int rc = tsx_allreduce_host(s, v, 2);
  if (rc) return rc;
  s->pcg_halo_ok = v[0] == 0.0;
  s->pcg_flow_ok = v[0] == 0.0 && v[1] == 0.0;
  s->pcg_key = key;
  return TSX_OK;
}
static size_t pcs_halo_doubles(const tsx_solver *s, int q) {  // message length in doubles (the exchange's unit), rounded up
  const TsxGeo &g = s->geo;
  const size_t n = (size_t)tsx_pcs_halo_nzp(g.Nz) * (q < 2 ? g.ym : g.xm);
  return (n + 1) / 2;
}
static int pcs_halo_buffers(tsx_solver *s) {
  for (int q = 0; q < 4; ++q) {
    const size_t bytes = (pcs_halo_doubles(s, q) * sizeof(double) + 15) & ~(size_t)15;  // tsx_k_pcs_halo_xchg copies 16-byte pieces
    if (!s->pch_send[q]) {
      HIPCHK(tsx_dev_malloc((void **)&s->pch_send[q], bytes));
      HIPCHK(hipMemsetAsync(s->pch_send[q], 0, bytes, s->stream));
    }
    if (!s->pch_recv[q]) {
      HIPCHK(tsx_dev_malloc((void **)&s->pch_recv[q], bytes));
      HIPCHK(hipMemsetAsync(s->pch_recv[q], 0, bytes, s->stream));
    }
  }
  return TSX_OK;
}
static TsxPcHalo pcs_halo_arg(const tsx_solver *s) {
  TsxPcHalo h;
  memset((void *)&h, 0, sizeof(h));
  if (!pcs_halo_on(s) || !s->pch_recv[0]) return h;
  const TsxGeo &g = s->geo;
  if (s->pch_inplace && tsx_peer_ready(s)) {  // the pass reads the neighbours' records where they land (pcs_halo_exchange)
    h.W = (const unsigned *)s->pch_slot[0];
    h.E = (const unsigned *)s->pch_slot[1];
    h.S = (const unsigned *)s->pch_slot[2];
    h.N = (const unsigned *)s->pch_slot[3];
    h.wait = s->pch_wait;
    return h;
  }
  if (!g.wrap_x) {
    h.W = s->pch_recv[0];
    h.E = s->pch_recv[1];
  }
  if (!g.wrap_y) {
    h.S = s->pch_recv[2];
    h.N = s->pch_recv[3];
  }
  return h;
}
static long long pcs_nframe(const TsxGeo &g) {  // host mirror of tsx_pcs_nframe
  const int h = g.xm >> 1;
  const int rows = g.wrap_y ? 0 : (g.ym >= 2 ? 2 : 1);
  return (long long)rows * h + (g.wrap_x ? 0 : g.ym - rows);
}
// after a pass: pack the boundary records (on the solver stream) and exchange them with the W, E, S, N neighbours -- on the
// solver stream, or (overlap) on comm_stream behind ev_pack, with ev_recv recorded after it
static int pcs_halo_exchange(tsx_solver *s, bool from_f32, const int *done, bool overlap) {
  const TsxGeo &g = s->geo;
  float *zs = (float *)s->vw;
  const unsigned *zb = (const unsigned *)(zs + (size_t)g.N);
  const float2 *zr = reinterpret_cast<const float2 *>(zs + (size_t)g.ntop * g.Nc);
  const int nzp = tsx_pcs_halo_nzp(g.Nz);
  const long long n = (g.wrap_x ? 0 : (long long)nzp * g.ym) + (g.wrap_y ? 0 : (long long)nzp * g.xm);
  // Peer transport, where the pass cannot send its records itself (8_16, non-default scan configurations; TSX_PEER_INPLACE=1): ONE
  // kernel packs the boundary records straight into the neighbours' mailboxes and publishes them; the next pass waits for its
  // neighbours' sequence numbers itself and reads the records in place -- no send buffer, no receive kernel, no copy out.
  // TSX_PEER_INPLACE=0: pack, tsx_k_peer_send, tsx_k_peer_recv as for any other exchange.
  s->pch_inplace = false;
  if (pcs_inplace_env() != 0 && !overlap && tsx_peer_ready(s) && n > 0) {
    const size_t bx = g.wrap_x ? 0 : (size_t)nzp * g.ym * sizeof(unsigned), by = g.wrap_y ? 0 : (size_t)nzp * g.xm * sizeof(unsigned);
    const size_t bytes[4] = {bx, bx, by, by};
    int rc;
    TsxPeerXArgs a;
    if ((rc = tsx_peer_prepare_send(s, bytes, &a))) return rc;
    int nblk = (int)((n + 4 * TSX_BLOCK - 1) / (4 * TSX_BLOCK));
    nblk = nblk < 1 ? 1 : (nblk > 16 ? 16 : nblk);
    hipLaunchKernelGGL(tsx_k_pcs_halo_send, dim3(nblk), dim3(TSX_BLOCK), 0, s->stream, g, zb, zr, from_f32 ? 1 : 0, a);
    HIPCHK(hipGetLastError());
    if ((rc = tsx_peer_expect(s, bytes, &s->pch_wait, s->pch_slot))) return rc;
    s->pch_inplace = true;
    return TSX_OK;
  }
  hipLaunchKernelGGL(tsx_k_pcs_halo_pack, dim3(grid_for(n)), dim3(TSX_BLOCK), 0, s->stream, g, zb, zr, from_f32 ? 1 : 0,
                     s->pch_send[0], s->pch_send[1], s->pch_send[2], s->pch_send[3], done);
  HIPCHK(hipGetLastError());
  double *const send[4] = {(double *)s->pch_send[0], (double *)s->pch_send[1], (double *)s->pch_send[2], (double *)s->pch_send[3]};
  double *const recv[4] = {(double *)s->pch_recv[0], (double *)s->pch_recv[1], (double *)s->pch_recv[2], (double *)s->pch_recv[3]};
  if (!overlap) return tsx_face_exchange_bufs(s, s->stream, send, recv, pcs_halo_doubles(s, 0), pcs_halo_doubles(s, 2));
  HIPCHK(hipEventRecord(s->ev_pack, s->stream));
  HIPCHK(hipStreamWaitEvent(s->comm_stream, s->ev_pack, 0));
  return TSX_OK;  // the caller queues the next pass's interior part first, then calls pcs_halo_exchange_finish
}
static int pcs_halo_exchange_finish(tsx_solver *s) {
  double *const send[4] = {(double *)s->pch_send[0], (double *)s->pch_send[1], (double *)s->pch_send[2], (double *)s->pch_send[3]};
  double *const recv[4] = {(double *)s->pch_recv[0], (double *)s->pch_recv[1], (double *)s->pch_recv[2], (double *)s->pch_recv[3]};
  int rc = tsx_face_exchange_bufs(s, s->comm_stream, send, recv, pcs_halo_doubles(s, 0), pcs_halo_doubles(s, 2));
  if (rc) return rc;
  HIPCHK(hipEventRecord(s->ev_recv, s->comm_stream));
  HIPCHK(hipStreamWaitEvent(s->stream, s->ev_recv, 0));
  return TSX_OK;
}

// rq: 0 = fp32 right-hand side, 1 = fp32 + leave the bf16-pair words, 2 = read the bf16-pair words (mode 0 only)
template <int L, int S, int CW>
static void pcs_launch(tsx_solver *s, bool gs, int mode, int rbc, int nonbr, float *zs, unsigned *zb, float *zfin, const int *done,
                       int rq, int part) {
  const TsxGeo &g = s->geo;
  const long long nthr = part == 2 ? pcs_nframe(g) : (long long)g.ym * (g.xm / 2);
  const int nb = (int)((nthr + CW - 1) / CW);
  const uint4 *P = (const uint4 *)s->coef_h;
  const float *r = (const float *)s->pc_rhs;
  const bool dd = s->coef_h_dd;
  const int *cidx = (const int *)s->pc_cidx_split;
  const long long nent = s->pc_nent;
  const uint4 *PE = P + g.Nc;
  const TsxPcHalo hal = pcs_halo_arg(s);
  unsigned *rb = zb + (size_t)4 * g.Nc;  // behind the iterate's bf16 records in s->vw
  const int *pidx = dd && s->pcr_on ? (const int *)s->pcr_idx : (const int *)nullptr;  // shared record 0 of the intermediate passes
  const uint4 *PT = (const uint4 *)s->pcr_tab;
  constexpr bool C16 = pcs_c16();
  // the pass stores its boundary records into the neighbours' mailboxes itself (tsx_pcs_apply decided; kernels instantiated
  // for the default configurations only, pcs_peer_kernel_ok)
  constexpr bool PEEROK = L == 4 && S == 16 && (CW == 16 || CW == 32);
  const bool peer = PEEROK && s->pch_snd_on && part == 0;
  TsxPeerXArgs snd;
  if (peer) snd = s->pch_snd;
  else memset((void *)&snd, 0, sizeof(snd));
#define TSX_PCS_GO1(GSV, MODEV, RQV, IDXV, PEERV)                                                                                \
  hipLaunchKernelGGL((tsx_k_pcs_rb<L, S, CW, GSV, MODEV, IDXV, RQV, C16, PEERV>), dim3(nb), dim3(CW *S), 0, s->stream, g, P, r,  \
                     zs, zb, zfin, done, rbc, nonbr, IDXV ? cidx : (const int *)nullptr, IDXV ? nent : 0ll,                      \
                     IDXV ? PE : (const uint4 *)nullptr, hal, rb, part, IDXV ? pidx : (const int *)nullptr,                      \
                     IDXV ? PT : (const uint4 *)nullptr, snd, (IDXV && s->pe_entry_major) ? TSX_PCS_ENT16_SLOTS : 1)
#define TSX_PCS_GO(GSV, MODEV, RQV)                                                                                              \
  do {                                                                                                                           \
    if constexpr (PEEROK && (MODEV) != 2) {                                                                                      \
      if (peer) {                                                                                                                \
        if (dd) TSX_PCS_GO1(GSV, MODEV, RQV, true, true);                                                                        \
        else TSX_PCS_GO1(GSV, MODEV, RQV, false, true);                                                                          \
        break;                                                                                                                   \
      }                                                                                                                          \
    }                                                                                                                            \
    if (dd) TSX_PCS_GO1(GSV, MODEV, RQV, true, false);                                                                           \
    else TSX_PCS_GO1(GSV, MODEV, RQV, false, false);                                                                             \
  } while (0)
  if (!gs) {
    if (rq == 1) TSX_PCS_GO(false, 0, 1);
    else if (rq == 2) TSX_PCS_GO(false, 0, 2);
    else TSX_PCS_GO(false, 0, 0);
  } else if (mode == 0) {
    if (rq == 2) TSX_PCS_GO(true, 0, 2);
    else if (rq == 1) TSX_PCS_GO(true, 0, 1);
    else TSX_PCS_GO(true, 0, 0);
  } else if (mode == 1) TSX_PCS_GO(true, 1, 0);
  else TSX_PCS_GO(true, 2, 0);  // (the two fp32 passes on the bf16 words too: measured, no gain -- 14.41 vs 14.48 ms)
#undef TSX_PCS_GO
#undef TSX_PCS_GO1
}

template <int L, int S>
static void pcs_launch_cw(tsx_solver *s, int cw, bool gs, int mode, int rbc, int nonbr, float *zs, unsigned *zb, float *zfin,
                          const int *done, int rq, int part) {
  if (cw == 64) pcs_launch<L, S, 64>(s, gs, mode, rbc, nonbr, zs, zb, zfin, done, rq, part);
  else if (cw == 32) pcs_launch<L, S, 32>(s, gs, mode, rbc, nonbr, zs, zb, zfin, done, rq, part);
  else pcs_launch<L, S, 16>(s, gs, mode, rbc, nonbr, zs, zb, zfin, done, rq, part);
}

template <int L, int S, int CW>
static void pcsh_launch(tsx_solver *s, bool gs, int mode, int rbc, int nonbr, float *zs, unsigned *zb, float *zfin, const int *done,
                        int rq, int part) {
  const TsxGeo &g = s->geo;
  const long long nthr = part == 2 ? pcs_nframe(g) : (long long)g.ym * (g.xm / 2);
  const int nb = (int)((nthr + CW - 1) / CW);
  const uint4 *Pcell = (const uint4 *)s->coef_h, *PB = Pcell + (size_t)TSX_S16H_CELL * g.Nc;
  // recurrence records: per cell, or the shared table behind the per-cell index (tsx_records_share)
  const uint4 *P = s->pcr_on ? (const uint4 *)s->pcr_tab : Pcell;
  const int *pidx = s->pcr_on ? (const int *)s->pcr_idx : (const int *)nullptr;
  const long long pstride = s->pcr_on ? s->pcr_n : (long long)g.Nc;
  const float *r = (const float *)s->pc_rhs;
  const bool dd = s->coef_h_dd;
  const long long bstride = dd ? (long long)s->pc_nent : g.Nc;
  const int *cidx = (const int *)s->pc_cidx_split;
  const TsxPcHalo hal = pcs_halo_arg(s);
  unsigned *rb = zb + (size_t)4 * g.Nc;
#define TSX_PCSH_GO(GSV, MODEV, RQV)                                                                                            \
  do {                                                                                                                          \
    if (dd)                                                                                                                     \
      hipLaunchKernelGGL((tsx_k_pcsh_rb<L, S, CW, GSV, MODEV, true, RQV>), dim3(nb), dim3(CW *S), 0, s->stream, g, P, PB,        \
                         bstride, cidx, r, zs, zb, zfin, done, rbc, nonbr, hal, rb, part, pidx, pstride);                       \
    else                                                                                                                        \
      hipLaunchKernelGGL((tsx_k_pcsh_rb<L, S, CW, GSV, MODEV, false, RQV>), dim3(nb), dim3(CW *S), 0, s->stream, g, P, PB,       \
                         bstride, (const int *)nullptr, r, zs, zb, zfin, done, rbc, nonbr, hal, rb, part, pidx, pstride);       \
  } while (0)
  if (!gs) {
    if (rq == 1) TSX_PCSH_GO(false, 0, 1);
    else if (rq == 2) TSX_PCSH_GO(false, 0, 2);
    else TSX_PCSH_GO(false, 0, 0);
  } else if (mode == 0) {
    if (rq == 2) TSX_PCSH_GO(true, 0, 2);
    else if (rq == 1) TSX_PCSH_GO(true, 0, 1);
    else TSX_PCSH_GO(true, 0, 0);
  } else if (mode == 1) TSX_PCSH_GO(true, 1, 0);
  else TSX_PCSH_GO(true, 2, 0);
#undef TSX_PCSH_GO
}

// one pass: mode as in tsx_k_pcs_rb; first = no neighbour values exist yet
// intermediate passes read their right-hand side as bf16 pairs (TSX_PC_RHS16=0: fp32 throughout).  Measured: 3_10 pass
// 43.6 -> 36.5 us; 8_16 pass 168 -> 162 us, which pays from about 14 passes on (the two passes that leave the words cost what
// four reading passes save); same iteration counts
// where the passes keep the bf16-pair words of their right-hand side (behind the iterate's records in s->vw)
unsigned *tsx_pcs_words(const tsx_solver *s) {
  float *zs = (float *)s->vw;
  return (unsigned *)(zs + (size_t)s->geo.N) + (size_t)4 * s->geo.Nc;
}
bool tsx_pcs_rhs16(const tsx_solver *s) {
  static const bool on = !(getenv("TSX_PC_RHS16") && atoi(getenv("TSX_PC_RHS16")) == 0);
  (void)s;
  return on;
}

// rq: right-hand side of an intermediate pass, see tsx_k_pcs_rb (RQ)
int tsx_pcs_pass(tsx_solver *s, int pass, int mode, float *zfin, const int *done, int rq, int part) {
  const TsxGeo &g = s->geo;
  const PcsCfg c = pcs_config(s);
  if (g.ntop == 8) {
    float *zs8 = (float *)s->vw;
    unsigned *zb8 = (unsigned *)(zs8 + (size_t)g.N);
    const bool first8 = pass == 0, gs8 = !(first8 && mode == 0);
    const int nonbr8 = first8 && mode != 0, rbc8 = pass & 1;
    if (mode != 0) rq = 0;
    if (c.lseg == 2 && c.cw == 16) pcsh_launch<2, 32, 16>(s, gs8, mode, rbc8, nonbr8, zs8, zb8, zfin, done, rq, part);
    else if (c.lseg == 2) pcsh_launch<2, 32, 8>(s, gs8, mode, rbc8, nonbr8, zs8, zb8, zfin, done, rq, part);
    else if (c.lseg == 4 && c.cw == 32) pcsh_launch<4, 16, 32>(s, gs8, mode, rbc8, nonbr8, zs8, zb8, zfin, done, rq, part);
    else if (c.lseg == 4) pcsh_launch<4, 16, 16>(s, gs8, mode, rbc8, nonbr8, zs8, zb8, zfin, done, rq, part);
    else if (c.cw == 32) pcsh_launch<8, 16, 32>(s, gs8, mode, rbc8, nonbr8, zs8, zb8, zfin, done, rq, part);
    else pcsh_launch<8, 16, 16>(s, gs8, mode, rbc8, nonbr8, zs8, zb8, zfin, done, rq, part);
    return TSX_OK;
  }
  float *zs = (float *)s->vw;                        // fp32 iterate (mode 1 writes, mode 2 reads)
  unsigned *zb = (unsigned *)(zs + (size_t)g.N);     // bf16 side-stream records of the intermediate passes
  const bool first = pass == 0;
  const bool gs = !(first && mode == 0);
  const int nonbr = first && mode != 0;
  const int rbc = pass & 1;
  if (mode != 0) rq = 0;
  if (c.lseg == 4) pcs_launch_cw<4, 16>(s, c.cw, gs, mode, rbc, nonbr, zs, zb, zfin, done, rq, part);
  else if (c.lseg == 8 && c.nseg == 8) pcs_launch_cw<8, 8>(s, c.cw, gs, mode, rbc, nonbr, zs, zb, zfin, done, rq, part);
  else if (c.lseg == 8) pcs_launch_cw<8, 16>(s, c.cw, gs, mode, rbc, nonbr, zs, zb, zfin, done, rq, part);
  else pcs_launch_cw<16, 16>(s, c.cw, gs, mode, rbc, nonbr, zs, zb, zfin, done, rq, part);
  return TSX_OK;
}

// z = M^-1 r (r = s->pc_rhs, fp32, colour-split): pc_sweeps + 1 half-grid passes, colours alternately
int tsx_pcs_apply(tsx_solver *s, float *z, const int *done) {
  const int P = s->pc_sweeps + 1;
  const bool halo = pcs_halo_on(s);
  if (halo) {
    int rc = pcs_halo_buffers(s);
    if (rc) return rc;
  }
  const bool rhs16 = tsx_pcs_rhs16(s);
  // the producer of the right-hand side has left the bf16-pair words already (fp32 Krylov vectors, tsx_k_psupdate_k32c)
  const bool words_ready = rhs16 && s->pc_words_ready && P >= 6;
  s->pc_words_ready = false;
  static const int every_env = getenv("TSX_PC_HALO_EVERY") ? atoi(getenv("TSX_PC_HALO_EVERY")) : 1;
  const int every = every_env > 0 ? every_env : 1;
  // Overlap (TSX_PC_OVERLAP != 0, several ranks): the exchange of pass p's boundary records runs on comm_stream while pass
  // p + 1 already works on the columns that touch no rank face (part 1, queued *before* the possibly host-synchronous
  // exchange so that it runs underneath it); the columns on a face follow once the records have arrived (part 2).
  static const bool overlap_env = !(getenv("TSX_PC_OVERLAP") && atoi(getenv("TSX_PC_OVERLAP")) == 0);
  const bool overlap = halo && overlap_env && tsx_overlap(s) && s->comm_stream != nullptr;  // tsx_overlap: as for the operator
  bool in_flight = false;  // an exchange has been packed and handed to comm_stream (ev_pack) but not issued yet
  // the intermediate Gauss-Seidel passes that read the bf16 right-hand side words -- passes [fp0, P - 2) -- as ONE launch
  // (tsx_k_pcs_flow): everything but the first pass (two where this application has to leave the words first) and the two fp32 passes
  const PcsCfg cfg = pcs_config(s);
  const int fp0 = words_ready ? 1 : 2;
  // the flow kernel's tiles are its own: 32 columns from 4096 columns per pass on (measured, scripts/flow_matrix.sh: 128 x 64 columns
  // 2.51 ms per solve against 2.71 with 16), 16 below; TSX_PCS_CFG's third number overrides as for the launches
  int fcw = cfg.cw;
  if (!getenv("TSX_PCS_CFG") && cfg.lseg == 4 && cfg.nseg == 16)
    fcw = ((long long)s->geo.ym * (s->geo.xm / 2) >= 4096 && (s->geo.xm / 2) % 32 == 0) ? 32 : 16;
  // several ranks: with the peer transport in its default mode (the passes send and read their boundary records themselves) the
  // flow kernel does the same inside its launch -- where EVERY rank can (tsx_pc_global_agree)
  const bool peerflow = halo && !overlap && every == 1 && pcs_peer_inkernel(s) && (s->grid.nranks <= 1 || s->pcg_flow_ok);
  const bool flow = (!halo || peerflow) && s->geo.ntop == 2 && P - 2 - fp0 >= 2 && tsx_pcs_flow_ok(s, cfg.lseg, cfg.nseg, fcw, halo);
  s->flow_last[0] = 0;
  for (int pass = 0; pass < P; ++pass) {
    const int mode = pass == P - 1 ? 2 : (pass == P - 2 ? 1 : 0);
    if (flow && pass == fp0) {
      int rcf = tsx_pcs_flow(s, fcw, fp0, P - 2, done, halo);
      if (rcf) return rcf;
      if (halo) {  // what the pass after the launch reads at the rank faces: the message of the launch's last pass, in place
        const TsxGeo &g = s->geo;
        const size_t nzp = (size_t)tsx_pcs_halo_nzp(g.Nz);
        const size_t bx = g.wrap_x ? 0 : nzp * g.ym * sizeof(unsigned), by = g.wrap_y ? 0 : nzp * g.xm * sizeof(unsigned);
        const size_t bytes[4] = {bx, bx, by, by};
        if ((rcf = tsx_peer_expect(s, bytes, &s->pch_wait, s->pch_slot))) return rcf;
        s->pch_inplace = true;
      }
      pass = P - 3;
      continue;
    }
    // a colour's intermediate visits are passes c, c + 2, ... < P - 2: the first leaves the bf16 right-hand side if another
    // one follows, the later ones read it
    const int rq = !rhs16 || mode != 0 ? 0 : (words_ready ? 2 : (pass >= 2 ? 2 : (pass + 2 < P - 2 ? 1 : 0)));
    int rc;
    // peer transport, default configuration: the pass sends its boundary records itself (TSX_PEER_INPLACE=2, the default)
    const bool xchg_after = halo && pass + 1 < P && pass % every == 0;
    s->pch_snd_on = false;
    if (xchg_after && !overlap && every == 1 && pcs_peer_inkernel(s)) {
      const TsxGeo &g = s->geo;
      const size_t nzp = (size_t)tsx_pcs_halo_nzp(g.Nz);
      const size_t bx = g.wrap_x ? 0 : nzp * g.ym * sizeof(unsigned), by = g.wrap_y ? 0 : nzp * g.xm * sizeof(unsigned);
      const size_t bytes[4] = {bx, bx, by, by};
      if ((rc = tsx_peer_prepare_send(s, bytes, &s->pch_snd))) return rc;
      s->pch_snd_on = true;
      if ((rc = tsx_pcs_pass(s, pass, mode, z, done, rq, 0))) return rc;
      s->pch_snd_on = false;
      if ((rc = tsx_peer_expect(s, bytes, &s->pch_wait, s->pch_slot))) return rc;  // the next pass reads them in place
      s->pch_inplace = true;
      continue;
    }
    if (in_flight) {
      if ((rc = tsx_pcs_pass(s, pass, mode, z, done, rq, 1))) return rc;  // interior, under the exchange
      if ((rc = pcs_halo_exchange_finish(s))) return rc;
      if ((rc = tsx_pcs_pass(s, pass, mode, z, done, rq, 2))) return rc;  // the columns on the rank faces
      in_flight = false;
    } else if ((rc = tsx_pcs_pass(s, pass, mode, z, done, rq, 0))) {
      return rc;
    }
    // what the next passes read at the rank faces; TSX_PC_HALO_EVERY = n exchanges after passes 0, n, 2n, ... only (the
    // passes in between see the neighbour rank's boundary columns n - 1 passes late at most)
    if (halo && pass + 1 < P && pass % every == 0) {
      if ((rc = pcs_halo_exchange(s, mode == 1, done, overlap))) return rc;
      in_flight = overlap;
    }
  }
  HIPCHK(hipGetLastError());
  return TSX_OK;
}

TSX_CODE_PROBE(pcs)  // tsx_host.hpp: this unit's code object as it sits in device memory (diagnostics)
