// tsx_pcs_flow.hip -- host side of tsx_k_pcs_flow (tsx_kernels_pcs.hpp): the intermediate red-black passes of one application of
// the scan preconditioner as ONE launch, a workgroup per (pass, tile) work item, items waiting for their four neighbour tiles'
// progress words instead of a kernel boundary.  A translation unit of its own so that it compiles next to tsx_pcs.hip.
//
// Replaces, per application of M^-1, 24-25 of the 28 launches the reference's PCILU application is replaced by here
// (src/pprts.F90:4350-4371); same arithmetic per cell, bit-identical results.
#include <stdio.h>
#include <string.h>

#include "tsx_host.hpp"
#include "tsx_peer.hpp"
#include "tsx_peer_dev.hpp"
#include "tsx_kernels_pcs.hpp"

#ifndef TSX_PCS_C16
#define TSX_PCS_C16 1
#endif

static int flow_env() {
  const char *e = getenv("TSX_PC_FLOW");  // 0: a launch per pass (A/B knob, and what the parity test compares with); read per call
  return e ? atoi(e) : 1;
}

template <int CW, bool IDX, bool FAT>
static const void *flow_kernel() {
  return (const void *)tsx_k_pcs_flow<4, 16, CW, IDX, TSX_PCS_C16 != 0, FAT, false>;
}

// resident workgroups of the flow kernel on this device.  Only a bound on the useful grid: tickets make any grid correct.
static int flow_capacity(tsx_solver *s, int cw, bool fat) {
  // the instantiation that will be launched: per-block records behind the index (IDX) or every cell's own -- their register use, and
  // so their residency, differs; the granule and rank-face variants of a body are within a few registers of it
  const bool idx = s->coef_h_dd;
  int &cap = s->flow_capacity[(cw == 32 ? 0 : 1) + (fat ? 2 : 0) + (idx ? 0 : 4)];
  if (cap > 0) return cap;
  int per_cu = 0, cus = 0;
  const void *k = idx ? (cw == 32 ? (fat ? flow_kernel<32, true, true>() : flow_kernel<32, true, false>())
                                  : (fat ? flow_kernel<16, true, true>() : flow_kernel<16, true, false>()))
                      : (cw == 32 ? (fat ? flow_kernel<32, false, true>() : flow_kernel<32, false, false>())
                                  : (fat ? flow_kernel<16, false, true>() : flow_kernel<16, false, false>()));
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, cw * 16, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device) != hipSuccess || cus < 1) cus = 256;
  cap = per_cu * cus;
  return cap;
}

// faces: the rank has real neighbours (or force_halo) -- the tiles on its faces exchange their records with the neighbour ranks
// through the peer transport inside the launch (FPEER: the fat body only)
bool tsx_pcs_flow_ok(tsx_solver *s, int lseg, int nseg, int cw, bool faces) {
  const TsxGeo &g = s->geo;
  if (!flow_env()) return false;
  if (g.ntop != 2 || lseg != 4 || nseg != 16 || (cw != 32 && cw != 16)) return false;  // the instantiated configurations
  if (!faces && !(g.wrap_x && g.wrap_y)) return false;
  // several instances of this process inside solves at once (their own streams): the launches leave the chip to the others between
  // the passes, the flow kernel's waiting workgroups do not (tsx_active_solves; TSX_PC_FLOW=1 set explicitly keeps the flow kernel)
  if (!faces && tsx_active_solves() > 1 && !getenv("TSX_PC_FLOW")) return false;
  if (g.pc_tile_x > 0 || g.pc_tile_y > 0) return false;
  const int h = g.xm >> 1;
  if (h % cw != 0) return false;  // a tile is CW columns of ONE row
  if (!tsx_pcs_rhs16(s)) return false;
  if (faces) {
    if (getenv("TSX_FLOW_PEER") && atoi(getenv("TSX_FLOW_PEER")) == 0) return false;  // A/B: the passes as launches on several ranks
    if (g.ym % 2 != 0) return false;
    // only where a pass's tiles are resident at once (the fat body: two waves per SIMD).  Larger shards were measured slower with the
    // faces inside the launch than with a launch per pass, twice: round 5 with the fat body (256 x 128 columns 8.85 -> 9.02 ms per
    // solve, 128 x 256 8.68 -> 9.08, profiles/r05/flow_peer_ab_large.txt), round 6 with the LEAN body carrying the faces (four waves
    // per SIMD, the face columns send after the scan; bit-identical, tests force it with TSX_FLOW_FAT=0): 256 x 128 8.76 -> 10.58,
    // 128 x 256 8.58 -> 10.57, 256 x 256 15.98 -> 18.17 ms (profiles/r06/flow_peer_lean_ab.txt) -- a workgroup there runs ~25 items
    // one after the other, and every face item carries the drain of its uncached stores and the tag round trip on its own back.
    // TSX_FLOW_PEER_LEAN=1 switches the lean variant on up to two tiles per resident workgroup, TSX_FLOW_PEER_ANY=1 lifts every bound
    const long long nt = (long long)(h / cw) * g.ym;
    if (getenv("TSX_FLOW_PEER_ANY") && atoi(getenv("TSX_FLOW_PEER_ANY"))) return true;
    if (nt <= flow_capacity(s, cw, true)) return true;
    if (!(getenv("TSX_FLOW_PEER_LEAN") && atoi(getenv("TSX_FLOW_PEER_LEAN")) != 0)) return false;
    return nt <= 2ll * flow_capacity(s, cw, false);
  }
  // beyond about two tiles per resident workgroup a pass is bound by its instruction stream and the launch boundary costs nothing
  // next to it: 256 x 256 columns (1024 tiles) 12.01 -> 11.65 ms per solve, 512 x 512 (4096 tiles) 53.7 -> 54.4 ms
  const long long ntiles = (long long)(h / cw) * g.ym;
  if (!getenv("TSX_FLOW_FAT") && ntiles > 2ll * flow_capacity(s, cw, false)) return false;
  return true;
}

// state, progress words, granules.  The epoch only grows (tags and progress words of earlier launches stay behind it); before the
// host's bound on it reaches 2^30 everything restarts from zero, in stream order.
static int flow_ensure(tsx_solver *s, int ntiles, bool fat, int npass) {
  const TsxGeo &g = s->geo;
  if (!s->flow_state) {
    HIPCHK(tsx_dev_malloc(&s->flow_state, sizeof(TsxFlowState)));
    HIPCHK(hipMemsetAsync(s->flow_state, 0, sizeof(TsxFlowState), s->stream));
    s->flow_epoch_bound = 0;
  }
  if (s->flow_prog_cap < 2 * ntiles) {
    if (s->flow_prog) {
      HIPCHK(hipStreamSynchronize(s->stream));
      HIPCHK(tsx_dev_free(s->flow_prog));
      s->flow_prog = nullptr;
    }
    HIPCHK(tsx_dev_malloc((void **)&s->flow_prog, sizeof(unsigned) * 2 * (size_t)ntiles));
    HIPCHK(hipMemsetAsync(s->flow_prog, 0, sizeof(unsigned) * 2 * (size_t)ntiles, s->stream));  // zero is behind every epoch
    s->flow_prog_cap = 2 * ntiles;
  }
  const size_t zb8_bytes = sizeof(uint2) * 4 * (size_t)g.Nc;
  if (fat && !s->flow_zb8) {
    HIPCHK(tsx_dev_malloc(&s->flow_zb8, zb8_bytes));
    HIPCHK(hipMemsetAsync(s->flow_zb8, 0, zb8_bytes, s->stream));
  }
  // (TSX_FLOW_EPOCH_LIMIT: tests lower the bound so that the restart is exercised)
  static const unsigned epoch_limit = getenv("TSX_FLOW_EPOCH_LIMIT") ? (unsigned)atoll(getenv("TSX_FLOW_EPOCH_LIMIT")) : (1u << 30);
  if (s->flow_epoch_bound > epoch_limit) {
    HIPCHK(hipMemsetAsync(s->flow_state, 0, sizeof(TsxFlowState), s->stream));
    HIPCHK(hipMemsetAsync(s->flow_prog, 0, sizeof(unsigned) * (size_t)s->flow_prog_cap, s->stream));
    if (s->flow_zb8) HIPCHK(hipMemsetAsync(s->flow_zb8, 0, zb8_bytes, s->stream));
    s->flow_epoch_bound = 0;
  }
  s->flow_epoch_bound += (unsigned)npass + 1u;
  return TSX_OK;
}

template <int CW>
static int flow_launch(tsx_solver *s, int p0, int p1, const int *done, bool faces) {
  const TsxGeo &g = s->geo;
  const int h = g.xm >> 1;
  const int R = h / CW, ntiles = R * g.ym;
  // FAT (every neighbour-independent load in front of the wait, two waves per SIMD, records as granules): where a pass has at
  // most as many tiles as such workgroups are resident -- there an item is a chain of latencies; TSX_FLOW_FAT=0 / 1 overrides (A/B)
  const int cap_fat = flow_capacity(s, CW, true);
  bool fat = ntiles <= cap_fat;
  if (const char *e = getenv("TSX_FLOW_FAT")) fat = atoi(e) != 0;
  int rc = flow_ensure(s, ntiles, fat, p1 - p0);
  if (rc) return rc;
  float *zs = (float *)s->vw;
  unsigned *zb = (unsigned *)(zs + (size_t)g.N);
  unsigned *rb = zb + (size_t)4 * g.Nc;
  const uint4 *P = (const uint4 *)s->coef_h;
  const float *r = (const float *)s->pc_rhs;
  const bool dd = s->coef_h_dd;
  const int *cidx = (const int *)s->pc_cidx_split;
  const long long nent = s->pc_nent;
  const uint4 *PE = P + g.Nc;
  const int *pidx = dd && s->pcr_on ? (const int *)s->pcr_idx : (const int *)nullptr;
  const uint4 *PT = (const uint4 *)s->pcr_tab;
  TsxFlowArgs f;
  f.st = (TsxFlowState *)s->flow_state;
  f.prog = s->flow_prog;
  f.p0 = p0;
  f.p1 = p1;
  f.ntiles = ntiles;
  f.R = R;
  f.err = &s->scal->flow_err;
  f.zb8 = (uint2 *)s->flow_zb8;
  f.prp = nullptr;
  memset(f.R0, 0, sizeof(f.R0));
  memset(f.S0, 0, sizeof(f.S0));
  if (faces) {
    const size_t nzp = (size_t)tsx_pcs_halo_nzp(g.Nz);
    const size_t bx = g.wrap_x ? 0 : nzp * g.ym * sizeof(unsigned), by = g.wrap_y ? 0 : nzp * g.xm * sizeof(unsigned);
    const size_t bytes[4] = {bx, bx, by, by};
    TsxFlowPeer pr;
    if ((rc = tsx_peer_flow_view(s, bytes, p1 - p0, &pr))) return rc;
    memcpy(f.R0, pr.R0, sizeof(f.R0));
    memcpy(f.S0, pr.S0, sizeof(f.S0));
    memset(pr.R0, 0, sizeof(pr.R0));
    memset(pr.S0, 0, sizeof(pr.S0));
    // the part that does not change between launches lives in device memory (re-sent only when the transport changed)
    if (!s->flow_pr_dev) {
      HIPCHK(tsx_dev_malloc(&s->flow_pr_dev, sizeof(TsxFlowPeer)));
      s->flow_pr_shadow = new TsxFlowPeer();
      memset((void *)s->flow_pr_shadow, 0xff, sizeof(TsxFlowPeer));
    }
    if (memcmp((const void *)s->flow_pr_shadow, (const void *)&pr, sizeof(pr)) != 0) {
      HIPCHK(hipStreamSynchronize(s->stream));
      HIPCHK(hipMemcpy(s->flow_pr_dev, &pr, sizeof(pr), hipMemcpyHostToDevice));
      memcpy((void *)s->flow_pr_shadow, (const void *)&pr, sizeof(pr));
    }
    f.prp = (const TsxFlowPeer *)s->flow_pr_dev;
  }
  {
    static const double tmo = getenv("TSX_FLOW_TIMEOUT_S") ? atof(getenv("TSX_FLOW_TIMEOUT_S")) : 5.0;
    f.ticks = (unsigned long long)(tmo * 1e8);  // wall_clock64: 100 MHz
  }
  // two passes' worth of tiles can be runnable at a time (a tile of pass p + 2 needs its neighbours' pass p + 1); more
  // workgroups than that only poll
  long long grid = 2ll * ntiles;
  int cap = fat ? cap_fat : flow_capacity(s, CW, false);
  // rank processes that share this device (the multi-process tests, a one-GPU box): each rank's resident grid spins on its
  // neighbours' tags, and a grid that fills the device keeps the neighbours' workgroups from being dispatched until the driver
  // time-slices the processes -- leave them their share.  One process per device (production): unchanged
  if (faces) {
    const int co = tsx_peer_colocated(s);
    if (co > 1) cap = cap / co > 1 ? cap / co : 1;
  }
  if (grid > cap) grid = cap;
  if (const char *e = getenv("TSX_FLOW_GRID")) {
    const int v = atoi(e);
    if (v > 0) grid = v;
  }
  const long long nitems = (long long)(p1 - p0) * ntiles;
  if (grid > nitems) grid = nitems;
  constexpr bool C16 = TSX_PCS_C16 != 0;
  // granules (the records as their own flags, TsxGran) only where most workgroups would otherwise idle: the workgroups ahead of the
  // wave front spin on their sixteen 8-byte loads per lane, which slows a chip whose every workgroup has work.  Measured per solve
  // (gpurun_out/flow_matrix_1.txt -> profiles/r05): 64 x 64 columns, 128 tiles of 16 columns: 1.86 ms with progress words, 1.71 with
  // granules; 128 x 64, 128 tiles of 32: 2.51 / 2.51; 128 x 128, 256 tiles of 32 on 256 workgroups: 3.64 / 4.37.
  bool gran = fat && 4 * ntiles <= cap_fat;
  if (const char *e = getenv("TSX_FLOW_GRAN")) gran = fat && atoi(e) != 0;
  if (faces) gran = false;
#define TSX_FLOW_GO(IDXV, FATV, GRV)                                                                                               \
  hipLaunchKernelGGL((tsx_k_pcs_flow<4, 16, CW, IDXV, C16, FATV, GRV>), dim3((unsigned)grid), dim3(CW * 16), 0, s->stream, g, P, r, zb, \
                     done, IDXV ? cidx : (const int *)nullptr, IDXV ? nent : 0ll, IDXV ? PE : (const uint4 *)nullptr, rb,          \
                     IDXV ? pidx : (const int *)nullptr, IDXV ? PT : (const uint4 *)nullptr,                                       \
                     (IDXV && s->pe_entry_major) ? TSX_PCS_ENT16_SLOTS : 1, f)
#define TSX_FLOW_GOP(IDXV, FATV)                                                                                                    \
  hipLaunchKernelGGL((tsx_k_pcs_flow<4, 16, CW, IDXV, C16, FATV, false, true>), dim3((unsigned)grid), dim3(CW * 16), 0, s->stream, g, P, \
                     r, zb, done, IDXV ? cidx : (const int *)nullptr, IDXV ? nent : 0ll, IDXV ? PE : (const uint4 *)nullptr, rb,   \
                     IDXV ? pidx : (const int *)nullptr, IDXV ? PT : (const uint4 *)nullptr,                                       \
                     (IDXV && s->pe_entry_major) ? TSX_PCS_ENT16_SLOTS : 1, f)
  if (faces) {
    if (dd) {
      if (fat) TSX_FLOW_GOP(true, true);
      else TSX_FLOW_GOP(true, false);
    } else {
      if (fat) TSX_FLOW_GOP(false, true);
      else TSX_FLOW_GOP(false, false);
    }
  } else
  if (dd) {
    if (gran) TSX_FLOW_GO(true, true, true);
    else if (fat) TSX_FLOW_GO(true, true, false);
    else TSX_FLOW_GO(true, false, false);
  } else {
    if (gran) TSX_FLOW_GO(false, true, true);
    else if (fat) TSX_FLOW_GO(false, true, false);
    else TSX_FLOW_GO(false, false, false);
  }
#undef TSX_FLOW_GO
#undef TSX_FLOW_GOP
  HIPCHK(hipGetLastError());
  const int rec[8] = {faces ? 2 : 1, p0, p1, CW, fat ? 1 : 0, gran ? 1 : 0, ntiles, (int)grid};
  memcpy(s->flow_last, rec, sizeof(rec));
  return TSX_OK;
}

int tsx_pcs_flow(tsx_solver *s, int cw, int p0, int p1, const int *done, bool faces) {
  if (p1 <= p0) return TSX_OK;
  return cw == 32 ? flow_launch<32>(s, p0, p1, done, faces) : flow_launch<16>(s, p0, p1, done, faces);
}

#ifdef TSX_FLOW_TRACE
// analysis builds only: the stamps of the last flow launch(es), n items of 12 words
extern "C" int tsx_debug_flow_trace(unsigned long long *out, int n) {
  if (n > TSX_FLOW_TL_N) n = TSX_FLOW_TL_N;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(tsx_flow_tl), sizeof(unsigned long long) * 12 * (size_t)n));
  return TSX_OK;
}
#endif

TSX_CODE_PROBE(pcsflow)  // tsx_host.hpp: this unit's code object as it sits in device memory (diagnostics)
