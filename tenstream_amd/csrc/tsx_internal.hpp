// tsx_internal.hpp -- internal types of libtsx (MI355X / gfx950 only).
//
// Data layout in HBM (see DESIGN.md "Data layout"):
//   The reference stores unknowns as (dof, z, x, y), dof fastest, owned by the column they sit on
//   (src/pprts_base.F90:140).  Every unknown is the *destination* of exactly one cell (the cell the
//   stream leaves, src/pprts.F90:5558-5648), so internally an unknown is stored at the index of that
//   cell, one plane per stream:
//        v[d * Nc + cell],  cell = (k * ym + j) * xm + i      (x fastest -> coalesced along i)
//   plus a "tail" of D values per column for the rows no cell writes:
//        Edn at level 0 (TOA), Eup at level Nz (surface/albedo row), side streams at level Nz (dummies)
//        v[D * Nc + d * ncol + col],  col = j * xm + i
//   N = D * (Nc + ncol) = D * L * xm * ym, the same unknown count as the reference.
//   Coefficients: one plane per (dst,src) pair, C[(dst * D + src) * Nc + cell], fp32 when lossless.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/tsx.h"

#define TSX_MAX_PARTIAL_BLOCKS 4096
#define TSX_NSLOTS 3

struct TsxGeo {
  int Nz, xm, ym;
  int ncol;        // xm*ym
  long long Nc;    // Nz*xm*ym
  long long N;     // D*(Nc+ncol)
  int D, ntop, nside;
  int wrap_x, wrap_y;  // 1: neighbour in that direction is this rank itself and faces wrap in-kernel
  int pc_tile_x, pc_tile_y;  // analysis knob (TSX_PC_TILE=tx,ty): the preconditioner drops couplings across tile edges as
                             // it does across rank faces, to study rank-local preconditioning on one GPU; 0 = off
};

// device-resident scalars of the Krylov loop (one instance per solver)
struct TsxScalars {
  double rho, rho_old, alpha, omega, beta;
  double rnorm, rnorm0;
  double rnorm_true;       // fp32 Krylov vectors: norm of the residual at its last replacement by b - A x (fp64)
  double red[TSX_NSLOTS];  // reduced (and, multi-rank, all-reduced) sums of the last stage
  double rtol, atol, dtol;
  int maxit;
  int its;
  int reason;
  int done;
  int nhist;
  int restart;   // set by the host before a breakdown restart: INIT keeps its / history / rnorm0
  int nranks;    // explicit solver: its residual is the mean over ranks of the local norms
  int aux_flag;  // scratch flag for kernels outside the loop (import: is the guess nonzero?)
  int half;      // the stop rule was met by s = r - alpha v, in the middle of an iteration (TSX_STAGE_HALF): x still lacks alpha p-hat
  double half_margin;  // factor on rtol / atol for that test (< 1 where the decision is re-taken on the true residual afterwards)
  int flow_err;        // a bounded wait of the flow kernel (tsx_k_pcs_flow) expired: the host reports TSX_ERR_HIP at its next look
  double hist[100];
};

struct TsxLutHost {  // device copy of one LUT + its description
  bool ready;
  int ndim, nvec;
  int n[8];
  long long nentries;
  float *d_axes;
  float *d_table;
};

struct TsxSolSlot {  // one stored solution (initial guess of the next solve with that uid), compressed to real32
  float *x32;    // diffuse streams, internal layout, N values
  float *e32;    // direct streams, S planes of (Nz+1)*ncol values (null if the slot never held a solar solution)
  int lsolar;
};

struct TsxPeer;  // tsx_peer.hip
struct TsxLog;   // below: log events / roctx ranges

// peer transport (tsx_peer_dev.hpp): what a kernel needs to consume a face message in place -- the sequence number that
// announces it, per face W, E, S, N
struct TsxPeerWait {
  char *mine;                   // my mailbox; null: no peer transport, nothing to wait for
  unsigned long long want[4];   // 0: face inactive
  unsigned long long ticks;     // bound of the wait (100 MHz wall clock)
  int heavy;                    // full system-scope fences (tsx_peer_dev.hpp)
};

// one exchange as its kernels see it (faces W, E, S, N)
struct TsxPeerXArgs {
  char *mine;                   // my mailbox
  char *remote[4];              // mailbox of the neighbour behind face q
  const char *src[4];           // send: the caller's send buffers;  recv: unused
  char *dst[4];                 // recv: the caller's receive buffers
  unsigned long long bytes[4];  // payload per face (0: face inactive)
  unsigned long long n[4];      // sequence number of this message per face
  unsigned long long ackn[4];   // send: messages received through face q by kernels that precede this one on the stream
  unsigned long long cap, data_off, ticks;
  unsigned int *blkctr;         // [4] workgroups that have finished their slice (local memory)
  int heavy;                    // 1: system-scope release / acquire fences around the flags (TSX_PEER_FENCES=1), see below
  const int *done;              // unused (the sequence numbers must stay in step whatever the convergence flag says)
};

// the flow kernel's view of the peer transport for one launch (tsx_k_pcs_flow FPEER, tsx_peer_flow_view), faces W, E, S, N
struct TsxFlowPeer {
  char *mine;
  char *remote[4];                  // null: face not active (the rank wraps onto itself there)
  unsigned long long R0[4], S0[4];  // messages received / sent through the face before the launch (host side only)
  unsigned long long cap, data_off, ticks;
  unsigned long long tag_off, tag_edge;  // the tags: [face][parity][tag_edge] words at tag_off of a mailbox
  unsigned *fctr;                   // (unused)
  int heavy;
};

struct tsx_solver {
  tsx_grid grid;
  TsxGeo geo;
  int device;
  hipStream_t stream;
  bool own_stream;

  // operator
  void *coef;          // planes, float or double
  int coef_bytes;      // 4 or 8
  void *coef_h;        // packed fp16 copy of the blocks for the preconditioner (tsx_k_pack_p16; built in prepare_ksp)
  bool coef_h_valid, pc_half;
  bool coef_h_dd;      // ... with groups 1..7 stored per distinct block (tsx_dedup.hip)
  bool coef_h_c16 = false;  // ... those in the 12-slot layout with fp16 couplings (tsx_k_pcs_pack_ent16)
  bool coef_h_scan;    // the packed copy is in the scan kernels' layout "S16" (tsx_kernels_pcs.hpp; always colour-split)
  bool coef_h_split;   // layout of the packed copy: colour-split (red-black preconditioner) or natural
  bool pc_split;       // the preconditioner's private arrays (packed blocks, fp32 rhs) are in colour-split order
  uint8_t *l1d;        // [Nz]
  double *a11, *a12;   // [Nc] cell-indexed (only read where l1d)
  double *albedo;      // [ncol]
  bool have_coeffs;
  bool any_l1d;
  // shared storage of bit-identical blocks (tsx_dedup.hip): planes over entries + per-cell entry index
  bool dd_valid, dd_on;
  bool dd_from_coords = false;  // the current grouping was made from the cells' LUT coordinates (tsx_dedup_from_coords): the next set may reuse it
  bool dd_reused = false;       // ... and the current one is the previous set's, found still valid
  bool coef_dense_valid = true;  // s->coef holds every cell's block (false: only the shared storage does, tsx_dedup_from_coords)
  bool pe_entry_major = false;  // the scan passes' per-block records are stored entry-major (tsx_k_pcs_pack_ent16)
  bool dd_pc = false;      // the index / entries group NEAR-identical blocks and serve the preconditioner only (tsx_dedup.hip);
                           // the operator then works on every cell's exact block (dd_on stays false)
  int dd_nent_near = 0;    // entries of that grouping (0: not attempted)
  // what the preconditioner packs from and indexes with: the near-identical grouping where it pays (own arrays: pc_own), else
  // the bit-identical one (aliases of the dd_ arrays); valid when dd_on || dd_pc
  float *pc_coef = nullptr;       // [D*D][pc_nent] plane-major
  int *pc_cidx_split = nullptr;   // [Nc] colour-split order
  int *pc_ent_cell = nullptr;     // [pc_nent]
  int pc_nent = 0, pc_cap = 0;
  float *pcn_coef = nullptr;      // the near grouping's own allocations (kept across coefficient sets)
  int *pcn_cidx_split = nullptr, *pcn_ent_cell = nullptr;
  int dd_nent, dd_cap;
  float *dd_coef;          // [D*D][dd_nent] plane-major (preconditioner packing)
  float *dd_coef_e = nullptr;  // [dd_nent][D*D] entry-major copy behind it (operator apply)
  int *dd_cidx;            // [Nc] natural cell order
  int *dd_cidx_split;      // [Nc] colour-split order (the preconditioner's)
  int *dd_ent_cell;        // [dd_nent] representative cell of every entry
  unsigned *pch_send[4], *pch_recv[4];  // preconditioner halo (bf16-pair records of the boundary columns), W E S N
  // peer transport: the next pass reads the records in place, from these mailbox slots, after waiting for pch_wait (tsx_pcs.hip)
  bool pch_inplace;
  const void *pch_slot[4];
  TsxPeerWait pch_wait;
  bool pch_snd_on;           // ... and the pass about to be launched stores its boundary records into the neighbours' mailboxes itself
  TsxPeerXArgs pch_snd;
  // several ranks: whether EVERY rank runs the scan red-black passes with the halo exchange (tsx_pc_global_agree); the
  // exchange is a matched send/recv with the neighbours, so it is on everywhere or nowhere.  pcg_key = the settings the
  // answer was agreed for (-1: not yet)
  int pcg_key = -1;
  bool pcg_halo_ok = false;
  bool pcg_flow_ok = false;   // ... and every rank can run the intermediate passes as one launch with its faces (tsx_k_pcs_flow FPEER)
  // shared storage of identical packed preconditioner records (tsx_records_share): per-cell index, table, capacity (records)
  bool pcr_on = false;
  int *pcr_idx = nullptr, *pcr_ent = nullptr;
  void *pcr_tab = nullptr;
  long long pcr_n = 0, pcr_cap = 0;
  int pcr_have_R = 0;           // > 0: pcr_idx / pcr_ent hold a grouping of the cells by their R records (of some earlier coefficient set)
  const void *pcr_have_P = nullptr;  // ... of the record planes at this address
  bool pcr_reused = false;      // the last tsx_records_share took the previous grouping over
  // the intermediate passes of an application as one launch (tsx_pcs_flow.hip): ticket / epoch words and the tiles' progress words
  void *flow_state = nullptr;
  unsigned *flow_prog = nullptr;
  void *flow_pr_dev = nullptr;    // device copy of the flow kernel's view of the peer transport (TsxFlowPeer), and what it holds
  TsxFlowPeer *flow_pr_shadow = nullptr;
  void *flow_zb8 = nullptr;       // the iterate records as 8-byte granules {bf16 pair, tag} (fat flow kernel), [4][Nc]
  int flow_last[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // the last application of M^-1: {flow kernel used, p0, p1, columns per tile, fat, granules, tiles per pass, workgroups}
  unsigned flow_epoch_bound = 0;  // host-side upper bound of the device's epoch word (tags and progress words restart before it wraps)
  int flow_prog_cap = 0;
  int flow_capacity[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // resident workgroups of tsx_k_pcs_flow<4, 16, 32 | 16, lean | fat, records per block | per cell> on this device (0: not asked yet)
  void *dd_scratch;        // work space of the build (hashes, table, scan)
  bool x_is_zero = false;      // the initial guess in vx is known to be zero on every rank (krylov_begin then skips A x0)
  bool dd_hash_ready = false;  // the hashes of the current blocks already sit in dd_scratch (left by tsx_k_lut_diff2diff)
  size_t dd_scratch_bytes;
  int n1d;             // number of 1-D layers (unconstrained_fraction = 1 - n1d/Nz, src/pprts.F90:721-723)
  TsxLutHost lut_diff;

  // Krylov work vectors (internal layout, N doubles each)
  double *vx, *vb, *vr, *vrhat, *vp, *vv, *vs, *vt, *vph, *vsh;
  double *vw;          // work vector of the multi-sweep preconditioner
  double *pc_tmp;      // column preconditioner: tsx_pc_ntmp planes of Nc doubles
  int pc, pc_sweeps;   // active preconditioner of the running solve
  bool mixed;          // fp32 storage of preconditioned directions and shadow residual
  bool pc_words_ready = false;  // the bf16-pair words of the right-hand side of the NEXT preconditioner application are already in
                                // place (left by tsx_k_psupdate_k32c): every intermediate pass reads them, none writes them
  bool k32 = false;    // ... and of the recurrence vectors r, s, v, t (tsx_ksp_opts.fp32_directions = 2), residual replacement in fp64
  // staging in reference layout (for TSX_HOST callers and conversion)
  double *stage_a, *stage_b;

  // halo buffers: [nside/2][Nz][edge] doubles per direction
  double *sendW, *sendE, *sendS, *sendN;
  double *recvW, *recvE, *recvS, *recvN;
  size_t halo_x_elems, halo_y_elems;

  double *partials;    // [TSX_NSLOTS][TSX_MAX_PARTIAL_BLOCKS]
  TsxScalars *scal;    // device
  TsxScalars *scal_host;  // pinned host mirror

  // ---- whole-g-point pipeline state (tsx_pipeline.hpp)
  bool have_sun;
  double sun_phi, sun_theta, sun_mu, sun_costheta, sun_symphi;
  int sun_xinc, sun_yinc;
  TsxLutHost lut_T, lut_S;
  float *dirT, *dirS;            // direct coefficient planes (S*S, S*D)
  bool dir_coeffs_valid;
  int its_hint_cold = 0, its_hint_warm = 0;  // iterations of the last converged Krylov solve from a zero / nonzero guess (krylov_run)
  unsigned long long its_hint_key = 0;       // ... and the settings they belong to (tolerances, preconditioner, passes, precision)
  bool in_retry = false;                     // the conservative retry solver is running (krylov_run_with_retry)
  bool have_albedo = false;  // s->albedo holds the caller's surface albedo (set_coeffs / set_optprop / set_optical_properties / setup_b_solar)
  bool dir_seam = false, dir_seam_S = false;  // dirT (and dirS) were handed over by tsx_dir_set_coeffs (the direct seam), not looked up
  double *dd_colsum;         // [D][dd_nent]: sum over dst of c(src, dst) per distinct block (absorptivity / emissivity terms of setup_b_thermal, flx_div)
  long long dd_colsum_cap;
  void *cell_samp;           // float4 per cell: the LUT coordinates in cell order (tsx_k_cell_samples) ...
  const void *cell_samp_src[4];  // ... of these arrays (every writer of them is followed by the diffuse lookup, which renews it)
  double cell_samp_dx;
  double *d_kabs, *d_ksca, *d_g, *d_dz;  // device copies, reference layout (k fastest)
  double opt_dx, opt_dy;
  bool have_optprop;
  double *a13, *a23, *a33;       // cell-indexed, 1-D layers only
  double *planck;                // (L, xm, ym) reference layout
  double *bsrfc = nullptr;       // (xm, ym): atm%Bsrfc, the surface's own Planck emission (planck_srfc of set_optical_properties); null = not given
  float *v32;                    // fp32 copy of the preconditioner's right-hand side s (mixed path)
  float *p32;                    // the search direction p, kept in fp32 only on the mixed + preconditioned path
  const float *pc_rhs;           // which of the two the next tsx_pc_apply (fp32 directions) reads
  double *edir_a, *edir_b;       // direct streams, current / scratch
  double *dsend[4], *drecv[4];   // direct-beam face buffers W, E, S, N (several ranks)
  void *dsc, *dsc_host;          // TsxDirScalars device / pinned
  double *abso;                  // (Nz, xm, ym) reference layout
  int last_lsolar;
  bool have_solution;
  void *slots;         // std::map<int, TsxSolSlot>*: solver%solutions(uid), src/pprts_base.F90:163-190
  int cur_uid;         // which uid the working vectors vx / edir_a belong to
  bool guess_foreign;  // the working vectors hold another uid's solution as initial guess (any kind of radiation)
  int niter_dir;
  double dir_rtol = -1.0, dir_atol = -1.0;  // direct sweep: caller's -solar_dir_ksp_rtol / _atol / _max_it (< 0: defaults)
  int dir_maxit = -1;

  void *nccl_comm;     // ncclComm_t when nranks > 1 (or force_halo with comm): the all-reduces on the solver stream
  void *nccl_comm_x = nullptr;  // a second communicator (ncclCommSplit) for the face exchanges on comm_stream: one communicator
                                // must not be driven from two streams at once; null = none (the exchanges use nccl_comm)
  TsxPeer *peer = nullptr;      // device-resident peer transport (tsx_peer.hip); takes precedence when attached
  bool comm_ready;
  tsx_exchange_fn xchg_cb;      // host-staged transport (MPI hosts, tests)
  tsx_allreduce_fn allred_cb;
  void *cb_ctx;
  double *host_send[4], *host_recv[4];  // pinned staging, order W,E,S,N

  hipEvent_t ev0, ev1;
  hipEvent_t ev_imp0, ev_imp1, ev_exp1;  // import / export timing of tsx_diff_solve
  hipStream_t comm_stream;   // face exchange runs here while the interior SpMV runs on `stream`
  hipEvent_t ev_pack, ev_recv;
  int max_lds;               // hipDeviceAttributeMaxSharedMemoryPerBlock
  int overlap_env;           // TSX_OVERLAP: -1 unset, else 0 / 1 (tsx_overlap, tsx_host.hpp: interior + frame launches around an exchange)
  double *pcx_rec = nullptr;  // [7][Nc] fp64 column recurrences of the exact scan preconditioner (tsx_pcx.hip), natural cell order
  bool pcx_valid = false;     // ... belong to the current coefficient set
  double *pcx_vz = nullptr;   // [2][N]: colour-split copies of the right-hand side and the iterate of that path
  TsxLog *log = nullptr;     // the reference's log events for this path + roctx ranges (tsx_log_enable; off: null)
};

// The reference brackets the phases of a solve with PETSc log events (solver%logs, src/pprts_base.F90:176-209; begun / ended at
// src/pprts.F90:1785-2077 set_optprop, :2694-2756 compute_Edir, :2760-2818 compute_Ediff, :2952-2954 setup_Mdiff, :3012-3021
// solve_Mdiff, :3422-3489 get_coeff_diff2diff, :5197-5479 compute_absorption) and reads them with -log_view.  The same names here:
// a count and the DEVICE time between two events recorded on the solver's stream per phase (tsx_log_get), and a roctx range of
// that name around the host code that enqueues it, so that a rocprofv3 --marker-trace of a spectral loop shows its g-points.
enum TsxLogEvent {
  TSX_EV_SET_OPTPROP = 0, TSX_EV_GET_COEFF_DIFF2DIFF, TSX_EV_GET_COEFF_DIR2DIR, TSX_EV_COMPUTE_EDIR, TSX_EV_SOLVE_MDIR,
  TSX_EV_SETUP_DIFF_SRC, TSX_EV_COMPUTE_EDIFF, TSX_EV_SETUP_MDIFF, TSX_EV_SOLVE_MDIFF, TSX_EV_COMPUTE_ABSORPTION, TSX_EV_GET_RESULT,
  TSX_EV_COUNT
};
struct TsxLogPending { int ev; hipEvent_t a, b; };
struct TsxLog {
  long long count[TSX_EV_COUNT] = {0};
  double ms[TSX_EV_COUNT] = {0};
  std::vector<TsxLogPending> pending;
  std::vector<hipEvent_t> pool;
};
void tsx_log_begin(tsx_solver *s, int ev, hipEvent_t *a);
void tsx_log_end(tsx_solver *s, int ev, hipEvent_t a);
struct TsxLogScope {  // begin at construction, end on every exit path
  tsx_solver *s;
  int ev;
  hipEvent_t a = nullptr;
  TsxLogScope(tsx_solver *s_, int ev_) : s(s_), ev(ev_) { if (s->log) tsx_log_begin(s, ev, &a); }
  ~TsxLogScope() { if (s->log) tsx_log_end(s, ev, a); }
  TsxLogScope(const TsxLogScope &) = delete;
  TsxLogScope &operator=(const TsxLogScope &) = delete;
};

void tsx_set_error(const std::string &msg);
