// tsx_api.hip -- C-ABI of libtsx (see include/tsx.h).  Host orchestration: HIP streams/events,
// device-resident Krylov loop (no host round trip per iteration), RCCL halo exchange.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <type_traits>

#include <stdlib.h>

#include "tsx_host.hpp"
#include "tsx_peer.hpp"
#include "tsx_peer_dev.hpp"
#include "tsx_kernels.hpp"
#include "tsx_pipeline.hpp"

// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
void tsx_set_error(const std::string &msg) { g_err = msg; }
extern "C" const char *tsx_last_error(void) { return g_err.c_str(); }
extern "C" int tsx_version(void) { return TSX_VERSION; }

extern "C" int tsx_abi_sizes(int32_t *sizes3) {
  if (!sizes3) return TSX_ERR_ARG;
  sizes3[0] = (int32_t)sizeof(tsx_grid);
  sizes3[1] = (int32_t)sizeof(tsx_ksp_opts);
  sizes3[2] = (int32_t)sizeof(tsx_ksp_result);
  return TSX_OK;
}

extern "C" int tsx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ------------------------------------------------------------------------------------------------
// RCCL, bound lazily (dlopen by soname so that a process that already loaded RCCL -- e.g. through
// torch.distributed -- shares that copy).
typedef struct { char internal[128]; } tsx_ncclUniqueId;
typedef void *tsx_ncclComm_t;
struct RcclApi {
  void *h = nullptr;
  int (*GetUniqueId)(tsx_ncclUniqueId *) = nullptr;
  int (*CommInitRank)(tsx_ncclComm_t *, int, tsx_ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(tsx_ncclComm_t) = nullptr;
  int (*CommSplit)(tsx_ncclComm_t, int, int, tsx_ncclComm_t *, void *) = nullptr;  // optional (RCCL >= 2.18)
  int (*Send)(const void *, size_t, int, int, tsx_ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, tsx_ncclComm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, tsx_ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;
static std::mutex g_rccl_mu;
enum { TSX_NCCL_FLOAT64 = 8, TSX_NCCL_SUM = 0 };  // ncclFloat64 / ncclSum in rccl.h

static int rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.h) return TSX_OK;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  // TSX_RCCL_LIB: another library with librccl's entry points -- the tests' double (tests/c/fake_rccl.cpp: the same calls over
  // shared memory, so that this transport's host code runs with several ranks on a one-GPU box).  Never set in production.
  const char *ov = getenv("TSX_RCCL_LIB");
  if (ov && *ov) {
    h = dlopen(ov, RTLD_NOW | RTLD_LOCAL);
  } else {
    for (const char *n : names) {
      h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
  }
  if (!h) {
    tsx_set_error(std::string("cannot dlopen librccl: ") + dlerror());
    return TSX_ERR_COMM;
  }
#define BIND(field, sym)                                         \
  *(void **)(&g_rccl.field) = dlsym(h, sym);                     \
  if (!g_rccl.field) {                                           \
    tsx_set_error(std::string("librccl lacks ") + sym);          \
    return TSX_ERR_COMM;                                         \
  }
  BIND(GetUniqueId, "ncclGetUniqueId");
  BIND(CommInitRank, "ncclCommInitRank");
  BIND(CommDestroy, "ncclCommDestroy");
  BIND(Send, "ncclSend");
  BIND(Recv, "ncclRecv");
  BIND(AllReduce, "ncclAllReduce");
  BIND(GroupStart, "ncclGroupStart");
  BIND(GroupEnd, "ncclGroupEnd");
  BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
  *(void **)(&g_rccl.CommSplit) = dlsym(h, "ncclCommSplit");
  g_rccl.h = h;
  return TSX_OK;
}
#define NCCLCHK(call)                                                                  \
  do {                                                                                 \
    int r_ = (call);                                                                   \
    if (r_ != 0) {                                                                     \
      tsx_set_error(std::string(#call) + ": " + g_rccl.GetErrorString(r_));            \
      return TSX_ERR_COMM;                                                             \
    }                                                                                  \
  } while (0)

extern "C" int tsx_comm_unique_id(void *id128) {
  ARGCHK(id128, "tsx_comm_unique_id: null");
  int rc = rccl_load();
  if (rc) return rc;
  tsx_ncclUniqueId id;
  NCCLCHK(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return TSX_OK;
}

extern "C" int tsx_comm_init(tsx_solver *s, const void *id128) {
  ARGCHK(s && id128, "tsx_comm_init: null");
  int rc = rccl_load();
  if (rc) return rc;
  HIPCHK(hipSetDevice(s->device));
  tsx_ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  tsx_ncclComm_t comm = nullptr;
  NCCLCHK(g_rccl.CommInitRank(&comm, s->grid.nranks, id, s->grid.rank));
  s->nccl_comm = comm;
  s->comm_ready = true;
  s->pcg_key = -1;  // decisions agreed over the previous transport are agreed again (tsx_pc_global_agree)
  // The face exchanges run on comm_stream while the all-reduces run on the solver stream: give each stream a communicator
  // of its own (two streams driving one communicator concurrently is the classic RCCL hang).  TSX_RCCL_SPLIT=0 keeps one.
  const char *e = getenv("TSX_RCCL_SPLIT");
  if (g_rccl.CommSplit && !(e && atoi(e) == 0)) {
    tsx_ncclComm_t cx = nullptr;
    NCCLCHK(g_rccl.CommSplit(comm, 0, s->grid.rank, &cx, nullptr));
    s->nccl_comm_x = cx;
  }
  return TSX_OK;
}

extern "C" int tsx_comm_set_callbacks(tsx_solver *s, tsx_exchange_fn exchange, tsx_allreduce_fn allreduce, void *ctx) {
  ARGCHK(s, "tsx_comm_set_callbacks: null");
  ARGCHK((exchange == nullptr) == (allreduce == nullptr), "tsx_comm_set_callbacks: set both callbacks or neither");
  s->xchg_cb = exchange;
  s->allred_cb = allreduce;
  s->cb_ctx = ctx;
  s->pcg_key = -1;  // decisions agreed over the previous transport are agreed again (tsx_pc_global_agree)
  return TSX_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" void tsx_default_ksp_opts(tsx_ksp_opts *o) {
  if (!o) return;
  o->rtol = 1e-5;   // determine_ksp_tolerances, src/pprts_base.F90:1128
  o->atol = 1e-8;
  o->dtol = 1e4;    // PETSc KSP default divergence tolerance
  o->maxit = 1000;  // src/pprts_base.F90:1118
  o->pc = TSX_PC_REDBLACK;  // this back-end's default preconditioner
  o->pc_sweeps = 0;  // automatic (prepare_ksp): 21 (22 passes) with the scan kernels, else 9
  o->check_every = 0;  // automatic: every 2 iterations (at most one enqueued in vain), the first look where the handle's previous solve ended (krylov_run); measured 1 / 2 / 3 / 4 / 6: 19.44 / 19.24 / 19.37 / 19.68 / 19.26 ms, warm start 3.40 / 3.47 / 3.58 / 3.69 / 3.82 ms
  o->fp32_directions = 2;
  o->pc_coeff_fp16 = 1;
  o->skip_complete_initial_run = 0;
  o->explicit_solver = 0;
  o->accept_incomplete_solve = 0;
  o->initial_guess_zero = 0;
}

extern "C" int tsx_determine_ksp_tolerances(const tsx_solver *s, double unconstrained_fraction, double *rtol,
                                            double *atol, int32_t *maxit) {
  ARGCHK(s && rtol && atol && maxit, "tsx_determine_ksp_tolerances: null");
  // src/pprts_base.F90:1126-1131 with C = C_diff: glob_zm = Nz + 1
  *maxit = 1000;
  *rtol = 1e-5;
  if (unconstrained_fraction < 0.0)  // the solver's own count of 1-D layers (after tsx_pprts_set_optical_properties)
    unconstrained_fraction = s->geo.Nz > 0 ? 1.0 - (double)s->n1d / (double)s->geo.Nz : 1.0;
  double a = 1e-4 * (double)s->grid.glob_xm * (double)s->grid.glob_ym * (double)(s->grid.Nz + 1) * unconstrained_fraction;
  *atol = a > 1e-8 ? a : 1e-8;
  return TSX_OK;
}

static int create_fill(tsx_solver *s, const tsx_grid *grid);
static void slots_free(tsx_solver *s);
static void tsx_log_free(tsx_solver *s);
extern "C" int tsx_create(const tsx_grid *grid, tsx_solver **out) {
  ARGCHK(grid && out, "tsx_create: null argument");
  ARGCHK(grid->solver_id == TSX_SOLVER_3_10 || grid->solver_id == TSX_SOLVER_8_16,
         "tsx_create: solver_id must be 310 (3_10) or 816 (8_16)");
  ARGCHK(grid->Nz >= 1 && grid->xm >= 1 && grid->ym >= 1, "tsx_create: empty grid");
  ARGCHK(grid->nranks >= 1 && grid->rank >= 0 && grid->rank < grid->nranks, "tsx_create: bad rank/nranks");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    tsx_set_error("tsx_create: no HIP device available (libtsx has no CPU fallback)");
    return TSX_ERR_NO_DEVICE;
  }
  tsx_solver *s = new tsx_solver();  // value-initialisation: every member zero, then the default member initialisers of tsx_internal.hpp
  const int rc = create_fill(s, grid);
  if (rc) {  // release the half-built solver (streams, events, buffers) on any failure
    (void)tsx_destroy(s);
    return rc;
  }
  if (const char *e = getenv("TSX_LOG"))
    if (atoi(e) != 0) s->log = new TsxLog();
  *out = s;
  return TSX_OK;
}

static int create_fill(tsx_solver *s, const tsx_grid *grid) {
  s->grid = *grid;
  if (grid->device >= 0) s->device = grid->device;
  else HIPCHK(hipGetDevice(&s->device));
  HIPCHK(hipSetDevice(s->device));

  TsxGeo &g = s->geo;
  g.Nz = grid->Nz;
  g.xm = grid->xm;
  g.ym = grid->ym;
  g.ncol = g.xm * g.ym;
  g.Nc = (long long)g.Nz * g.ncol;
  g.ntop = grid->solver_id == TSX_SOLVER_3_10 ? 2 : 8;
  g.nside = 4;
  g.D = g.ntop + 2 * g.nside;
  g.N = (long long)g.D * (g.Nc + g.ncol);
  const bool self_x = grid->nranks == 1 || (grid->neigh_w == grid->rank && grid->neigh_e == grid->rank);
  const bool self_y = grid->nranks == 1 || (grid->neigh_s == grid->rank && grid->neigh_n == grid->rank);
  g.wrap_x = self_x && !grid->force_halo;
  g.pc_tile_x = g.pc_tile_y = 0;
  if (const char *e = getenv("TSX_PC_TILE")) sscanf(e, "%d,%d", &g.pc_tile_x, &g.pc_tile_y);
  g.wrap_y = self_y && !grid->force_halo;

  HIPCHK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  s->own_stream = true;
  HIPCHK(hipEventCreate(&s->ev0));
  HIPCHK(hipEventCreate(&s->ev1));
  HIPCHK(hipEventCreate(&s->ev_imp0));
  HIPCHK(hipEventCreate(&s->ev_imp1));
  HIPCHK(hipEventCreate(&s->ev_exp1));
  HIPCHK(hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking));
  HIPCHK(hipDeviceGetAttribute(&s->max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, s->device));
  HIPCHK(hipEventCreateWithFlags(&s->ev_pack, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&s->ev_recv, hipEventDisableTiming));
  {
    const char *e = getenv("TSX_OVERLAP");
    s->overlap_env = e ? (atoi(e) != 0 ? 1 : 0) : -1;
  }

  const size_t nb = (size_t)g.N * sizeof(double);
  // what this solver will hold after its first solve, roughly: 8 Krylov vectors here, then 4-5 more for the preconditioned
  // directions and staging, the fp32 copies, the coefficient planes (fp32) with their packed copy, the pipeline's fields -- one slab
  // for all of it instead of a quarantine per buffer (tsx_pool.hip)
  tsx_dev_reserve((size_t)15 * nb + (size_t)g.Nc * g.D * g.D * 4 * 3 / 2 + (size_t)g.Nc * 16 * (g.ntop == 2 ? 10 : 36) + ((size_t)32 << 20));
  double **vecs[] = {&s->vx, &s->vb, &s->vr, &s->vrhat, &s->vp, &s->vv, &s->vs, &s->vt};
  for (double **v : vecs) {
    HIPCHK(tsx_dev_malloc((void **)v, nb));
    HIPCHK(hipMemsetAsync(*v, 0, nb, s->stream));
  }
  s->vph = s->vp;  // no preconditioner: p-hat aliases p
  s->vsh = s->vs;
  s->halo_x_elems = (size_t)(g.nside / 2) * g.Nz * g.ym;
  s->halo_y_elems = (size_t)(g.nside / 2) * g.Nz * g.xm;
  double **hx[] = {&s->sendW, &s->sendE, &s->recvW, &s->recvE};
  double **hy[] = {&s->sendS, &s->sendN, &s->recvS, &s->recvN};
  for (double **v : hx) {
    HIPCHK(tsx_dev_malloc((void **)v, s->halo_x_elems * sizeof(double)));
    HIPCHK(hipMemsetAsync(*v, 0, s->halo_x_elems * sizeof(double), s->stream));
  }
  for (double **v : hy) {
    HIPCHK(tsx_dev_malloc((void **)v, s->halo_y_elems * sizeof(double)));
    HIPCHK(hipMemsetAsync(*v, 0, s->halo_y_elems * sizeof(double), s->stream));
  }
  HIPCHK(tsx_dev_malloc((void **)&s->partials, sizeof(double) * TSX_NSLOTS * TSX_MAX_PARTIAL_BLOCKS));
  HIPCHK(hipMemsetAsync(s->partials, 0, sizeof(double) * TSX_NSLOTS * TSX_MAX_PARTIAL_BLOCKS, s->stream));
  HIPCHK(tsx_dev_malloc((void **)&s->scal, sizeof(TsxScalars)));
  HIPCHK(hipMemsetAsync(s->scal, 0, sizeof(TsxScalars), s->stream));
  HIPCHK(hipHostMalloc((void **)&s->scal_host, sizeof(TsxScalars), hipHostMallocDefault));
  HIPCHK(tsx_dev_malloc((void **)&s->l1d, (size_t)g.Nz));
  HIPCHK(tsx_dev_malloc((void **)&s->albedo, sizeof(double) * g.ncol));
  HIPCHK(hipStreamSynchronize(s->stream));
  return TSX_OK;
}

extern "C" int tsx_destroy(tsx_solver *s) {
  if (!s) return TSX_OK;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  void *ptrs[] = {s->v32, s->p32, s->dsend[0], s->dsend[1], s->dsend[2], s->dsend[3], s->drecv[0], s->drecv[1], s->drecv[2], s->drecv[3],
                  s->coef_h, s->coef, s->dd_coef, s->dd_cidx, s->dd_cidx_split, s->dd_ent_cell, s->dd_scratch, s->pcn_coef, s->pcn_cidx_split, s->pcn_ent_cell, s->pcr_idx, s->pcr_ent, s->pcr_tab, s->pch_send[0], s->pch_send[1], s->pch_send[2], s->pch_send[3],
                  s->pch_recv[0], s->pch_recv[1], s->pch_recv[2], s->pch_recv[3], s->l1d,   s->a11,   s->a12,   s->albedo, s->vx,    s->vb,    s->vr,      s->vrhat, s->vp,
                  s->vv,    s->vs,    s->vt,    s->stage_a, s->stage_b, s->sendW, s->sendE, s->sendS, s->sendN, s->recvW,
                  s->recvE, s->recvS, s->recvN, s->partials, s->scal, s->vw, s->pc_tmp, s->lut_diff.d_axes, s->lut_diff.d_table,
                  s->lut_T.d_axes, s->lut_T.d_table, s->lut_S.d_axes, s->lut_S.d_table, s->dirT, s->dirS, s->d_kabs, s->d_ksca,
                  s->d_g, s->d_dz, s->a13, s->a23, s->a33, s->planck, s->bsrfc, s->edir_a, s->edir_b, s->dsc, s->abso, s->cell_samp, s->dd_colsum, s->pcx_rec, s->pcx_vz, s->flow_state, s->flow_prog, s->flow_zb8, s->flow_pr_dev};
  for (void *p : ptrs)
    if (p) (void)tsx_dev_free(p);
  if (s->vph && s->vph != s->vp) (void)tsx_dev_free(s->vph);
  if (s->vsh && s->vsh != s->vs) (void)tsx_dev_free(s->vsh);
  if (s->scal_host) (void)hipHostFree(s->scal_host);
  if (s->dsc_host) (void)hipHostFree(s->dsc_host);
  for (int q = 0; q < 4; ++q) {
    if (s->host_send[q]) (void)hipHostFree(s->host_send[q]);
    if (s->host_recv[q]) (void)hipHostFree(s->host_recv[q]);
  }
  slots_free(s);
  tsx_log_free(s);
  delete s->flow_pr_shadow;
  if (s->comm_ready && g_rccl.CommDestroy) {
    if (s->nccl_comm_x) g_rccl.CommDestroy(s->nccl_comm_x);
    g_rccl.CommDestroy(s->nccl_comm);
  }
  tsx_peer_destroy(s);
  for (hipEvent_t e : {s->ev0, s->ev1, s->ev_imp0, s->ev_imp1, s->ev_exp1, s->ev_pack, s->ev_recv})
    if (e) (void)hipEventDestroy(e);
  if (s->comm_stream) (void)hipStreamDestroy(s->comm_stream);
  if (s->own_stream && s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
  return TSX_OK;
}

extern "C" int tsx_set_stream(tsx_solver *s, void *hip_stream) {
  ARGCHK(s, "tsx_set_stream: null");
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  if (s->own_stream) HIPCHK(hipStreamDestroy(s->stream));
  if (hip_stream) {
    s->stream = (hipStream_t)hip_stream;
    s->own_stream = false;
  } else {
    HIPCHK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    s->own_stream = true;
  }
  return TSX_OK;
}

// ------------------------------------------------------------------------------------------------
// exchange the four face buffers with the W/E/S/N neighbours.  recvW <- west's sendE, recvE <- east's
// sendW, recvS <- south's sendN, recvN <- north's sendS.  Point-to-point messages to one peer are
// matched in issue order, so receives are posted E,W,N,S against sends W,E,S,N (matters when both
// x-neighbours are the same rank, e.g. 2 ranks along a periodic axis).
// Exchange of four face buffers with the W, E, S, N neighbours (what I send W-ward lands in my west neighbour's
// east buffer ...).  st: the stream the transfers are issued on (the solver stream, or comm_stream when overlapping;
// the caller has made st wait for the pack kernel).  send/recv: device buffers in the order W, E, S, N; cx / cy:
// doubles per x / y face message.  Used for the diffuse halo (tsx_face_exchange) and the direct beam's.
int tsx_face_exchange_bufs(tsx_solver *s, hipStream_t st, double *const send[4], double *const recv[4], size_t cx, size_t cy) {
  const TsxGeo &g = s->geo;
  const size_t bx = cx, by = cy;
  if (tsx_peer_ready(s)) return tsx_peer_exchange(s, st, send, recv, cx, cy, nullptr);
  if (s->xchg_cb) {
    const tsx_grid &gr = s->grid;
    const size_t count[4] = {g.wrap_x ? 0 : bx, g.wrap_x ? 0 : bx, g.wrap_y ? 0 : by, g.wrap_y ? 0 : by};
    const int peer[4] = {gr.neigh_w, gr.neigh_e, gr.neigh_s, gr.neigh_n};
    for (int q = 0; q < 4; ++q) {
      const size_t cap = (q < 2 ? s->halo_x_elems : s->halo_y_elems) * sizeof(double);
      if (count[q] * sizeof(double) > cap) {
        tsx_set_error("face_exchange: message larger than the staging buffers");
        return TSX_ERR_ARG;
      }
      if (!s->host_send[q]) HIPCHK(hipHostMalloc((void **)&s->host_send[q], cap, hipHostMallocDefault));
      if (!s->host_recv[q]) HIPCHK(hipHostMalloc((void **)&s->host_recv[q], cap, hipHostMallocDefault));
      if (count[q]) HIPCHK(hipMemcpyAsync(s->host_send[q], send[q], count[q] * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    if (s->xchg_cb(s->cb_ctx, (const double *const *)s->host_send, (double *const *)s->host_recv, count, peer)) {
      tsx_set_error("face_exchange: exchange callback failed");
      return TSX_ERR_COMM;
    }
    for (int q = 0; q < 4; ++q)
      if (count[q]) HIPCHK(hipMemcpyAsync(recv[q], s->host_recv[q], count[q] * sizeof(double), hipMemcpyHostToDevice, st));
    return TSX_OK;
  }
  if (s->comm_ready) {
    tsx_ncclComm_t c = s->nccl_comm_x ? s->nccl_comm_x : s->nccl_comm;
    const tsx_grid &gr = s->grid;
    NCCLCHK(g_rccl.GroupStart());
    if (!g.wrap_x) {
      NCCLCHK(g_rccl.Send(send[0], bx, TSX_NCCL_FLOAT64, gr.neigh_w, c, st));
      NCCLCHK(g_rccl.Send(send[1], bx, TSX_NCCL_FLOAT64, gr.neigh_e, c, st));
      NCCLCHK(g_rccl.Recv(recv[1], bx, TSX_NCCL_FLOAT64, gr.neigh_e, c, st));
      NCCLCHK(g_rccl.Recv(recv[0], bx, TSX_NCCL_FLOAT64, gr.neigh_w, c, st));
    }
    if (!g.wrap_y) {
      NCCLCHK(g_rccl.Send(send[2], by, TSX_NCCL_FLOAT64, gr.neigh_s, c, st));
      NCCLCHK(g_rccl.Send(send[3], by, TSX_NCCL_FLOAT64, gr.neigh_n, c, st));
      NCCLCHK(g_rccl.Recv(recv[3], by, TSX_NCCL_FLOAT64, gr.neigh_n, c, st));
      NCCLCHK(g_rccl.Recv(recv[2], by, TSX_NCCL_FLOAT64, gr.neigh_s, c, st));
    }
    NCCLCHK(g_rccl.GroupEnd());
    return TSX_OK;
  }
  if (s->grid.nranks > 1) {
    tsx_set_error("face_exchange: nranks > 1 but neither tsx_comm_init nor tsx_comm_set_callbacks was called");
    return TSX_ERR_STATE;
  }
  // single rank with force_halo: every neighbour is this rank
  if (!g.wrap_x) {
    HIPCHK(hipMemcpyAsync(recv[1], send[0], bx * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(recv[0], send[1], bx * sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  if (!g.wrap_y) {
    HIPCHK(hipMemcpyAsync(recv[3], send[2], by * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(recv[2], send[3], by * sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  return TSX_OK;
}

// the diffuse halo: entering side streams (exchange_diffuse_boundary's traffic)
int tsx_face_exchange(tsx_solver *s, hipStream_t st) {
  double *const send[4] = {s->sendW, s->sendE, s->sendS, s->sendN};
  double *const recv[4] = {s->recvW, s->recvE, s->recvS, s->recvN};
  return tsx_face_exchange_bufs(s, st, send, recv, s->halo_x_elems, s->halo_y_elems);
}

int tsx_face_exchange_elems(tsx_solver *s, hipStream_t st, size_t elem_bytes) {
  double *const send[4] = {s->sendW, s->sendE, s->sendS, s->sendN};
  double *const recv[4] = {s->recvW, s->recvE, s->recvS, s->recvN};
  return tsx_face_exchange_bufs(s, st, send, recv, (s->halo_x_elems * elem_bytes + 7) / 8, (s->halo_y_elems * elem_bytes + 7) / 8);
}

// reduce partials -> (all-reduce) -> scalar algebra
static int scalar_stage(tsx_solver *s, int nblocks, int nslots, int stage) {
  TsxPeerArArgs noar;
  memset((void *)&noar, 0, sizeof(noar));
  if (tsx_peer_ready(s) && s->grid.nranks > 1) {  // ONE kernel on the solver stream: partial sums, the sum over the ranks through
    TsxPeerArArgs ar;                              // the mailboxes, the scalar algebra; no library call, no host
    int rc = tsx_peer_ar_args(s, TSX_NSLOTS, &ar);
    if (rc) return rc;
    hipLaunchKernelGGL(tsx_k_scalar, dim3(1), dim3(1024), 0, s->stream, s->scal, s->partials, nblocks, nslots, stage, 3, ar);
    HIPCHK(hipGetLastError());
    return TSX_OK;
  }
  if (s->allred_cb) {
    hipLaunchKernelGGL(tsx_k_scalar, dim3(1), dim3(1024), 0, s->stream, s->scal, s->partials, nblocks, nslots, stage, 1, noar);
    HIPCHK(hipMemcpyAsync(s->scal_host->red, s->scal->red, sizeof(double) * TSX_NSLOTS, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->allred_cb(s->cb_ctx, s->scal_host->red, TSX_NSLOTS)) {
      tsx_set_error("scalar_stage: allreduce callback failed");
      return TSX_ERR_COMM;
    }
    HIPCHK(hipMemcpyAsync(s->scal->red, s->scal_host->red, sizeof(double) * TSX_NSLOTS, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(tsx_k_scalar, dim3(1), dim3(1024), 0, s->stream, s->scal, s->partials, nblocks, nslots, stage, 2, noar);
    HIPCHK(hipGetLastError());
    return TSX_OK;
  }
  if (s->comm_ready) {  // also with a 1-rank communicator (exercised by the single-GPU RCCL test)
    hipLaunchKernelGGL(tsx_k_scalar, dim3(1), dim3(1024), 0, s->stream, s->scal, s->partials, nblocks, nslots, stage, 1, noar);
    NCCLCHK(g_rccl.AllReduce(s->scal->red, s->scal->red, TSX_NSLOTS, TSX_NCCL_FLOAT64, TSX_NCCL_SUM, s->nccl_comm,
                             s->stream));
    hipLaunchKernelGGL(tsx_k_scalar, dim3(1), dim3(1024), 0, s->stream, s->scal, s->partials, nblocks, nslots, stage, 2, noar);
  } else {
    hipLaunchKernelGGL(tsx_k_scalar, dim3(1), dim3(1024), 0, s->stream, s->scal, s->partials, nblocks, nslots, stage, 3, noar);
  }
  HIPCHK(hipGetLastError());
  return TSX_OK;
}

// ------------------------------------------------------------------------------------------------
// workgroups of the layout conversion: one per tile of TSX_CV_TI columns x TSX_CV_TK levels of one row (grid-stride above 2^20)
static int convert_grid(const TsxGeo &g) {
  const long long nt = (long long)((g.xm + TSX_CV_TI - 1) / TSX_CV_TI) * g.ym * ((g.Nz + 1 + TSX_CV_TK - 1) / TSX_CV_TK);
  return (int)(nt < (1ll << 20) ? nt : (1ll << 20));
}

template <int NTOP, int NSIDE>
static int import_vec(tsx_solver *s, const double *ref_dev, double *v, int *nzflag = nullptr) {
  const TsxGeo &g = s->geo;
  hipLaunchKernelGGL((tsx_k_convert_vec<NTOP, NSIDE, false>), dim3(convert_grid(g)), dim3(TSX_BLOCK), 0, s->stream, g,
                     const_cast<double *>(ref_dev), v, s->sendW, s->sendS, nzflag);
  HIPCHK(hipGetLastError());
  if (!(g.wrap_x && g.wrap_y)) {
    // only the W-ward / S-ward messages carry data; E/N-ward buffers travel as they are (ignored)
    int rc = tsx_face_exchange(s, s->stream);
    if (rc) return rc;
    const long long n = (long long)s->halo_x_elems + (long long)s->halo_y_elems;
    // a direction that wraps in-kernel has nothing to unpack: pass through harmlessly by guarding in host
    if (!g.wrap_x || !g.wrap_y) {
      TsxGeo g2 = g;
      hipLaunchKernelGGL((tsx_k_import_unpack<NTOP, NSIDE>), dim3(grid_for(n)), dim3(TSX_BLOCK), 0, s->stream, g2, v,
                         g.wrap_x ? (const double *)nullptr : s->recvE, g.wrap_y ? (const double *)nullptr : s->recvN);
      HIPCHK(hipGetLastError());
    }
  }
  return TSX_OK;
}

template <int NTOP, int NSIDE>
static int export_vec(tsx_solver *s, const double *v, double *ref_dev) {
  const TsxGeo &g = s->geo;
  int rc = halo_update<NTOP, NSIDE>(s, v, false);
  if (rc) return rc;
  hipLaunchKernelGGL((tsx_k_convert_vec<NTOP, NSIDE, true>), dim3(convert_grid(g)), dim3(TSX_BLOCK), 0, s->stream, g,
                     ref_dev, const_cast<double *>(v), s->recvW, s->recvS, (int *)nullptr);
  HIPCHK(hipGetLastError());
  return TSX_OK;
}

static int ensure_stage(tsx_solver *s) {
  const size_t nb = (size_t)s->geo.N * sizeof(double);
  if (!s->stage_a) HIPCHK(tsx_dev_malloc((void **)&s->stage_a, nb));
  if (!s->stage_b) HIPCHK(tsx_dev_malloc((void **)&s->stage_b, nb));
  return TSX_OK;
}

// ------------------------------------------------------------------------------------------------
// l1d / a11 / a12 / albedo: shared by set_coeffs and set_optprop
static int set_aux(tsx_solver *s, const uint8_t *l1d, const double *a11, const double *a12, const double *albedo, int where) {
  const TsxGeo &g = s->geo;
  const hipMemcpyKind mk = where == TSX_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  std::vector<uint8_t> l1d_h(g.Nz);
  if (where == TSX_HOST) memcpy(l1d_h.data(), l1d, g.Nz);
  else HIPCHK(hipMemcpy(l1d_h.data(), l1d, g.Nz, hipMemcpyDeviceToHost));
  s->any_l1d = false;
  s->n1d = 0;
  for (int k = 0; k < g.Nz; ++k) {
    s->any_l1d |= l1d_h[k] != 0;
    s->n1d += l1d_h[k] != 0;
  }
  HIPCHK(hipMemcpyAsync(s->l1d, l1d_h.data(), g.Nz, hipMemcpyHostToDevice, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  ARGCHK(!s->any_l1d || (a11 && a12), "a11/a12 required when any layer is 1-D");
  HIPCHK(hipMemcpyAsync(s->albedo, albedo, sizeof(double) * g.ncol, mk, s->stream));
  s->have_albedo = true;
  if (s->any_l1d) {
    if (!s->a11) HIPCHK(tsx_dev_malloc((void **)&s->a11, sizeof(double) * g.Nc));
    if (!s->a12) HIPCHK(tsx_dev_malloc((void **)&s->a12, sizeof(double) * g.Nc));
    TsxDevTmp g11, g12;
    double *t11 = nullptr, *t12 = nullptr;
    const double *p11 = a11, *p12 = a12;
    if (where == TSX_HOST) {
      HIPCHK(g11.alloc(sizeof(double) * g.Nc));
      HIPCHK(g12.alloc(sizeof(double) * g.Nc));
      t11 = g11.as<double>();
      t12 = g12.as<double>();
      HIPCHK(hipMemcpyAsync(t11, a11, sizeof(double) * g.Nc, hipMemcpyHostToDevice, s->stream));
      HIPCHK(hipMemcpyAsync(t12, a12, sizeof(double) * g.Nc, hipMemcpyHostToDevice, s->stream));
      p11 = t11;
      p12 = t12;
    }
    hipLaunchKernelGGL(tsx_k_import_cellfield, dim3(grid_for(g.Nc)), dim3(TSX_BLOCK), 0, s->stream, g, p11, s->a11);
    hipLaunchKernelGGL(tsx_k_import_cellfield, dim3(grid_for(g.Nc)), dim3(TSX_BLOCK), 0, s->stream, g, p12, s->a12);
    HIPCHK(hipStreamSynchronize(s->stream));
  }
  HIPCHK(hipStreamSynchronize(s->stream));
  return TSX_OK;
}

static int ensure_coef_storage(tsx_solver *s, int out_bytes) {
  const size_t ncoef = (size_t)s->geo.D * s->geo.D * s->geo.Nc;
  if (s->coef && s->coef_bytes != out_bytes) {
    HIPCHK(tsx_dev_free(s->coef));
    s->coef = nullptr;
  }
  if (!s->coef) HIPCHK(tsx_dev_malloc(&s->coef, ncoef * out_bytes));
  s->coef_bytes = out_bytes;
  return TSX_OK;
}

extern "C" int tsx_diff_set_coeffs(tsx_solver *s, const void *diff2diff, int coeff_kind, const uint8_t *l1d,
                                   const double *a11, const double *a12, const double *albedo, int where) {
  ARGCHK(s && diff2diff && l1d && albedo, "tsx_diff_set_coeffs: null argument");
  ARGCHK(coeff_kind == 4 || coeff_kind == 8, "tsx_diff_set_coeffs: coeff_kind must be 4 or 8");
  HIPCHK(hipSetDevice(s->device));
  const TsxGeo &g = s->geo;
  const int DD = g.D * g.D;
  const size_t ncoef = (size_t)DD * g.Nc;
  int rc = set_aux(s, l1d, a11, a12, albedo, where);
  if (rc) return rc;

  const void *src_dev = diff2diff;
  TsxDevTmp tmp, flag_guard;  // released on every exit path (one call per g-point: a leak here is 3.4 GB per call)
  if (where == TSX_HOST) {
    HIPCHK(tmp.alloc(ncoef * coeff_kind));
    HIPCHK(hipMemcpyAsync(tmp.p, diff2diff, ncoef * coeff_kind, hipMemcpyHostToDevice, s->stream));
    src_dev = tmp.p;
  }
  int out_bytes = 4;
  if (coeff_kind == 8) {  // keep fp64 unless every value survives the round trip through fp32
    HIPCHK(flag_guard.alloc(sizeof(int)));
    int *flag = flag_guard.as<int>();
    HIPCHK(hipMemsetAsync(flag, 0, sizeof(int), s->stream));
    hipLaunchKernelGGL(tsx_k_check_fp32_lossless, dim3(grid_for((long long)ncoef)), dim3(TSX_BLOCK), 0, s->stream,
                       (long long)ncoef, (const double *)src_dev, flag);
    int bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    out_bytes = bad ? 8 : 4;
  }
  if ((rc = ensure_coef_storage(s, out_bytes))) return rc;
  const int TI = DD > 128 ? 16 : 32;  // keep the LDS tile under 64 KiB for D = 16
  const int nbk = grid_for((long long)((g.xm + TI - 1) / TI) * g.ym * g.Nz * TSX_BLOCK, 8192);
  const size_t lds = (size_t)TI * (DD + 1) * out_bytes;
  if (coeff_kind == 8 && out_bytes == 8)
    hipLaunchKernelGGL((tsx_k_import_coeff<double, double>), dim3(nbk), dim3(TSX_BLOCK), lds, s->stream, g, DD, TI,
                       (const double *)src_dev, (double *)s->coef);
  else if (coeff_kind == 8)
    hipLaunchKernelGGL((tsx_k_import_coeff<double, float>), dim3(nbk), dim3(TSX_BLOCK), lds, s->stream, g, DD, TI,
                       (const double *)src_dev, (float *)s->coef);
  else
    hipLaunchKernelGGL((tsx_k_import_coeff<float, float>), dim3(nbk), dim3(TSX_BLOCK), lds, s->stream, g, DD, TI,
                       (const float *)src_dev, (float *)s->coef);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s->stream));
  s->have_coeffs = true;
  s->coef_h_valid = false;
  s->pcx_valid = false;
  s->dd_valid = false;
  s->dd_on = false;
  s->dd_pc = false;
  s->dd_from_coords = false;
  s->dd_hash_ready = false;  // imported blocks: tsx_dedup.hip hashes them itself
  s->coef_dense_valid = true;
  return TSX_OK;
}

// ------------------------------------------------------------------------------------------------
// LUT on the device
// axes of LUT_3_10 / LUT_8_16 diffuse tables (src/optprop_base.F90:200-212, 228-240): presets from
// src/optprop_parameters.F90:145-154 (tau31), :194-199 (w020), :107-110 (aspect23), :245 (g6)
static const float k_preset_tau31[31] = {
    1e-10f, 3.62266272998e-07f, 7.04565803675e-06f, 4.47545500233e-05f, 0.000172126759821f, 0.000495994753047f,
    0.00119161313679f, 0.00251026980343f, 0.00480799264297f, 0.00856221891924f, 0.0143961482731f, 0.0231530284254f,
    0.0358868239775f, 0.0541358315379f, 0.079959118223f, 0.11623968405f, 0.167882053841f, 0.246414427244f,
    0.350199325489f, 0.502459974196f, 0.759082408765f, 1.08083180518f, 1.5415157991f, 2.19832932733f, 3.04549626819f,
    4.27145477454f, 6.16953841432f, 9.43719309835f, 15.7335501106f, 29.5819342206f, 100.0f};
static const float k_preset_w020[20] = {
    0.0f, 0.152960717624f, 0.295085090042f, 0.416951893959f, 0.521358613652f, 0.610087211908f, 0.684967634054f,
    0.747886390181f, 0.800286677013f, 0.84336972609f, 0.878674797098f, 0.906377786525f, 0.928097831502f,
    0.943463164595f, 0.954135786554f, 0.963824066888f, 0.972632134967f, 0.981529289348f, 0.990759644674f, 0.99999f};
static const float k_preset_aspect23[23] = {0.02f, 0.032f, 0.042f, 0.056f, 0.075f, 0.1f, 0.133f, 0.178f, 0.237f, 0.316f,
                                            0.422f, 0.562f, 0.75f, 1.f, 1.25f, 1.562f, 1.953f, 2.441f, 3.052f, 3.815f,
                                            4.768f, 5.96f, 7.451f};
static const float k_preset_g6[6] = {0.0f, 0.2424f, 0.4137f, 0.5717f, 0.7144f, 0.85f};

extern "C" int tsx_lut_set_diffuse(tsx_solver *s, const float *table, int32_t nvec, int64_t nentries, int32_t ndim,
                                   const int32_t *n, const float *axes_concat, int where) {
  ARGCHK(s && table && n && axes_concat, "tsx_lut_set_diffuse: null argument");
  ARGCHK(ndim == 4, "tsx_lut_set_diffuse: diffuse tables have 4 dimensions (tau, w0, aspect_zx, g)");
  ARGCHK(nvec == s->geo.D * s->geo.D, "tsx_lut_set_diffuse: nvec must be D*D");
  long long prod = 1, nax = 0;
  for (int d = 0; d < ndim; ++d) {
    ARGCHK(n[d] >= 1, "tsx_lut_set_diffuse: empty axis");
    prod *= n[d];
    nax += n[d];
  }
  ARGCHK(prod == nentries, "tsx_lut_set_diffuse: nentries != product of axis lengths");
  HIPCHK(hipSetDevice(s->device));
  TsxLutHost &L = s->lut_diff;
  if (L.d_axes) HIPCHK(tsx_dev_free(L.d_axes));
  if (L.d_table) HIPCHK(tsx_dev_free(L.d_table));
  L = TsxLutHost();
  HIPCHK(tsx_dev_malloc((void **)&L.d_axes, sizeof(float) * nax));
  HIPCHK(tsx_dev_malloc((void **)&L.d_table, sizeof(float) * (size_t)nvec * nentries));
  const hipMemcpyKind mk = where == TSX_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  HIPCHK(hipMemcpy(L.d_axes, axes_concat, sizeof(float) * nax, mk));
  HIPCHK(hipMemcpy(L.d_table, table, sizeof(float) * (size_t)nvec * nentries, mk));
  L.ndim = ndim;
  L.nvec = nvec;
  L.nentries = nentries;
  for (int d = 0; d < ndim; ++d) L.n[d] = n[d];
  L.ready = true;
  return TSX_OK;
}

extern "C" int tsx_lut_load_diffuse_mmap4(tsx_solver *s, const char *path) {
  ARGCHK(s && path, "tsx_lut_load_diffuse_mmap4: null argument");
  // src/mmap.F90:129-203: header = one page of size_t, data starts at the page boundary
  const long pagesize = sysconf(_SC_PAGESIZE);
  int fd = open(path, O_RDONLY);
  if (fd < 0) {
    tsx_set_error(std::string("tsx_lut_load_diffuse_mmap4: cannot open ") + path);
    return TSX_ERR_ARG;
  }
  std::vector<size_t> header((size_t)pagesize / sizeof(size_t));
  if (read(fd, header.data(), (size_t)pagesize) != pagesize) {
    close(fd);
    tsx_set_error("tsx_lut_load_diffuse_mmap4: short header");
    return TSX_ERR_ARG;
  }
  const size_t dtype_size = header[0], n_elems = header[1], n_bytes = header[2], dim1 = header[3], dim2 = header[4];
  if (dtype_size != 4 || n_bytes != 4 * n_elems || dim1 * dim2 != n_elems || header[5] != 0) {
    close(fd);
    tsx_set_error("tsx_lut_load_diffuse_mmap4: not a 2-D real32 mmap4 table");
    return TSX_ERR_ARG;
  }
  void *m = mmap(nullptr, n_bytes + (size_t)pagesize, PROT_READ, MAP_PRIVATE | MAP_NORESERVE, fd, 0);
  close(fd);
  if (m == MAP_FAILED) {
    tsx_set_error("tsx_lut_load_diffuse_mmap4: mmap failed");
    return TSX_ERR_ARG;
  }
  const int32_t n[4] = {31, 20, 23, 6};
  std::vector<float> axes;
  axes.insert(axes.end(), k_preset_tau31, k_preset_tau31 + 31);
  axes.insert(axes.end(), k_preset_w020, k_preset_w020 + 20);
  axes.insert(axes.end(), k_preset_aspect23, k_preset_aspect23 + 23);
  axes.insert(axes.end(), k_preset_g6, k_preset_g6 + 6);
  int rc = tsx_lut_set_diffuse(s, (const float *)((const char *)m + pagesize), (int32_t)dim1, (int64_t)dim2, 4, n,
                               axes.data(), TSX_HOST);
  munmap(m, n_bytes + (size_t)pagesize);
  return rc;
}

// the cells' LUT coordinates in cell order (tsx_k_cell_samples) -> s->cell_samp; TSX_CELL_SAMPLES=0: the coefficient kernels read
// the level-fastest arrays themselves
int tsx_cell_samples(tsx_solver *s, const double *kabs, const double *ksca, const double *g, const double *dz, double dx) {
  const TsxGeo &gm = s->geo;
  const char *e = getenv("TSX_CELL_SAMPLES");
  if (e && atoi(e) == 0) {
    if (s->cell_samp) (void)tsx_dev_free(s->cell_samp);
    s->cell_samp = nullptr;
    return TSX_OK;
  }
  if (!s->cell_samp) HIPCHK(tsx_dev_malloc(&s->cell_samp, sizeof(float4) * (size_t)gm.Nc));
  s->cell_samp_src[0] = kabs, s->cell_samp_src[1] = ksca, s->cell_samp_src[2] = g, s->cell_samp_src[3] = dz;
  s->cell_samp_dx = dx;
  hipLaunchKernelGGL(tsx_k_cell_samples, dim3((gm.ncol + 31) / 32, (gm.Nz + 31) / 32), dim3(TSX_BLOCK), 0, s->stream, gm, kabs, ksca, g, dz,
                     dx, (float4 *)s->cell_samp);
  HIPCHK(hipGetLastError());
  return TSX_OK;
}

// alloc_coeff_diff2diff on the device: kabs/ksca/g/dz are device pointers in the reference layout
static int lut_diffuse_launch(tsx_solver *s, const double *kabs, const double *ksca, const double *g, const double *dz, double dx) {
  TsxLogScope log_scope(s, TSX_EV_GET_COEFF_DIFF2DIFF);  // get_coeff_diff2diff, src/pprts.F90:3422-3489
  const TsxGeo &gm = s->geo;
  // a new coefficient set: whatever the shared storage held is gone (the callers used to reset these after the launch; the
  // coordinate-keyed build below sets them itself)
  s->dd_valid = false;
  s->dd_on = false;
  s->dd_pc = false;
  s->coef_dense_valid = true;
  TsxLutDev L;
  memset(&L, 0, sizeof(L));
  const TsxLutHost &H = s->lut_diff;
  L.ndim = H.ndim;
  L.nvec = H.nvec;
  long long off = 1;
  int aoff = 0;
  for (int d = 0; d < H.ndim; ++d) {
    L.n[d] = H.n[d];
    L.axis_off[d] = aoff;
    aoff += H.n[d];
    L.offs[d] = off;
    off *= H.n[d];
  }
  L.axes = H.d_axes;
  L.table = H.d_table;
  int rcs = tsx_cell_samples(s, kabs, ksca, g, dz, dx);
  if (rcs) return rcs;
  // sharing keyed on the cells' LUT coordinates, before anything is interpolated (tsx_dedup.hip "coordinates first"): where
  // it pays only the distinct tuples are interpolated, straight into the shared storage, and no dense planes are written
  {
    bool built = false;
    int rc = tsx_dedup_from_coords(s, L, &built);
    if (rc) return rc;
    if (built) return TSX_OK;
  }
  unsigned long long *hash = nullptr;  // the kernel leaves the blocks' hashes for the shared storage (tsx_dedup.hip)
  {
    int rc = tsx_dedup_hash_buffer(s, &hash);
    if (rc) return rc;
  }
  const int nbk = grid_for(gm.Nc, 8192);
  const float4 *samp = (const float4 *)s->cell_samp;
  if (gm.D == 10)
    hipLaunchKernelGGL((tsx_k_lut_diff2diff<100>), dim3(nbk), dim3(TSX_BLOCK), 0, s->stream, gm, L, kabs, ksca, g, dz, dx,
                       s->l1d, (float *)s->coef, hash, samp);
  else
    hipLaunchKernelGGL((tsx_k_lut_diff2diff<256>), dim3(nbk), dim3(TSX_BLOCK), 0, s->stream, gm, L, kabs, ksca, g, dz, dx,
                       s->l1d, (float *)s->coef, hash, samp);
  s->dd_hash_ready = hash != nullptr;
  return TSX_OK;
}

extern "C" int tsx_diff_set_optprop(tsx_solver *s, const double *kabs, const double *ksca, const double *g,
                                    const double *dz, double dx, const uint8_t *l1d, const double *a11, const double *a12,
                                    const double *albedo, int where) {
  ARGCHK(s && kabs && ksca && g && dz && l1d && albedo, "tsx_diff_set_optprop: null argument");
  ARGCHK(dx > 0, "tsx_diff_set_optprop: dx <= 0");
  if (!s->lut_diff.ready) {
    tsx_set_error("tsx_diff_set_optprop: load the diffuse LUT first (tsx_lut_set_diffuse / tsx_lut_load_diffuse_mmap4)");
    return TSX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(s->device));
  const TsxGeo &gm = s->geo;
  int rc = set_aux(s, l1d, a11, a12, albedo, where);
  if (rc) return rc;
  if ((rc = ensure_coef_storage(s, 4))) return rc;
  const size_t nb = sizeof(double) * gm.Nc;
  const double *p[4] = {kabs, ksca, g, dz};
  TsxDevTmp tmp[4];
  if (where == TSX_HOST) {
    for (int q = 0; q < 4; ++q) {
      HIPCHK(tmp[q].alloc(nb));
      HIPCHK(hipMemcpyAsync(tmp[q].p, p[q], nb, hipMemcpyHostToDevice, s->stream));
      p[q] = tmp[q].as<double>();
    }
  }
  if ((rc = lut_diffuse_launch(s, p[0], p[1], p[2], p[3], dx))) return rc;  // (resets / sets the shared-storage state)
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s->stream));
  s->have_coeffs = true;
  s->coef_h_valid = false;
  s->pcx_valid = false;
  return TSX_OK;
}

extern "C" int tsx_diff_get_coeffs(tsx_solver *s, double *diff2diff, int where) {
  ARGCHK(s && diff2diff, "tsx_diff_get_coeffs: null argument");
  if (!s->have_coeffs) {
    tsx_set_error("tsx_diff_get_coeffs: no coefficients set");
    return TSX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(s->device));
  {
    int rc = tsx_coef_ensure_dense(s);  // the LUT path may have left the blocks in the shared storage only
    if (rc) return rc;
  }
  const TsxGeo &g = s->geo;
  const int DD = g.D * g.D;
  const size_t ncoef = (size_t)DD * g.Nc;
  double *out = diff2diff;
  TsxDevTmp tmp_guard;
  double *tmp = nullptr;
  if (where == TSX_HOST) {
    HIPCHK(tmp_guard.alloc(ncoef * sizeof(double)));
    tmp = tmp_guard.as<double>();
    out = tmp;
  }
  if (s->coef_bytes == 4)
    hipLaunchKernelGGL((tsx_k_export_coeff<float>), dim3(grid_for((long long)ncoef, 8192)), dim3(TSX_BLOCK), 0, s->stream, g,
                       DD, (const float *)s->coef, out);
  else
    hipLaunchKernelGGL((tsx_k_export_coeff<double>), dim3(grid_for((long long)ncoef, 8192)), dim3(TSX_BLOCK), 0, s->stream, g,
                       DD, (const double *)s->coef, out);
  HIPCHK(hipGetLastError());
  if (where == TSX_HOST) HIPCHK(hipMemcpyAsync(diff2diff, tmp, ncoef * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  return TSX_OK;
}

// ------------------------------------------------------------------------------------------------
template <int NTOP, int NSIDE>
static int diff_apply_t(tsx_solver *s, const double *x, double *y, int where) {
  const TsxGeo &g = s->geo;
  const size_t nb = (size_t)g.N * sizeof(double);
  int rc = ensure_stage(s);
  if (rc) return rc;
  const double *xd = x;
  double *yd = y;
  if (where == TSX_HOST) {
    HIPCHK(hipMemcpyAsync(s->stage_a, x, nb, hipMemcpyHostToDevice, s->stream));
    xd = s->stage_a;
    yd = s->stage_b;
  }
  if ((rc = import_vec<NTOP, NSIDE>(s, xd, s->vp))) return rc;
  if ((rc = launch_spmv<NTOP, NSIDE, 0>(s, (const double *)s->vp, s->vv, (const double *)nullptr, false))) return rc;
  if ((rc = export_vec<NTOP, NSIDE>(s, s->vv, yd))) return rc;
  if (where == TSX_HOST) HIPCHK(hipMemcpyAsync(y, s->stage_b, nb, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  return TSX_OK;
}

extern "C" int tsx_diff_apply(tsx_solver *s, const double *x, double *y, int where) {
  ARGCHK(s && x && y, "tsx_diff_apply: null argument");
  if (!s->have_coeffs) {
    tsx_set_error("tsx_diff_apply: call tsx_diff_set_coeffs first");
    return TSX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(s->device));
  {
    int rc = tsx_dedup_ensure(s);
    if (rc) return rc;
  }
  return s->geo.ntop == 2 ? diff_apply_t<2, 4>(s, x, y, where) : diff_apply_t<8, 4>(s, x, y, where);
}

// TSX_HALF_EXIT=0 switches the stop test at BiCGStab's half step off (A/B; read per call)
static bool tsx_half_exit() {
  const char *e = getenv("TSX_HALF_EXIT");
  return !(e && atoi(e) == 0);
}

// One BiCGStab iteration on the stream (no host synchronisation).
// One BiCGStab iteration on the stream (no host synchronisation).  MIX: preconditioned directions and the shadow
// residual live in fp32 (s->mixed); x, r, p, s, v, t stay fp64.
template <int NTOP, int NSIDE, bool MIX>
static int enqueue_iteration_t(tsx_solver *s, bool first, bool half_ok) {
  using PT = typename std::conditional<MIX, float, double>::type;  // directions
  using RT = PT;                                                    // shadow residual
  const TsxGeo &g = s->geo;
  const long long n2 = g.N / 2;
  const int nbv = grid_for(n2);
  const bool half = half_ok && tsx_half_exit();
  int rc;
  if (!first) {
    if (MIX && s->pc != TSX_PC_NONE)  // p lives in fp32 only (s->v32), see tsx_k_pupdate32
      hipLaunchKernelGGL(tsx_k_pupdate32, dim3(nbv), dim3(TSX_BLOCK), 0, s->stream, n2, s->scal, (const double2 *)s->vr,
                         (const double2 *)s->vv, s->p32, g, (int)s->pc_split);
    else
      hipLaunchKernelGGL(tsx_k_pupdate, dim3(nbv), dim3(TSX_BLOCK), 0, s->stream, n2, s->scal, (const double2 *)s->vr,
                         (double2 *)s->vp, (const double2 *)s->vv, (float2 *)nullptr, g, 0);
  }
  const RT *rhat = (const RT *)s->vrhat;
  if (s->pc != TSX_PC_NONE) {
    PT *ph = (PT *)s->vph, *sh = (PT *)s->vsh;
    s->pc_rhs = s->p32;
    if ((rc = tsx_pc_apply(s, s->vp, ph, std::is_same<PT, float>::value, true))) return rc;
    if ((rc = launch_spmv<NTOP, NSIDE, 1, PT, RT>(s, ph, s->vv, rhat, true))) return rc;
    if ((rc = scalar_stage(s, spmv_nblocks(s), 1, TSX_STAGE_ALPHA))) return rc;
    hipLaunchKernelGGL(tsx_k_supdate, dim3(nbv), dim3(TSX_BLOCK), 0, s->stream, n2, s->scal, (const double2 *)s->vr,
                       (const double2 *)s->vv, (double2 *)s->vs, MIX ? (float2 *)s->v32 : (float2 *)nullptr, g, (int)s->pc_split,
                       half ? s->partials : (double *)nullptr);
    if (half && (rc = scalar_stage(s, nbv, 1, TSX_STAGE_HALF))) return rc;
    s->pc_rhs = s->v32;
    if ((rc = tsx_pc_apply(s, s->vs, sh, std::is_same<PT, float>::value, true))) return rc;
    if ((rc = launch_spmv<NTOP, NSIDE, 5, PT, double>(s, sh, s->vt, s->vs, true))) return rc;
    if ((rc = scalar_stage(s, spmv_nblocks(s), 3, TSX_STAGE_OMEGA))) return rc;
    hipLaunchKernelGGL((tsx_k_xrupdate<PT, RT>), dim3(nbv), dim3(TSX_BLOCK), 0, s->stream, n2, s->scal, (double2 *)s->vx, ph, sh,
                       (const double2 *)s->vs, (const double2 *)s->vt, rhat, (double2 *)s->vr, s->partials);
  } else {
    if ((rc = launch_spmv<NTOP, NSIDE, 1, double, RT>(s, s->vp, s->vv, rhat, true))) return rc;
    if ((rc = scalar_stage(s, spmv_nblocks(s), 1, TSX_STAGE_ALPHA))) return rc;
    hipLaunchKernelGGL(tsx_k_supdate, dim3(nbv), dim3(TSX_BLOCK), 0, s->stream, n2, s->scal, (const double2 *)s->vr,
                       (const double2 *)s->vv, (double2 *)s->vs, (float2 *)nullptr, g, 0, half ? s->partials : (double *)nullptr);
    if (half && (rc = scalar_stage(s, nbv, 1, TSX_STAGE_HALF))) return rc;
    if ((rc = launch_spmv<NTOP, NSIDE, 5, double, double>(s, s->vs, s->vt, s->vs, true))) return rc;
    if ((rc = scalar_stage(s, spmv_nblocks(s), 3, TSX_STAGE_OMEGA))) return rc;
    hipLaunchKernelGGL((tsx_k_xrupdate<double, RT>), dim3(nbv), dim3(TSX_BLOCK), 0, s->stream, n2, s->scal, (double2 *)s->vx,
                       (const double *)s->vp, (const double *)s->vs, (const double2 *)s->vs, (const double2 *)s->vt, rhat,
                       (double2 *)s->vr, s->partials);
  }
  if ((rc = scalar_stage(s, nbv, 2, TSX_STAGE_RHO))) return rc;
  HIPCHK(hipGetLastError());
  return TSX_OK;
}
// The same iteration with the recurrence vectors in fp32 (tsx_kernels.hpp "fp32 Krylov vectors"): r in vr, v in vv, s in vs
// (natural order; the preconditioner's colour-split copy in v32), t in vt -- the fp64 buffers reused as float arrays.
template <int NTOP, int NSIDE>
static int enqueue_iteration_k32(tsx_solver *s, bool first, bool half_ok) {
  const TsxGeo &g = s->geo;
  const long long n2 = g.N / 2;
  const int nbv = grid_for(n2);
  int rc;
  float *r32 = (float *)s->vr, *v32k = (float *)s->vv, *s32n = (float *)s->vs, *t32 = (float *)s->vt;
  const float *rhat = (const float *)s->vrhat;
  float *ph = (float *)s->vph, *sh = (float *)s->vsh;
  // with the scan passes the updates of p and s run cell by cell and leave the passes' bf16-pair words too
  const bool words = s->pc_split && s->coef_h_scan && tsx_pcs_rhs16(s) && s->pc_sweeps + 1 >= 6 &&
                     !(getenv("TSX_PC_WORDS") && atoi(getenv("TSX_PC_WORDS")) == 0);
  const int nbc = grid_for(g.Nc);
  const bool half = half_ok && tsx_half_exit();
  double *hp = half ? s->partials : (double *)nullptr;
  if (!first) {
    if (words) {
      if (NTOP == 2)
        hipLaunchKernelGGL((tsx_k_psupdate_k32c<10, 0>), dim3(nbc), dim3(TSX_BLOCK), 0, s->stream, g, s->scal, (const float *)r32,
                           (const float *)v32k, s->p32, (float *)nullptr, tsx_pcs_words(s), (double *)nullptr);
      else
        hipLaunchKernelGGL((tsx_k_psupdate_k32c<16, 0>), dim3(nbc), dim3(TSX_BLOCK), 0, s->stream, g, s->scal, (const float *)r32,
                           (const float *)v32k, s->p32, (float *)nullptr, tsx_pcs_words(s), (double *)nullptr);
      s->pc_words_ready = true;
    } else {
      hipLaunchKernelGGL(tsx_k_pupdate_k32, dim3(nbv), dim3(TSX_BLOCK), 0, s->stream, n2, s->scal, (const float2 *)r32,
                         (const float2 *)v32k, s->p32, g, (int)s->pc_split);
    }
  }
  s->pc_rhs = s->p32;
  if ((rc = tsx_pc_apply(s, s->vp, ph, true, true))) return rc;
  if ((rc = launch_spmv_f32<NTOP, NSIDE, 1>(s, ph, v32k, rhat, true))) return rc;
  if ((rc = scalar_stage(s, spmv_nblocks(s), 1, TSX_STAGE_ALPHA))) return rc;
  // the preconditioner reads s in its own order: natural (then one copy serves both) or colour-split (a second copy)
  float *sdst = s->pc_split ? s32n : s->v32;
  if (words) {
    if (NTOP == 2)
      hipLaunchKernelGGL((tsx_k_psupdate_k32c<10, 1>), dim3(nbc), dim3(TSX_BLOCK), 0, s->stream, g, s->scal, (const float *)r32,
                         (const float *)v32k, s->v32, sdst, tsx_pcs_words(s), hp);
    else
      hipLaunchKernelGGL((tsx_k_psupdate_k32c<16, 1>), dim3(nbc), dim3(TSX_BLOCK), 0, s->stream, g, s->scal, (const float *)r32,
                         (const float *)v32k, s->v32, sdst, tsx_pcs_words(s), hp);
    s->pc_words_ready = true;
    if (half && (rc = scalar_stage(s, nbc, 1, TSX_STAGE_HALF))) return rc;
  } else {
    hipLaunchKernelGGL(tsx_k_supdate_k32, dim3(nbv), dim3(TSX_BLOCK), 0, s->stream, n2, s->scal, (const float2 *)r32,
                       (const float2 *)v32k, (float2 *)sdst, s->pc_split ? s->v32 : (float *)nullptr, g, hp);
    if (half && (rc = scalar_stage(s, nbv, 1, TSX_STAGE_HALF))) return rc;
  }
  s->pc_rhs = s->v32;
  if ((rc = tsx_pc_apply(s, s->vs, sh, true, true))) return rc;
  if ((rc = launch_spmv_f32<NTOP, NSIDE, 5>(s, sh, t32, sdst, true))) return rc;
  if ((rc = scalar_stage(s, spmv_nblocks(s), 3, TSX_STAGE_OMEGA))) return rc;
  hipLaunchKernelGGL(tsx_k_xrupdate_k32, dim3(nbv), dim3(TSX_BLOCK), 0, s->stream, n2, s->scal, (double2 *)s->vx, (const float2 *)ph,
                     (const float2 *)sh, (const float2 *)sdst, (const float2 *)t32, (const float2 *)rhat, (float2 *)r32, s->partials);
  if ((rc = scalar_stage(s, nbv, 2, TSX_STAGE_RHO))) return rc;
  HIPCHK(hipGetLastError());
  return TSX_OK;
}
// replace the recurrence residual by b - A x evaluated in fp64 on the exact blocks; renews rho, the norm and the stop decision
template <int NTOP, int NSIDE>
static int k32_replace_residual(tsx_solver *s) {
  const TsxGeo &g = s->geo;
  int rc;
  // t (fp32, in vt) is dead between iterations: its buffer takes A x in fp64
  if ((rc = launch_spmv<NTOP, NSIDE, 0>(s, (const double *)s->vx, s->vt, (const double *)nullptr, true))) return rc;
  const int nbv = grid_for(g.N);
  hipLaunchKernelGGL(tsx_k_residual_k32, dim3(nbv), dim3(TSX_BLOCK), 0, s->stream, g.N, s->vb, s->vt, (float *)s->vr,
                     (float *)s->vrhat, s->p32, s->partials, g, (int)s->pc_split, 0, 0, &s->scal->done);
  if ((rc = scalar_stage(s, nbv, 2, TSX_STAGE_REPLACE))) return rc;
  return TSX_OK;
}
template <int NTOP, int NSIDE>
static int enqueue_iteration(tsx_solver *s, bool first, bool half_ok = true) {
  // half_ok: run the stop test at the half step (TSX_STAGE_HALF: one more scalar stage, on several ranks one more all-reduce);
  // the Krylov loop asks for it only in iterations in which it can plausibly fire
  if (s->k32) return enqueue_iteration_k32<NTOP, NSIDE>(s, first, half_ok);
  return s->mixed ? enqueue_iteration_t<NTOP, NSIDE, true>(s, first, half_ok) : enqueue_iteration_t<NTOP, NSIDE, false>(s, first, half_ok);
}

template <int NTOP, int NSIDE>
static int krylov_begin(tsx_solver *s, const tsx_ksp_opts *o, bool restart = false) {
  const TsxGeo &g = s->geo;
  int rc;
  if (!restart) {
    TsxScalars init;
    memset(&init, 0, sizeof(init));
    init.rtol = o->rtol;
    init.atol = o->atol;
    init.dtol = o->dtol;
    init.maxit = o->maxit;
    // half-step stop test: on the fp32 recurrence the decision is taken again on the true residual, so ask for a little more
    init.half_margin = s->k32 ? 0.9 : 1.0;
    *s->scal_host = init;
  } else {  // breakdown restart: keep iteration count, history and the initial norm; new shadow residual
    s->scal_host->done = 0;
    s->scal_host->reason = 0;
    s->scal_host->half = 0;
    s->scal_host->restart = 1;
  }
  HIPCHK(hipMemcpyAsync(s->scal, s->scal_host, sizeof(TsxScalars), hipMemcpyHostToDevice, s->stream));
  // r = b - A x0 (nonzero initial guess, src/pprts.F90:4343); rhat = p = r.  A guess known to be zero on every rank (a cold
  // start) spares the operator apply: r = b
  const int yzero = s->x_is_zero && !restart ? 1 : 0;
  if (!yzero && (rc = launch_spmv<NTOP, NSIDE, 0>(s, (const double *)s->vx, s->vt, (const double *)nullptr, false))) return rc;
  s->x_is_zero = false;  // from here on x is the iterate
  const int nbv = grid_for(g.N);
  if (s->k32)
    hipLaunchKernelGGL(tsx_k_residual_k32, dim3(nbv), dim3(TSX_BLOCK), 0, s->stream, g.N, s->vb, s->vt, (float *)s->vr,
                       (float *)s->vrhat, s->p32, s->partials, g, (int)s->pc_split, yzero, 1, (const int *)nullptr);
  else if (s->mixed)
    hipLaunchKernelGGL(tsx_k_residual0<float>, dim3(nbv), dim3(TSX_BLOCK), 0, s->stream, g.N, s->vb, s->vt, s->vr,
                       (float *)s->vrhat, s->vp, s->pc != TSX_PC_NONE ? s->p32 : (float *)nullptr, s->partials, g, (int)s->pc_split,
                       yzero);
  else
    hipLaunchKernelGGL(tsx_k_residual0<double>, dim3(nbv), dim3(TSX_BLOCK), 0, s->stream, g.N, s->vb, s->vt, s->vr, s->vrhat,
                       s->vp, (float *)nullptr, s->partials, g, 0, yzero);
  if ((rc = scalar_stage(s, nbv, 2, TSX_STAGE_INIT))) return rc;
  return TSX_OK;
}

// what an iteration-count hint was measured with (krylov_run)
static unsigned long long hint_key(const tsx_solver *s, const tsx_ksp_opts *o) {
  unsigned long long h = 0x9e3779b97f4a7c15ull, v;
  const double d[2] = {o->rtol, o->atol};
  for (int q = 0; q < 2; ++q) {
    memcpy(&v, &d[q], sizeof(v));
    h = (h ^ v) * 0xff51afd7ed558ccdull;
  }
  v = ((unsigned long long)(unsigned)s->pc << 40) | ((unsigned long long)(unsigned)s->pc_sweeps << 8) | (s->mixed ? 2u : 0u) | (s->k32 ? 1u : 0u);
  return (h ^ v) * 0xff51afd7ed558ccdull;
}

// The Krylov loop on the internal vectors s->vb (rhs) and s->vx (initial guess in, solution out).
// Records ev0/ev1 around it; leaves the final scalars in s->scal_host.
template <int NTOP, int NSIDE>
static int krylov_run(tsx_solver *s, const tsx_ksp_opts *o) {
  int rc;
  HIPCHK(hipEventRecord(s->ev0, s->stream));
  const bool cold = s->x_is_zero;   // a zero guess: the first iterations are far from the stop rule
  if ((rc = krylov_begin<NTOP, NSIDE>(s, o))) return rc;
  const int chunk = o->check_every > 0 ? o->check_every : 2;
  // the half-step test pays where the residual is about to meet the rule: predicted from the last two known norms (several
  // ranks hold the same all-reduced history, so they agree); a warm start may be converged at once -- there from the start
  auto half_plausible = [&](int ahead) {
    const TsxScalars &h = *s->scal_host;
    if (h.nhist < 1 || h.its == 0) return !cold;
    const double last = h.hist[h.nhist - 1], prev = h.nhist >= 2 ? h.hist[h.nhist - 2] : last * 10.0;
    const double ratio = prev > 0.0 && last < prev ? last / prev : 1.0;
    return last * pow(ratio, (double)ahead) <= 30.0 * fmax(o->rtol * h.rnorm0, o->atol);
  };
  int enq = 0, nrestart = 0;
  bool done = false, first_after_begin = true;
  while (!done) {
    int todo = (o->maxit - enq) < chunk ? (o->maxit - enq) : chunk;
    // the first look at the flag comes where the previous solve of this handle with the same kind of start (cold / warm) ended,
    // less one: a repeated solve (a spectral loop, a time step) then synchronises with the host twice instead of every
    // check_every iterations; overshooting costs the empty launches of the surplus iterations (about 0.1 ms each)
    // The hint belongs to the solver settings it was measured with (tolerances, preconditioner, pass count, precision of the
    // recurrence), and it is capped: on the fp32 recurrence the re-anchoring to the true residual and the breakdown restart
    // are only evaluated at a look, so a hard solve must not run dozens of iterations before the first one
    const unsigned long long hkey = hint_key(s, o);
    if (enq == 0 && o->check_every <= 0 && hkey == s->its_hint_key) {
      int hint = cold ? s->its_hint_cold : s->its_hint_warm;
      if (hint - 1 > 8) hint = 9;
      if (hint - 1 > todo) todo = (hint - 1) < (o->maxit - enq) ? hint - 1 : (o->maxit - enq);
    }
    {
      // if the last iteration's reduction, applied once more, meets the stop rule, enqueue ONE iteration before the next look
      // at the flag: the iterations enqueued beyond convergence return at once, but that is ~60 empty launches (0.1 ms)
      const TsxScalars &h = *s->scal_host;
      if (enq > 0 && todo > 1 && h.nhist >= 2 && h.nhist <= 100 && h.its == enq) {
        const double last = h.hist[h.nhist - 1], prev = h.hist[h.nhist - 2];
        if (prev > 0.0 && last < prev && last * (last / prev) <= fmax(o->rtol * h.rnorm0, o->atol)) todo = 1;
      }
    }
    for (int q = 0; q < todo; ++q, ++enq) {
      if ((rc = enqueue_iteration<NTOP, NSIDE>(s, first_after_begin, half_plausible(q + 1)))) return rc;
      first_after_begin = false;
    }
    HIPCHK(hipMemcpyAsync(s->scal_host, s->scal, sizeof(TsxScalars), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->scal_host->flow_err) {  // tsx_k_pcs_flow: a bounded wait for a neighbour tile expired (cannot happen by construction)
      if (tsx_peer_check(s) != TSX_OK) return TSX_ERR_COMM;  // ... or for a neighbour rank's message: the mailbox says which
      tsx_set_error("preconditioner flow kernel: a wait for a neighbour tile's progress word expired (TSX_PC_FLOW=0 runs a launch per pass)");
      return TSX_ERR_HIP;
    }
    const bool was_half = s->scal_host->done && s->scal_host->half;
    if (was_half) {
      // the stop rule was met by s = r - alpha v in the middle of an iteration (TSX_STAGE_HALF): the kernels of its second half
      // returned at once; the iterate is x + alpha p-hat
      const TsxGeo &gq = s->geo;
      if (s->pc == TSX_PC_NONE)   // no preconditioner: p-hat is p itself (fp64 in every variant)
        hipLaunchKernelGGL(tsx_k_xhalf<double>, dim3(grid_for(gq.N)), dim3(TSX_BLOCK), 0, s->stream, gq.N, s->scal, s->vx,
                           (const double *)s->vp);
      else if (s->k32 || s->mixed)
        hipLaunchKernelGGL(tsx_k_xhalf<float>, dim3(grid_for(gq.N)), dim3(TSX_BLOCK), 0, s->stream, gq.N, s->scal, s->vx,
                           (const float *)s->vph);
      else
        hipLaunchKernelGGL(tsx_k_xhalf<double>, dim3(grid_for(gq.N)), dim3(TSX_BLOCK), 0, s->stream, gq.N, s->scal, s->vx,
                           (const double *)s->vph);
      HIPCHK(hipGetLastError());
      s->scal_host->half = 0;
      enq = s->scal_host->its;
    }
    if (s->k32) {
      // fp32 recurrence: convergence is declared on the true residual only, and the recurrence is re-anchored to it whenever
      // it has fallen four orders of magnitude since the last anchor (it cannot follow b - A x much further in fp32)
      TsxScalars &h = *s->scal_host;
      const bool conv = h.done && h.reason > 0;
      const bool far = !h.done && h.rnorm <= 1e-4 * h.rnorm_true;
      if ((conv || far) && h.rnorm_true != h.rnorm) {
        if (conv) {  // un-declare: the kernels of the replacement must run
          h.done = 0;
          h.reason = 0;
          HIPCHK(hipMemcpyAsync(s->scal, s->scal_host, sizeof(TsxScalars), hipMemcpyHostToDevice, s->stream));
        }
        if ((rc = k32_replace_residual<NTOP, NSIDE>(s))) return rc;
        HIPCHK(hipMemcpyAsync(s->scal_host, s->scal, sizeof(TsxScalars), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        if (conv && !h.done) enq = h.its;  // iterations enqueued after the premature stop did not run
        if (was_half && !h.done && enq < o->maxit) {
          // the half step's recurrence residual met the rule, the true residual does not (yet): omega of this iteration was never
          // formed, so the recurrence cannot go on -- start a new Krylov sequence from the current iterate (rhat = p = r = b - A x)
          if ((rc = krylov_begin<NTOP, NSIDE>(s, o, true))) return rc;
          first_after_begin = true;
        }
      }
    }
    done = s->scal_host->done != 0 || enq >= o->maxit;
    if (s->scal_host->done && s->scal_host->reason == -5 && nrestart < 3 && s->scal_host->its < o->maxit) {
      // rho / (rhat,v) breakdown: restart from the current iterate with rhat = r (x keeps its progress).  The
      // reference falls back to GMRES from a zero guess here (src/pprts.F90:4277-4296).
      ++nrestart;
      enq = s->scal_host->its;
      if ((rc = krylov_begin<NTOP, NSIDE>(s, o, true))) return rc;
      first_after_begin = true;
      done = false;
    }
  }
  HIPCHK(hipEventRecord(s->ev1, s->stream));
  if (s->scal_host->reason > 0 && !s->in_retry) {  // (the conservative retry solver's count says nothing about the next default solve)
    const unsigned long long hkey = hint_key(s, o);
    if (hkey != s->its_hint_key) s->its_hint_cold = s->its_hint_warm = 0;
    s->its_hint_key = hkey;
    (cold ? s->its_hint_cold : s->its_hint_warm) = s->scal_host->its;
  }
  return TSX_OK;
}

static int prepare_ksp(tsx_solver *s, const tsx_ksp_opts *opts, tsx_ksp_opts *o);

// The reference's failure path (src/pprts.F90:4277-4302): a solve that ends with a non-positive reason is repeated once from
// a zero initial guess with a second, more conservative solver (there: GMRES on the same preconditioner); only if that
// fails too the negative reason is reported (and the caller aborts).  Here the second solver is the same flexible
// BiCGStab on exact fp64 blocks and fp64 directions with the zebra-ordered exact column solves -- nothing reduced.
// The explicit (stationary) solver on the internal vectors: x += M^-1 (b - A x) until the change of the iterate meets
// explicit_ediff's stop rule (see tsx_k_defect / TSX_STAGE_EXPLICIT).  One outer iteration = one operator apply, one
// application of the sweeps (pc_sweeps + 1 half-grid passes: the reference's -pc_sub_it), one update.
template <int NTOP, int NSIDE, bool MIX>
static int explicit_run_t(tsx_solver *s, const tsx_ksp_opts *o) {
  using PT = typename std::conditional<MIX, float, double>::type;
  const TsxGeo &g = s->geo;
  int rc;
  if (s->pc == TSX_PC_NONE) {
    tsx_set_error("explicit solver: needs the sweeps (pc != TSX_PC_NONE)");
    return TSX_ERR_ARG;
  }
  HIPCHK(hipEventRecord(s->ev0, s->stream));
  s->x_is_zero = false;  // vx becomes the iterate (only krylov_begin consumes the flag; it must not outlive this solve)
  TsxScalars init;
  memset(&init, 0, sizeof(init));
  init.rtol = o->rtol;
  init.atol = o->atol;
  init.dtol = o->dtol;
  init.maxit = o->maxit;
  init.nranks = s->grid.nranks > 1 ? s->grid.nranks : 1;
  *s->scal_host = init;
  HIPCHK(hipMemcpyAsync(s->scal, s->scal_host, sizeof(TsxScalars), hipMemcpyHostToDevice, s->stream));
  const int nbv = grid_for(g.N);
  const int chunk = o->check_every > 0 ? o->check_every : 2;
  PT *z = (PT *)s->vph;
  int enq = 0;
  bool done = false;
  while (!done) {
    const int todo = (o->maxit - enq) < chunk ? (o->maxit - enq) : chunk;
    for (int q = 0; q < todo; ++q, ++enq) {
      if ((rc = launch_spmv<NTOP, NSIDE, 0>(s, (const double *)s->vx, s->vt, (const double *)nullptr, true))) return rc;
      if (MIX) {
        hipLaunchKernelGGL(tsx_k_defect<float>, dim3(nbv), dim3(TSX_BLOCK), 0, s->stream, g.N, s->scal, s->vb, s->vt, s->p32, g,
                           (int)s->pc_split);
        s->pc_rhs = s->p32;
      } else {
        hipLaunchKernelGGL(tsx_k_defect<double>, dim3(nbv), dim3(TSX_BLOCK), 0, s->stream, g.N, s->scal, s->vb, s->vt, s->vp, g, 0);
      }
      if ((rc = tsx_pc_apply(s, s->vp, z, MIX, true))) return rc;
      hipLaunchKernelGGL(tsx_k_xplus<PT>, dim3(nbv), dim3(TSX_BLOCK), 0, s->stream, g.N, s->scal, s->vx, (const PT *)z, s->partials);
      if ((rc = scalar_stage(s, nbv, 1, TSX_STAGE_EXPLICIT))) return rc;
    }
    HIPCHK(hipMemcpyAsync(s->scal_host, s->scal, sizeof(TsxScalars), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    done = s->scal_host->done != 0 || enq >= o->maxit;
  }
  HIPCHK(hipEventRecord(s->ev1, s->stream));
  return TSX_OK;
}

// solves in flight in this process: the flow kernel's workgroups spin while they wait for their neighbours, which is time other
// instances' kernels could have had (bench_specint.py with four instances: 137 g-points/s with it, 140 with a launch per pass; the
// cold call 71 against 84) -- an application of M^-1 takes the launches while another instance is inside a solve (tsx_pcs_flow_ok)
static std::atomic<int> g_active_solves{0};
int tsx_active_solves() { return g_active_solves.load(std::memory_order_relaxed); }
struct TsxSolveInFlight {
  TsxSolveInFlight() { g_active_solves.fetch_add(1, std::memory_order_relaxed); }
  ~TsxSolveInFlight() { g_active_solves.fetch_sub(1, std::memory_order_relaxed); }
};

template <int NTOP, int NSIDE>
static int krylov_run_with_retry(tsx_solver *s, tsx_ksp_opts *o) {
  TsxSolveInFlight in_flight;
  TsxLogScope log_scope(s, TSX_EV_SOLVE_MDIFF);  // solve_Mdiff, src/pprts.F90:3012-3021
  if (o->explicit_solver)
    return s->mixed ? explicit_run_t<NTOP, NSIDE, true>(s, o) : explicit_run_t<NTOP, NSIDE, false>(s, o);
  int rc = krylov_run<NTOP, NSIDE>(s, o);
  if (rc) return rc;
  const int reason = s->scal_host->done ? s->scal_host->reason : -3;
  // -accept_incomplete_solve: the reference returns before its retry (src/pprts.F90:4271-4273) -- the partial iterate stays
  if (reason > 0 || o->accept_incomplete_solve || getenv("TSX_NO_RETRY")) return TSX_OK;
  const int its_first = s->scal_host->its;
  tsx_ksp_opts o2 = *o;
  o2.fp32_directions = 0;
  o2.pc_coeff_fp16 = 0;
  o2.pc = TSX_PC_REDBLACK;  // on the exact blocks (tsx_pcx.hip); zebra rows where that is not available (prepare_ksp)
  o2.pc_sweeps = 0;         // ... with that path's own pass count
  tsx_ksp_opts o3;
  if ((rc = prepare_ksp(s, &o2, &o3))) return rc;
  HIPCHK(hipMemsetAsync(s->vx, 0, sizeof(double) * (size_t)s->geo.N, s->stream));
  s->x_is_zero = true;
  hipEvent_t keep0 = s->ev0;  // solve_ms covers both attempts: keep the first start event
  hipEvent_t tmp;
  HIPCHK(hipEventCreate(&tmp));
  s->ev0 = tmp;
  s->in_retry = true;
  rc = krylov_run<NTOP, NSIDE>(s, &o3);
  s->in_retry = false;
  s->ev0 = keep0;
  (void)hipEventDestroy(tmp);
  if (rc) return rc;
  s->scal_host->its += its_first;  // Niter_diff counts the work of both attempts
  return TSX_OK;
}

static int fill_result(tsx_solver *s, tsx_ksp_result *res) {
  {
    int rc = tsx_peer_check(s);  // the stream has been synchronised: did a bounded wait of the peer transport expire?
    if (rc) return rc;
  }
  if (!res) return TSX_OK;
  const TsxScalars &h = *s->scal_host;
  memset(res, 0, sizeof(*res));
  res->reason = h.done ? h.reason : -3;  // KSP_DIVERGED_ITS
  res->niter = h.its;
  res->rnorm0 = h.rnorm0;
  res->rnorm = h.rnorm;
  res->nhist = h.nhist;
  memcpy(res->res_hist, h.hist, sizeof(double) * 100);
  HIPCHK(hipEventElapsedTime(&res->solve_ms, s->ev0, s->ev1));
  return TSX_OK;
}

static int allreduce_host(tsx_solver *s, double *v, int n);  // tsx_pipeline_api.inc

template <int NTOP, int NSIDE>
static int diff_solve_t(tsx_solver *s, const double *b, double *x, int where, const tsx_ksp_opts *o,
                        tsx_ksp_result *res) {
  const TsxGeo &g = s->geo;
  const size_t nb = (size_t)g.N * sizeof(double);
  int rc;
  const double *bd = b;
  double *xd = x;
  const hipEvent_t e_imp0 = s->ev_imp0, e_imp1 = s->ev_imp1, e_exp1 = s->ev_exp1;  // created once in tsx_create
  if (where == TSX_HOST) {
    if ((rc = ensure_stage(s))) return rc;
    HIPCHK(hipMemcpyAsync(s->stage_a, b, nb, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(s->stage_b, x, nb, hipMemcpyHostToDevice, s->stream));
    bd = s->stage_a;
    xd = s->stage_b;
  }
  HIPCHK(hipEventRecord(e_imp0, s->stream));
  if ((rc = import_vec<NTOP, NSIDE>(s, bd, s->vb))) return rc;
  if (o->initial_guess_zero) {  // the caller says so: x is not read
    HIPCHK(hipMemsetAsync(s->vx, 0, nb, s->stream));
    s->x_is_zero = true;
  } else {  // the import of the guess also tells whether it is zero
    int *nzflag = &s->scal->aux_flag;
    HIPCHK(hipMemsetAsync(nzflag, 0, sizeof(int), s->stream));
    if ((rc = import_vec<NTOP, NSIDE>(s, xd, s->vx, nzflag))) return rc;
    int nz = 1;
    HIPCHK(hipMemcpyAsync(&nz, nzflag, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    double anynz = nz;
    if (s->grid.nranks > 1 && (rc = allreduce_host(s, &anynz, 1))) return rc;  // zero only if zero on every rank
    s->x_is_zero = anynz == 0.0;
  }
  HIPCHK(hipEventRecord(e_imp1, s->stream));

  {
    tsx_ksp_opts oo = *o;
    if ((rc = krylov_run_with_retry<NTOP, NSIDE>(s, &oo))) return rc;
  }
  HIPCHK(hipEventRecord(s->ev1, s->stream));
  if ((rc = export_vec<NTOP, NSIDE>(s, s->vx, xd))) return rc;
  HIPCHK(hipEventRecord(e_exp1, s->stream));
  if (where == TSX_HOST) HIPCHK(hipMemcpyAsync(x, s->stage_b, nb, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));

  if (res) {
    if ((rc = fill_result(s, res))) return rc;
    HIPCHK(hipEventElapsedTime(&res->import_ms, e_imp0, e_imp1));
    HIPCHK(hipEventElapsedTime(&res->export_ms, s->ev1, e_exp1));
  }
  return TSX_OK;
}

static int prepare_ksp(tsx_solver *s, const tsx_ksp_opts *opts, tsx_ksp_opts *o) {
  TsxLogScope log_scope(s, TSX_EV_SETUP_MDIFF);  // setup_Mdiff, src/pprts.F90:2952-2954: here the packing / sharing of the coefficient set
  if (opts) *o = *opts;
  else tsx_default_ksp_opts(o);
  ARGCHK(o->maxit >= 1, "solve: maxit < 1");
  ARGCHK(o->pc >= TSX_PC_NONE && o->pc <= TSX_PC_REDBLACK, "solve: unsupported preconditioner");
  ARGCHK(o->pc_sweeps >= 0 && o->pc_sweeps <= 32, "solve: pc_sweeps out of range");
  HIPCHK(hipSetDevice(s->device));
  s->pc = o->pc;
  s->pc_sweeps = o->pc_sweeps;
  // fp32 directions (default on) always come with the packed reduced-precision preconditioner blocks; the multi-sweep
  // Jacobi refinement and pc_coeff_fp16 = 0 work on fp64 directions and the exact blocks
  s->mixed = o->fp32_directions != 0 && !(o->pc == TSX_PC_COLUMN && o->pc_sweeps > 1) &&
             (o->pc == TSX_PC_NONE || o->pc_coeff_fp16 != 0);
  // fp32 recurrence vectors too (fp32_directions = 2): with a preconditioner on the fp32 path; TSX_K32=0 falls back to 1
  // ... and for tolerances an fp32 recurrence reaches between two replacements (rtol >= 1e-7: the reference's default is 1e-5);
  // tighter solves keep the fp64 recurrence -- every replacement costs BiCGStab some of its super-linear convergence (a
  // 130-level column at rtol 1e-10: 36 instead of 30 iterations)
  s->k32 = s->mixed && o->fp32_directions >= 2 && o->pc != TSX_PC_NONE && !o->explicit_solver && o->rtol >= 1e-7 &&
           !(getenv("TSX_K32") && atoi(getenv("TSX_K32")) == 0);
  // red-black ordering exists on the packed path; it needs an even number of columns per row, and an even
  // number of rows where the rank wraps onto itself in y (a periodic seam between equal colours) -- else zebra rows
  if (s->pc == TSX_PC_REDBLACK) {
    const TsxGeo &g = s->geo;
    // ... and, round 6, on the exact blocks with fp64 iterates (tsx_pcx.hip: what fp32_directions = 0 / pc_coeff_fp16 = 0 get)
    const bool ok = (s->mixed && g.xm % 2 == 0 && g.xm >= 2 && (!g.wrap_y || g.ym % 2 == 0)) || (!s->mixed && tsx_pcx_eligible(s));
    if (!ok) s->pc = TSX_PC_ZEBRA;
  }
  s->pc_split = s->pc == TSX_PC_REDBLACK && s->mixed;  // the colour-split layout belongs to the packed fp32 path
  s->pc_half = false;
  {
    int rc = tsx_dedup_ensure(s);  // shared storage of identical blocks (operator apply and scan preconditioner)
    if (rc) return rc;
    // dense per-cell planes: the operator and the scan passes on the packed path read the shared storage; everything else (exact
    // fp64 preconditioner blocks, one-lane kernels, zebra rows on odd grids, 8_16) reads s->coef
    const bool shared_only = s->dd_on && s->geo.ntop == 2 &&
                             (s->pc == TSX_PC_NONE || (s->mixed && s->pc_split && tsx_pcs_eligible(s)) ||
                              (!s->mixed && s->pc == TSX_PC_REDBLACK && s->coef_bytes == 4));  // (the exact scan passes read the entries too)
    if (!shared_only && (rc = tsx_coef_ensure_dense(s))) return rc;
    if ((rc = tsx_pc_global_agree(s))) return rc;  // several ranks: the preconditioner's halo exchange is on everywhere or nowhere
  }
  if (o->pc_sweeps == 0) {
    // pass count of the Gauss-Seidel orderings, measured on the metric domain, config 2, config 5 and config 4's 252
    // g-points (scripts/ab_env.sh, scripts/sweeps_study.py): 20 passes where the scan kernels run -- a
    // pass (36 us on 256 x 256 x 64) is cheap next to the operator and the vector updates of an iteration (1.3 ms), and 20
    // passes need 6 iterations where 10 need 10 -- and 10 passes with the one-lane-per-column kernels (zebra rows, odd grids)
    const bool scan = s->pc == TSX_PC_REDBLACK && s->mixed && tsx_pcs_eligible(s);
    const bool exact_scan = s->pc == TSX_PC_REDBLACK && !s->mixed;  // tsx_pcx.hip: the pass count measured in round 6 (profiles/r06)
    // round 3: with the side -> top couplings in fp16 (3_10 scan kernels, C16) the residual after 5 iterations of 20 passes
    // sits at 1.02-1.08e-5 on every measured domain -- 22 passes take it below the reference's rtol 1e-5: 5 iterations instead
    // of 6 (256 x 256 x 64: 18.7 -> 16.4 ms; 128 x 128: 6.05 -> 5.29 ms; all blocks distinct: 35.7 -> 31.5 ms; 24 / 26 passes:
    // still 5 iterations, 17.2 / 17.8 ms)
    // 8_16 likewise: 20 passes 1.15e-5 after 5 iterations, 22 passes: 5 iterations, 47.0 -> 41.7 ms (24 passes: 44.6 ms)
    // round 4: with the stop test at the half step the grid was measured again (scripts/sweeps_grid.py, 20 ... 32 passes x ten
    // workloads): 28 passes reach the rule after FOUR iterations where 22 need five -- metric domain 14.2 -> 12.8 ms, 128 x 128
    // 4.86 -> 4.46, 512 x 256 30.1 -> 27.5, every block distinct 20.8 -> 18.5, 8_16 39.1 -> 37.0 ms, config 4 123 -> 125 g-points/s,
    // a 128 x 64 shard 3.55 -> 3.36 ms; slower on three of the ten (full cloud cover, cover 0.6 of another seed, 64 x 64 columns:
    // BiCGStab's iteration counts are integers); 24 / 26 passes: still five iterations, 30 / 32: four, more expensive ones
    const int auto_scan = 27;
    o->pc_sweeps = (s->pc == TSX_PC_REDBLACK || s->pc == TSX_PC_ZEBRA) ? (scan ? auto_scan : (exact_scan ? 19 : 9)) : 1;
    s->pc_sweeps = o->pc_sweeps;
  }
  if (o->pc != TSX_PC_NONE) {
    int rc = tsx_pc_ensure_buffers(s);
    if (rc) return rc;
    if (s->mixed && (rc = tsx_pc_ensure_half(s))) return rc;
  }
  return TSX_OK;
}

extern "C" int tsx_diff_solve(tsx_solver *s, const double *b, double *x, int where, const tsx_ksp_opts *opts,
                              tsx_ksp_result *res) {
  ARGCHK(s && b && x, "tsx_diff_solve: null argument");
  if (!s->have_coeffs) {
    tsx_set_error("tsx_diff_solve: call tsx_diff_set_coeffs first");
    return TSX_ERR_STATE;
  }
  tsx_ksp_opts o;
  {
    int rc = prepare_ksp(s, opts, &o);
    if (rc) return rc;
  }
  return s->geo.ntop == 2 ? diff_solve_t<2, 4>(s, b, x, where, &o, res) : diff_solve_t<8, 4>(s, b, x, where, &o, res);
}

// ------------------------------------------------------------------------------------------------
template <int NTOP, int NSIDE>
static int pc_apply_t(tsx_solver *s, const double *v, double *z, int where, bool mixed) {
  const TsxGeo &g = s->geo;
  const size_t nb = (size_t)g.N * sizeof(double);
  int rc = ensure_stage(s);
  if (rc) return rc;
  const double *vd = v;
  double *zd = z;
  if (where == TSX_HOST) {
    HIPCHK(hipMemcpyAsync(s->stage_a, v, nb, hipMemcpyHostToDevice, s->stream));
    vd = s->stage_a;
    zd = s->stage_b;
  }
  if ((rc = import_vec<NTOP, NSIDE>(s, vd, s->vp))) return rc;
  if (mixed) {  // the solver's default path: fp32 directions from the packed fp16 blocks, widened for the export
    if ((rc = tsx_pc_narrow(s, s->vp))) return rc;
    if ((rc = tsx_pc_apply(s, s->vp, s->vsh, true, false))) return rc;
    if ((rc = tsx_pc_widen(s, (const float *)s->vsh, s->vph))) return rc;
  } else if ((rc = tsx_pc_apply(s, s->vp, s->vph, false, false))) {
    return rc;
  }
  if ((rc = export_vec<NTOP, NSIDE>(s, s->vph, zd))) return rc;
  if (where == TSX_HOST) HIPCHK(hipMemcpyAsync(z, s->stage_b, nb, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  if (s->flow_state) {  // the flow kernel may have run: did one of its bounded waits expire?
    int fe = 0;
    HIPCHK(hipMemcpy(&fe, &s->scal->flow_err, sizeof(int), hipMemcpyDeviceToHost));
    if (fe) {
      tsx_set_error("preconditioner flow kernel: a wait for a neighbour tile's progress word expired");
      return TSX_ERR_HIP;
    }
  }
  return TSX_OK;
}

extern "C" int tsx_diff_pc_apply(tsx_solver *s, const double *v, double *z, int where, int pc, int pc_sweeps, int mixed) {
  ARGCHK(s && v && z, "tsx_diff_pc_apply: null argument");
  ARGCHK(pc >= TSX_PC_COLUMN && pc <= TSX_PC_REDBLACK && pc_sweeps >= 1 && pc_sweeps <= 32, "tsx_diff_pc_apply: bad preconditioner");
  if (!s->have_coeffs) {
    tsx_set_error("tsx_diff_pc_apply: call tsx_diff_set_coeffs first");
    return TSX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(s->device));
  s->pc = pc;
  s->pc_sweeps = pc_sweeps;
  if (s->pc == TSX_PC_REDBLACK) {  // same eligibility rule as prepare_ksp
    const TsxGeo &g = s->geo;
    if (!((mixed && g.xm % 2 == 0 && g.xm >= 2 && (!g.wrap_y || g.ym % 2 == 0)) || (!mixed && tsx_pcx_eligible(s)))) s->pc = TSX_PC_ZEBRA;
  }
  s->pc_split = s->pc == TSX_PC_REDBLACK && mixed;
  s->mixed = mixed != 0;
  int rc = tsx_dedup_ensure(s);
  if (rc) return rc;
  if ((rc = tsx_coef_ensure_dense(s))) return rc;  // a test hook: every variant may run, some read the dense planes
  if ((rc = tsx_pc_global_agree(s))) return rc;
  if ((rc = tsx_pc_ensure_buffers(s))) return rc;
  s->pc_half = false;
  if (mixed && (rc = tsx_pc_ensure_half(s))) return rc;
  return s->geo.ntop == 2 ? pc_apply_t<2, 4>(s, v, z, where, mixed != 0) : pc_apply_t<8, 4>(s, v, z, where, mixed != 0);
}

// ------------------------------------------------------------------------------------------------
extern "C" int tsx_algorithmic_bytes(const tsx_solver *s, int kernel, double *bytes) {
  ARGCHK(s && bytes, "tsx_algorithmic_bytes: null");
  const TsxGeo &g = s->geo;
  const double sc = s->coef_bytes ? s->coef_bytes : 4, sv = 8;
  const double Nc = (double)g.Nc, N = (double)g.N;
  // entries the operator reads (bit-identical blocks) and entries the preconditioner reads (the near grouping where it is on)
  const double nent_op = (double)s->dd_nent, nent = (double)((s->dd_on || s->dd_pc) ? s->pc_nent : s->dd_nent);
  const bool dd_op = s->dd_on;                    // the operator reads shared blocks only where they are bit-identical
  const bool dd = s->dd_on || s->dd_pc;            // the preconditioner also where they are near-identical (tsx_dedup.hip)
  // B_spmv = Nc*D^2*sc + 2*N*sv ; B_iter = 2*B_spmv + 16*N*sv (every cell's block stored: kernels 10, 11).
  // With shared storage of identical blocks (tsx_dedup.hip) the operator's least traffic is every distinct block once, a
  // 4-byte index per cell and the two vectors: kernel 0 reports the bytes of the storage format in use.
  const double bspmv_full = Nc * g.D * g.D * sc + 2.0 * N * sv;
  const double bspmv = dd_op ? nent_op * g.D * g.D * 4.0 + Nc * 4.0 + 2.0 * N * sv : bspmv_full;
  if (kernel == 0) *bytes = bspmv;
  else if (kernel == 10) *bytes = bspmv_full;
  else if (kernel == 1 || kernel == 11) *bytes = 2.0 * bspmv_full + 16.0 * N * sv;
  else if (kernel == 3 || kernel == 2) {
    // the red-black passes on the packed blocks (tsx_kernels_pcs.hpp), bytes per cell of the pass's colour {Gauss-Seidel
    // pass, first pass (no neighbours), fp32 pass of the first colour, last pass (both colours' result in the Krylov
    // layout)} and per distinct block per pass when the per-block records are shared:
    //   3_10, every cell's records: 8 records x 16 B + rhs 10 x 4 B + 4 neighbour records x 4 B (bf16 pairs) + 4 x 4 B
    //         stored = 200;  first: 3 records + rhs + stores = 104;  fp32 pass 224;  last 320
    //   3_10, shared: record 0 (the column recurrence) stays per cell, records 1..7 per distinct block behind the index:
    //         16 (the intermediate passes' copy carries the block index in the word of A_k, which they never use) + 40 + 16
    //         + 16 = 88 (+112 per block);  first 72 (+32);  fp32 pass 16 + 4 + ... = 116;  last 212
    //   8_16, every cell's records: 12 recurrence records (14 in the fp32 passes) + 16 block records (8 in the first pass)
    //         + rhs 16 x 4 B + 4 neighbour records + 4 stored = 544;  first 400;  fp32 624;  last 768
    //   8_16, shared: the 16 block records per distinct block: 292 (+256);  first 276 (+128);  fp32 372;  last 516
    //   recurrence records shared too (tsx_records_share; npid distinct ones): a cell reads a 4-byte index instead of them
    //         3_10 intermediate passes: record 0 (16 B) -> 4 B (+16 per distinct record and pass);
    //         8_16 all passes: 12 (14) records -> 4 B (+192 / 224 per distinct set and pass)
    const bool h = g.ntop == 8;
    const double cell[2][2][4] = {{{200, 104, 224, 320}, {88, 72, 116, 212}}, {{544, 400, 624, 768}, {292, 276, 372, 516}}};
    // 3_10 with the side -> top couplings in fp16 (C16): one record more per cell / per distinct block in the passes that read them
    // (8_16: four records more, 64 B)
    const double c16 = s->coef_h_c16 ? (h ? 64.0 : 16.0) : 0.0;
    const double ent[2][2] = {{112.0 + c16, 32}, {256.0 + c16, 128}};
    double c[4] = {cell[h][dd][0], cell[h][dd][1], cell[h][dd][2], cell[h][dd][3]};
    if (!dd) c[0] += c16, c[2] += c16, c[3] += c16;  // every pass with neighbours reads the extra record per cell
    const double half = 0.5 * Nc, e_gs0 = dd ? nent * ent[h][0] : 0.0, e_first0 = dd ? nent * ent[h][1] : 0.0;
    double r_gs = 0.0, r_f32 = 0.0;  // bytes of the shared recurrence table per pass
    if (dd && s->pcr_on) {
      const double np = (double)s->pcr_n;
      if (h) {
        c[0] -= 188.0, c[1] -= 188.0, c[2] -= 220.0, c[3] -= 220.0;
        r_gs = np * 192.0, r_f32 = np * 224.0;
      } else {
        c[0] -= 12.0, c[1] -= 12.0;
        r_gs = np * 16.0;
      }
    }
    const double e_gs = e_gs0 + r_gs, e_first = e_first0 + r_gs, e_f32 = e_gs0 + r_f32;
    const int P = s->pc_sweeps > 0 ? s->pc_sweeps + 1 : 22;
    const double ngs = P > 3 ? P - 3 : 0;
    // bf16 right-hand side of the intermediate passes (tsx_k_pcs_rb RQ): a colour's first visit leaves 5 words (+20 B), the
    // later intermediate visits read 20 B instead of 40 B
    const bool r16 = tsx_pcs_rhs16(s) && P >= 6;
    const double w16 = h ? 32.0 : 20.0;  // the bf16-pair words of a cell
    const double gs_b = r16 ? c[0] - w16 : c[0];
    if (kernel == 3) *bytes = gs_b * half + e_gs;
    else if (r16) *bytes = ((c[1] + w16) + (c[0] + w16) + gs_b * (ngs - 1.0) + c[2] + c[3]) * half + e_first + ngs * e_gs + 2.0 * e_f32;
    else *bytes = (c[1] + c[0] * ngs + c[2] + c[3]) * half + e_first + ngs * e_gs + 2.0 * e_f32;
  } else if (kernel == 4) {
    // the flow kernel (tsx_k_pcs_flow): p1 - p0 intermediate passes in one launch; with granules (TsxGran) the four records a cell
    // stores and the four it reads are 8 bytes each instead of 4
    if (!s->flow_last[0]) {
      tsx_set_error("tsx_algorithmic_bytes: kernel 4: the last application of M^-1 did not use the flow kernel");
      return TSX_ERR_STATE;
    }
    double pass = 0.0;
    int rc = tsx_algorithmic_bytes(s, 3, &pass);
    if (rc) return rc;
    if (s->flow_last[5]) pass += 32.0 * 0.5 * Nc;
    *bytes = pass * (double)(s->flow_last[2] - s->flow_last[1]);
  } else {
    tsx_set_error("tsx_algorithmic_bytes: kernel must be 0..4, 10 or 11");
    return TSX_ERR_ARG;
  }
  return TSX_OK;
}

template <int NTOP, int NSIDE>
static int bench_kernel_t(tsx_solver *s, int kernel, int reps, float *avg_ms) {
  int rc;
  if (kernel == 0) {
    if ((rc = launch_spmv<NTOP, NSIDE, 0>(s, (const double *)s->vp, s->vv, (const double *)nullptr, false))) return rc;  // warm
    HIPCHK(hipEventRecord(s->ev0, s->stream));
    for (int q = 0; q < reps; ++q)
      if ((rc = launch_spmv<NTOP, NSIDE, 0>(s, (const double *)s->vp, s->vv, (const double *)nullptr, false))) return rc;
    HIPCHK(hipEventRecord(s->ev1, s->stream));
  } else if (kernel == 2 || kernel == 3 || kernel == 4) {
    // the default preconditioner on the fp32 right-hand side p32: 2 = one application (pc_sweeps + 1 half-grid passes),
    // 3 = one intermediate Gauss-Seidel pass of the scan kernels as a launch of its own, 4 = the flow kernel: the intermediate
    // passes of an application in one launch, as the Krylov loop issues it (the vector update before it has left the bf16 words)
    tsx_ksp_opts o, ou;
    if ((rc = prepare_ksp(s, nullptr, &o))) return rc;
    (void)ou;
    s->pc_rhs = s->p32;
    if (kernel >= 3 && !s->coef_h_scan) {
      tsx_set_error("tsx_bench_kernel: kernel 3 / 4 needs the scan preconditioner (3_10, red-black, TSX_PC_SCAN != 0)");
      return TSX_ERR_UNSUPPORTED;
    }
    if ((rc = tsx_pc_apply(s, s->vp, s->vph, true, false))) return rc;  // warm (and it leaves the bf16 words)
    int fl[8];
    memcpy(fl, s->flow_last, sizeof(fl));
    if (kernel == 4 && !fl[0]) {
      tsx_set_error("tsx_bench_kernel: kernel 4: this configuration does not run the flow kernel");
      return TSX_ERR_UNSUPPORTED;
    }
    if (kernel == 4 && fl[0] == 2) {  // (timing the variant without faces on a shard with real neighbours is another kernel)
      tsx_set_error("tsx_bench_kernel: kernel 4: the last application ran the flow kernel with rank faces inside; it cannot be timed alone");
      return TSX_ERR_UNSUPPORTED;
    }
    if (kernel == 4 && fl[1] > 1) {  // as inside a solve: pass 0 alone precedes it
      fl[1] = 1;
      if ((rc = tsx_pcs_flow(s, fl[3], fl[1], fl[2], nullptr))) return rc;
    }
    HIPCHK(hipEventRecord(s->ev0, s->stream));
    for (int q = 0; q < reps; ++q) {
      if (kernel == 2) rc = tsx_pc_apply(s, s->vp, s->vph, true, false);
      else if (kernel == 3) rc = tsx_pcs_pass(s, 2 + (q & 1), 0, (float *)s->vph, nullptr, tsx_pcs_rhs16(s) ? 2 : 0);
      else rc = tsx_pcs_flow(s, fl[3], fl[1], fl[2], nullptr);
      if (rc) return rc;
    }
    HIPCHK(hipEventRecord(s->ev1, s->stream));
  } else {
    // iterations on whatever state the vectors hold; scalars are neutralised so nothing diverges/stops
    tsx_ksp_opts o;
    tsx_default_ksp_opts(&o);
    o.rtol = 0;
    o.atol = 0;
    o.dtol = 1e300;
    o.maxit = 1 << 30;
    HIPCHK(hipMemsetAsync(s->vx, 0, sizeof(double) * s->geo.N, s->stream));
    if ((rc = krylov_begin<NTOP, NSIDE>(s, &o))) return rc;
    if ((rc = enqueue_iteration<NTOP, NSIDE>(s, true))) return rc;
    if (kernel == 5) {
      // the same iterations replayed from a hipGraph (measurement only, round 6: the previous reviews asked for the number instead of
      // the argument): a second steady-state iteration eagerly (every lazily allocated buffer exists, every host-side flag has its
      // steady value), the third captured from the solver's stream, instantiated, launched `reps` times
      if (s->grid.nranks > 1 || !(s->geo.wrap_x && s->geo.wrap_y)) {
        tsx_set_error("tsx_bench_kernel: kernel 5 (graph replay) is a one-rank measurement");
        return TSX_ERR_UNSUPPORTED;
      }
      if ((rc = enqueue_iteration<NTOP, NSIDE>(s, false))) return rc;
      HIPCHK(hipStreamSynchronize(s->stream));
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      HIPCHK(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
      rc = enqueue_iteration<NTOP, NSIDE>(s, false);
      hipError_t ce = hipStreamEndCapture(s->stream, &graph);
      if (rc) return rc;
      HIPCHK(ce);
      HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      HIPCHK(hipGraphLaunch(exec, s->stream));  // warm
      HIPCHK(hipEventRecord(s->ev0, s->stream));
      for (int q = 0; q < reps; ++q) HIPCHK(hipGraphLaunch(exec, s->stream));
      HIPCHK(hipEventRecord(s->ev1, s->stream));
      HIPCHK(hipStreamSynchronize(s->stream));
      (void)hipGraphExecDestroy(exec);
      (void)hipGraphDestroy(graph);
    } else {
      HIPCHK(hipEventRecord(s->ev0, s->stream));
      for (int q = 0; q < reps; ++q)
        if ((rc = enqueue_iteration<NTOP, NSIDE>(s, false))) return rc;
      HIPCHK(hipEventRecord(s->ev1, s->stream));
    }
  }
  HIPCHK(hipStreamSynchronize(s->stream));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
  *avg_ms = ms / (float)reps;
  return TSX_OK;
}

extern "C" int tsx_bench_kernel(tsx_solver *s, int kernel, int reps, float *avg_ms) {
  ARGCHK(s && avg_ms && reps >= 1, "tsx_bench_kernel: bad argument");
  ARGCHK(kernel >= 0 && kernel <= 5, "tsx_bench_kernel: kernel must be 0..5");
  if (!s->have_coeffs) {
    tsx_set_error("tsx_bench_kernel: call tsx_diff_set_coeffs first");
    return TSX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(s->device));
  {
    int rc = tsx_dedup_ensure(s);
    if (rc) return rc;
  }
  return s->geo.ntop == 2 ? bench_kernel_t<2, 4>(s, kernel, reps, avg_ms) : bench_kernel_t<8, 4>(s, kernel, reps, avg_ms);
}

extern "C" int tsx_dedup_info(tsx_solver *s, int32_t *on, int64_t *nent) {
  ARGCHK(s && on && nent, "tsx_dedup_info: null");
  HIPCHK(hipSetDevice(s->device));
  int rc = tsx_dedup_ensure(s);
  if (rc) return rc;
  // bit 0: bit-identical blocks shared (operator and preconditioner); bit 1: near-identical blocks grouped for the
  // preconditioner (on top of bit 0, or alone where nothing is bit-identical)
  // bit 2: the grouping is the previous coefficient set's, found still valid for this set's LUT coordinates (tsx_dedup_from_coords)
  // bit 3 (round 6): the grouping of the cells by their packed recurrence records (tsx_records_share) was the previous set's as well
  *on = (s->dd_on ? 1 : 0) | (s->dd_pc ? 2 : 0) | ((s->dd_on && s->dd_from_coords && s->dd_reused) ? 4 : 0) |
        ((s->pcr_on && s->pcr_reused) ? 8 : 0);
  *nent = s->dd_on ? s->dd_nent : (s->dd_pc ? s->pc_nent : s->dd_nent);
  return TSX_OK;
}

extern "C" int tsx_flow_info(const tsx_solver *s, int32_t *info8) {
  ARGCHK(s && info8, "tsx_flow_info: null");
  for (int q = 0; q < 8; ++q) info8[q] = s->flow_last[q];
  return TSX_OK;
}

extern "C" int tsx_pc_info(const tsx_solver *s, int32_t *pc, int32_t *pc_sweeps, int32_t *scan) {
  ARGCHK(s && pc && pc_sweeps && scan, "tsx_pc_info: null");
  *pc = s->pc;
  *pc_sweeps = s->pc_sweeps;
  *scan = s->coef_h_scan ? (s->pcr_on ? 3 : 1) : 0;  // bit 1: identical recurrence records are stored once (tsx_records_share)
  return TSX_OK;
}

// ---- bandwidth probes: what this device's memory system delivers to plain streaming kernels, as a ceiling to report the
// rooflines against beside the nominal 8 TB/s (MI355X_MICROARCH.md: about 6.3 TB/s achievable).  U independent 16-byte
// accesses per lane in flight, a capped grid with a grid-stride loop, optionally non-temporal; the best variant counts.
typedef float tsx_f4v __attribute__((ext_vector_type(4)));  // a native vector: the non-temporal builtins take no HIP_vector_type
template <int U, bool NT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_copy16u(long long n, const tsx_f4v *__restrict__ a, tsx_f4v *__restrict__ b) {
  const long long stride = (long long)gridDim.x * TSX_BLOCK;
  long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x;
  for (; q + (U - 1) * stride < n; q += U * stride) {
    tsx_f4v v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&a[q + u * stride]) : a[q + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], &b[q + u * stride]);
      else b[q + u * stride] = v[u];
    }
  }
  for (; q < n; q += stride) b[q] = a[q];
}
template <int U, bool NT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_read16u(long long n, const tsx_f4v *__restrict__ a, float *__restrict__ out) {
  const long long stride = (long long)gridDim.x * TSX_BLOCK;
  long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x;
  float acc = 0.0f;
  for (; q + (U - 1) * stride < n; q += U * stride) {
    tsx_f4v v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&a[q + u * stride]) : a[q + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
  }
  for (; q < n; q += stride) acc += a[q][0];
  if (acc == 123.456f) out[blockIdx.x] = acc;  // keeps the loads alive; the buffer holds a different constant
}

// best copy (read + write bytes) and best read rate over the variants, GB/s; variant ids for the record
static int probe_bandwidth(tsx_solver *s, size_t bytes, int reps, double *copy_gbps, double *read_gbps, int *copy_variant,
                           int *read_variant) {
  HIPCHK(hipSetDevice(s->device));
  TsxDevTmp A, B;
  HIPCHK(A.alloc(bytes));
  HIPCHK(B.alloc(bytes));
  HIPCHK(hipMemsetAsync(A.p, 1, bytes, s->stream));
  HIPCHK(hipMemsetAsync(B.p, 0, bytes, s->stream));
  const long long n = (long long)(bytes / 16);
  const tsx_f4v *a = A.as<tsx_f4v>();
  tsx_f4v *b = B.as<tsx_f4v>();
  const int grids[3] = {2048, 4096, 16384};
  double best_c = 0, best_r = 0;
  int vc = -1, vr = -1;
  auto timed = [&](auto launch, double moved, double *best, int *bv, int id) -> int {
    launch();  // warm
    HIPCHK(hipEventRecord(s->ev0, s->stream));
    for (int q = 0; q < reps; ++q) launch();
    HIPCHK(hipEventRecord(s->ev1, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    const double g = moved * reps / (ms * 1e-3) / 1e9;
    if (g > *best) {
      *best = g;
      *bv = id;
    }
    return TSX_OK;
  };
  int rc;
  for (int gi = 0; gi < 3; ++gi) {
    const int nb = (int)(n / TSX_BLOCK < grids[gi] ? (n / TSX_BLOCK > 0 ? n / TSX_BLOCK : 1) : grids[gi]);
#define TSX_PROBE(U, NT, ID)                                                                                                        \
  if ((rc = timed([&] { hipLaunchKernelGGL((tsx_k_copy16u<U, NT>), dim3(nb), dim3(TSX_BLOCK), 0, s->stream, n, a, b); },            \
                  2.0 * (double)(n * 16), &best_c, &vc, gi * 10 + ID)))                                                             \
    return rc;                                                                                                                      \
  if ((rc = timed([&] { hipLaunchKernelGGL((tsx_k_read16u<U, NT>), dim3(nb), dim3(TSX_BLOCK), 0, s->stream, n, a, (float *)b); },   \
                  (double)(n * 16), &best_r, &vr, gi * 10 + ID)))                                                                   \
    return rc;
    TSX_PROBE(1, false, 0)
    TSX_PROBE(4, false, 1)
    TSX_PROBE(8, false, 2)
    TSX_PROBE(4, true, 3)
    TSX_PROBE(8, true, 4)
#undef TSX_PROBE
  }
  HIPCHK(hipGetLastError());
  *copy_gbps = best_c;
  *read_gbps = best_r;
  if (copy_variant) *copy_variant = vc;
  if (read_variant) *read_variant = vr;
  return TSX_OK;
}

extern "C" int tsx_probe_copy_bandwidth(tsx_solver *s, size_t bytes, int reps, double *gbps) {
  ARGCHK(s && gbps && reps >= 1 && bytes >= 16, "tsx_probe_copy_bandwidth: bad argument");
  double r = 0;
  return probe_bandwidth(s, bytes, reps, gbps, &r, nullptr, nullptr);
}
// out4: best copy GB/s (bytes read + written), best read GB/s, and the variants that gave them (grid index * 10 + kernel id:
// kernel 0 one access per lane, 1 / 2 four / eight in flight, 3 / 4 the same non-temporal; grids 2048, 4096, 16384 workgroups)
extern "C" int tsx_probe_bandwidth(tsx_solver *s, size_t bytes, int reps, double *out4) {
  ARGCHK(s && out4 && reps >= 1 && bytes >= 16, "tsx_probe_bandwidth: bad argument");
  int vc = -1, vr = -1;
  int rc = probe_bandwidth(s, bytes, reps, &out4[0], &out4[1], &vc, &vr);
  out4[2] = vc;
  out4[3] = vr;
  return rc;
}

// ---- log events + roctx ranges (TsxLog, tsx_internal.hpp).  roctx comes from librocprofiler-sdk-roctx (ROCm 7; libroctx64 before
// it), bound at run time on first use: libtsx links neither, and without the library the ranges are no-ops.
static const char *const kLogNames[TSX_EV_COUNT] = {"set_optprop", "get_coeff_diff2diff", "get_coeff_dir2dir", "compute_Edir", "solve_Mdir",
                                                   "setup_diff_src", "compute_Ediff", "setup_Mdiff", "solve_Mdiff", "compute_absorption",
                                                   "get_result"};
struct TsxRoctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  TsxRoctx() {
    void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
    pop = (int (*)())dlsym(h, "roctxRangePop");
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
static TsxRoctx &tsx_roctx() {
  static TsxRoctx r;
  return r;
}
static void tsx_log_retire(tsx_solver *s, bool wait) {
  TsxLog *L = s->log;
  size_t keep = 0;
  for (size_t q = 0; q < L->pending.size(); ++q) {
    TsxLogPending &p = L->pending[q];
    bool ready = hipEventQuery(p.b) == hipSuccess;
    if (!ready && wait) ready = hipEventSynchronize(p.b) == hipSuccess;
    float ms = 0;
    if (ready && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      L->ms[p.ev] += ms;
      L->pool.push_back(p.a);
      L->pool.push_back(p.b);
    } else if (ready) {  // (an event pair that cannot be read: drop it)
      (void)hipEventDestroy(p.a);
      (void)hipEventDestroy(p.b);
    } else {
      L->pending[keep++] = p;
    }
  }
  L->pending.resize(keep);
  (void)hipGetLastError();  // hipEventQuery's hipErrorNotReady is not an error of the caller
}
static hipEvent_t tsx_log_event(TsxLog *L) {
  hipEvent_t e = nullptr;
  if (!L->pool.empty()) {
    e = L->pool.back();
    L->pool.pop_back();
  } else if (hipEventCreate(&e) != hipSuccess) {
    e = nullptr;
  }
  return e;
}
void tsx_log_begin(tsx_solver *s, int ev, hipEvent_t *a) {
  TsxRoctx &r = tsx_roctx();
  if (r.push) r.push(kLogNames[ev]);
  *a = tsx_log_event(s->log);
  if (*a) (void)hipEventRecord(*a, s->stream);
}
void tsx_log_end(tsx_solver *s, int ev, hipEvent_t a) {
  TsxLog *L = s->log;
  TsxRoctx &r = tsx_roctx();
  L->count[ev] += 1;
  hipEvent_t b = a ? tsx_log_event(L) : nullptr;
  if (b && hipEventRecord(b, s->stream) == hipSuccess) {
    L->pending.push_back({ev, a, b});
    if (L->pending.size() > 256) tsx_log_retire(s, false);
  }
  if (r.pop) r.pop();
}
static void tsx_log_free(tsx_solver *s) {
  if (!s->log) return;
  tsx_log_retire(s, true);
  for (hipEvent_t e : s->log->pool) (void)hipEventDestroy(e);
  delete s->log;
  s->log = nullptr;
}
extern "C" int tsx_log_enable(tsx_solver *s, int on) {
  ARGCHK(s, "tsx_log_enable: null");
  HIPCHK(hipSetDevice(s->device));
  if (on && !s->log) s->log = new TsxLog();
  if (!on) tsx_log_free(s);
  return TSX_OK;
}
extern "C" int tsx_log_get(tsx_solver *s, int32_t *nevents, const char **names, int64_t *counts, double *ms) {
  ARGCHK(s && nevents, "tsx_log_get: null");
  *nevents = TSX_EV_COUNT;
  if (!s->log) {
    tsx_set_error("tsx_log_get: log events are off (tsx_log_enable, or TSX_LOG=1 at tsx_create)");
    return TSX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  tsx_log_retire(s, true);
  for (int q = 0; q < TSX_EV_COUNT; ++q) {
    if (names) names[q] = kLogNames[q];
    if (counts) counts[q] = s->log->count[q];
    if (ms) ms[q] = s->log->ms[q];
  }
  return TSX_OK;
}

// ---- diagnostics: a translation unit's code as it sits in device memory (TSX_CODE_PROBE, tsx_host.hpp).  unit 0..7 = api, spmv310,
// spmv816, pc, pcs, pcsflow, dedup, peer (the order of the code objects in libtsx.so is the link order, scripts/code_verify.py finds
// them by the probe's symbol).  Copies nwords 32-bit words from (probe's pc + delta) to host_out and returns the pc in *pc_out; with
// nwords = 0 only the pc.  The caller is responsible for the range lying inside the loaded code object.
extern "C" int tsx_debug_code_read(int device, int unit, long long delta, long long nwords, void *host_out, unsigned long long *pc_out) {
  ARGCHK(unit >= 0 && unit < 8 && nwords >= 0 && pc_out && (nwords == 0 || host_out), "tsx_debug_code_read: bad arguments");
  if (device >= 0) HIPCHK(hipSetDevice(device));
  typedef int (*probe_fn)(long long, long long, unsigned *, unsigned long long *, hipStream_t);
  static const probe_fn probes[8] = {tsx_code_probe_api, tsx_code_probe_spmv310, tsx_code_probe_spmv816, tsx_code_probe_pc,
                                     tsx_code_probe_pcs, tsx_code_probe_pcsflow, tsx_code_probe_dedup, tsx_code_probe_peer};
  TsxDevTmp out, pc;
  HIPCHK(out.alloc(sizeof(unsigned) * (size_t)(nwords > 0 ? nwords : 1)));
  HIPCHK(pc.alloc(sizeof(unsigned long long)));
  if (probes[unit](delta, nwords, out.as<unsigned>(), pc.as<unsigned long long>(), nullptr)) {
    tsx_set_error("tsx_debug_code_read: launch failed");
    return TSX_ERR_HIP;
  }
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(pc_out, pc.p, sizeof(unsigned long long), hipMemcpyDeviceToHost));
  if (nwords > 0) HIPCHK(hipMemcpy(host_out, out.p, sizeof(unsigned) * (size_t)nwords, hipMemcpyDeviceToHost));
  return TSX_OK;
}

#include "tsx_pipeline_api.inc"
#include "tsx_seam_api.inc"

TSX_CODE_PROBE(api)  // tsx_host.hpp: this unit's code object as it sits in device memory (diagnostics)
