// tsx_host.hpp -- host-side helpers shared by the translation units of libtsx (tsx_api.hip, tsx_spmv_*.hip, tsx_pc.hip)
#pragma once
#include <stdlib.h>

#include <string>
#include <type_traits>

#include "tsx_dev.hpp"

#ifndef TSX_DEFAULT_CPT
#define TSX_DEFAULT_CPT 2
#endif

#define HIPCHK(call)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      tsx_set_error(std::string(#call) + ": " + hipGetErrorString(e_) + " @" + __FILE__ + ":" + std::to_string(__LINE__)); \
      return e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? TSX_ERR_NO_DEVICE : TSX_ERR_HIP; \
    }                                                                                     \
  } while (0)

#define ARGCHK(cond, msg)          \
  do {                             \
    if (!(cond)) {                 \
      tsx_set_error(msg);          \
      return TSX_ERR_ARG;          \
    }                              \
  } while (0)

static inline int grid_for(long long n, int cap = 2048) {
  long long b = (n + TSX_BLOCK - 1) / TSX_BLOCK;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// TSX_SPMV_CPT=1|2 selects cells per thread (default 2 when xm is even)
static inline int spmv_cpt(const tsx_solver *s) {
  const char *e = getenv("TSX_SPMV_CPT");  // read per call: tests switch it
  const int env = e ? atoi(e) : 0;
  // with shared block storage (tsx_dedup.hip) the coefficient loads are 4-byte gathers anyway: one cell per thread measured
  // 5-8 % faster there (more waves, fewer registers)
  // ... and with the fp32 recurrence vectors (round 3: every block stored, 256 x 256 x 64: 0.506 vs 0.520 ms per apply, 191 vs 186 M cells/s)
  int want = env > 0 ? env : ((s->dd_on || s->k32) ? 1 : TSX_DEFAULT_CPT);
  if (want > 2) want = 2;
  while (want > 1 && (s->geo.xm % want) != 0) want >>= 1;
  return want;
}

#define TSX_FRAME_BLOCKS 256
// groups of the frame (cells whose gather reads a received face), see tsx_k_spmv_w
static inline long long frame_groups(const TsxGeo &g, int cpt) {
  const int gx = g.xm / cpt;
  const int nfull = g.wrap_y ? 0 : (g.ym >= 2 ? 2 : 1);
  const int ex = g.wrap_x ? 0 : (gx >= 2 ? 2 : 1);
  return (long long)g.Nz * (nfull * gx + (g.ym - nfull) * ex);
}
// Interior / frame split around an exchange (the operator's and the passes'), the exchange on comm_stream between two events.
// TSX_OVERLAP=0|1 decides; unset: only where an exchange is a library call or a host round trip (RCCL, the callbacks).  With
// the peer transport, and with self neighbours (device copies), an exchange is a few microseconds of kernels on the solver
// stream, and the split -- a second launch with the pass's latency floor plus two cross-stream event waits -- costs more than
// it hides at every size (scripts/shard_study.py, one rank with itself as its four neighbours: 128 x 64 columns 15.1 -> 7.1 ms
// per solve, 128 x 256 24.5 -> 11.6, 256 x 256 30.6 -> 18.3, 512 x 512 82.2 -> 66.7 ms)
bool tsx_peer_ready(const tsx_solver *s);  // tsx_peer.hip
static inline bool tsx_overlap(const tsx_solver *s) {
  if (s->overlap_env >= 0) return s->overlap_env != 0;
  if (tsx_peer_ready(s)) return false;
  return s->xchg_cb != nullptr || s->comm_ready;
}
static inline bool spmv_split(const tsx_solver *s) { return tsx_overlap(s) && !(s->geo.wrap_x && s->geo.wrap_y); }

static inline int spmv_nblocks(const tsx_solver *s) {
  const int cpt = spmv_cpt(s);
  const int nbmain = grid_for(s->geo.Nc / cpt, TSX_MAX_PARTIAL_BLOCKS - TSX_FRAME_BLOCKS);
  return spmv_split(s) ? nbmain + grid_for(frame_groups(s->geo, cpt), TSX_FRAME_BLOCKS) : nbmain;
}

// ---- device memory comes from the library's pool (tsx_pool.hip): driver allocations are taken once, quarantined until their
// contents are proven stable, and sub-allocated from then on.  Same contracts as hipMalloc / hipFree (the free synchronises the device).
hipError_t tsx_dev_malloc_bytes(void **out, size_t bytes);
hipError_t tsx_dev_free(void *p);
hipError_t tsx_dev_quarantine(void *p, size_t bytes);
void tsx_dev_reserve(size_t bytes);  // a hint: one slab for what a solver is about to allocate piece by piece  // for driver allocations of another kind (the uncached peer mailbox), in place
template <typename T>
static inline hipError_t tsx_dev_malloc(T **p, size_t bytes) {
  return tsx_dev_malloc_bytes((void **)p, bytes);
}

// scratch device buffer that is released on every exit path (the HIPCHK / ARGCHK macros return early)
struct TsxDevTmp {
  void *p = nullptr;
  TsxDevTmp() = default;
  TsxDevTmp(const TsxDevTmp &) = delete;
  TsxDevTmp &operator=(const TsxDevTmp &) = delete;
  ~TsxDevTmp() {
    if (p) (void)tsx_dev_free(p);
  }
  hipError_t alloc(size_t bytes) { return tsx_dev_malloc(&p, bytes); }
  template <typename T> T *as() const { return static_cast<T *>(p); }
};

// ---- code probe (diagnostics): every translation unit is a code object of its own inside libtsx.so (no -fgpu-rdc); the probe
// kernel of a unit reports its own program counter and copies `nwords` 32-bit words from pc + delta to `out`, so that a host tool
// that knows the unit's ELF (scripts/code_verify.py: symbol table + the s_getpc_b64 inside the probe) can compare the code AS IT
// SITS IN DEVICE MEMORY with the bytes in the file (tsx_debug_code_read, tsx_api.hip).  Round 6: the hunt for the rare wrong
// result of the four-process pipeline test asked whether a process can run a damaged copy of a kernel.
#define TSX_CODE_PROBE(TU)                                                                                                     \
  extern "C" __global__ void tsx_k_code_probe_##TU(long long delta, long long nwords, unsigned *out, unsigned long long *pc_out) { \
    unsigned long long pc;                                                                                                     \
    asm volatile("s_getpc_b64 %0" : "=s"(pc));                                                                              \
    if (blockIdx.x == 0 && threadIdx.x == 0) *pc_out = pc;                                                                     \
    const unsigned *src = (const unsigned *)(pc + delta);                                                                      \
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (long long)gridDim.x * blockDim.x)   \
      out[i] = __builtin_nontemporal_load(src + i);                                                                            \
  }                                                                                                                            \
  int tsx_code_probe_##TU(long long delta, long long nwords, unsigned *out, unsigned long long *pc_out, hipStream_t st) {        \
    hipLaunchKernelGGL(tsx_k_code_probe_##TU, dim3(nwords > 0 ? 256 : 1), dim3(256), 0, st, delta, nwords, out, pc_out);      \
    return hipGetLastError() == hipSuccess ? 0 : 1;                                                                            \
  }
#define TSX_CODE_PROBE_DECL(TU) int tsx_code_probe_##TU(long long, long long, unsigned *, unsigned long long *, hipStream_t);
TSX_CODE_PROBE_DECL(api) TSX_CODE_PROBE_DECL(spmv310) TSX_CODE_PROBE_DECL(spmv816) TSX_CODE_PROBE_DECL(pc) TSX_CODE_PROBE_DECL(pcs)
TSX_CODE_PROBE_DECL(pcsflow) TSX_CODE_PROBE_DECL(dedup) TSX_CODE_PROBE_DECL(peer)

// ---- cross-unit entry points --------------------------------------------------------------------
// face exchange on stream st (RCCL / host-staged callbacks / self copies), tsx_api.hip
int tsx_face_exchange(tsx_solver *s, hipStream_t st);
int tsx_face_exchange_elems(tsx_solver *s, hipStream_t st, size_t elem_bytes);  // the diffuse halo with elements of that size
int tsx_face_exchange_bufs(tsx_solver *s, hipStream_t st, double *const send[4], double *const recv[4], size_t cx, size_t cy);

// operator apply: one translation unit per stream configuration (tsx_spmv_3_10.hip, tsx_spmv_8_16.hip).
// combo = (fused dots, type of x, type of w): the variants the Krylov loop uses
enum TsxSpmvCombo { TSX_SPMV_0DD = 0, TSX_SPMV_1FF, TSX_SPMV_5FD, TSX_SPMV_1DF, TSX_SPMV_1DD, TSX_SPMV_5DD, TSX_SPMV_1FF_YF, TSX_SPMV_5FF_YF };
int tsx_spmv_launch_310(tsx_solver *s, int combo, const void *x, void *y, const void *w, bool in_solve);
int tsx_spmv_launch_816(tsx_solver *s, int combo, const void *x, void *y, const void *w, bool in_solve);
int tsx_halo_update_310(tsx_solver *s, const double *v, bool in_solve);
int tsx_halo_update_816(tsx_solver *s, const double *v, bool in_solve);

template <int FUSE, typename XT, typename WT>
constexpr int tsx_spmv_combo() {
  constexpr bool xf = std::is_same<XT, float>::value, wf = std::is_same<WT, float>::value;
  static_assert(FUSE == 0 || FUSE == 1 || FUSE == 5, "unsupported fused-dot set");
  static_assert(!(FUSE == 0 && (xf || wf)) && !(FUSE == 5 && wf) && !(FUSE == 1 && xf && !wf), "variant not instantiated");
  return FUSE == 0 ? TSX_SPMV_0DD : FUSE == 1 ? (xf ? TSX_SPMV_1FF : (wf ? TSX_SPMV_1DF : TSX_SPMV_1DD)) : (xf ? TSX_SPMV_5FD : TSX_SPMV_5DD);
}
template <int NTOP, int NSIDE, int FUSE, typename XT = double, typename WT = double>
static inline int launch_spmv(tsx_solver *s, const XT *x, double *y, const WT *w, bool in_solve) {
  constexpr int combo = tsx_spmv_combo<FUSE, XT, WT>();
  return NTOP == 2 ? tsx_spmv_launch_310(s, combo, x, y, w, in_solve) : tsx_spmv_launch_816(s, combo, x, y, w, in_solve);
}
// v = A p-hat (fused (rhat, v)) / t = A s-hat (fused (s, t), (t, t)) with fp32 input, shadow vector and RESULT
template <int NTOP, int NSIDE, int FUSE>
static inline int launch_spmv_f32(tsx_solver *s, const float *x, float *y, const float *w, bool in_solve) {
  static_assert(FUSE == 1 || FUSE == 5, "fp32-result variants: FUSE 1 and 5");
  const int combo = FUSE == 1 ? TSX_SPMV_1FF_YF : TSX_SPMV_5FF_YF;
  return NTOP == 2 ? tsx_spmv_launch_310(s, combo, x, y, w, in_solve) : tsx_spmv_launch_816(s, combo, x, y, w, in_solve);
}
template <int NTOP, int NSIDE>
static inline int halo_update(tsx_solver *s, const double *v, bool in_solve) {
  return NTOP == 2 ? tsx_halo_update_310(s, v, in_solve) : tsx_halo_update_816(s, v, in_solve);
}

// preconditioner (tsx_pc.hip): z = M^-1 v, z fp64 (exact blocks, reads v) or fp32 (the solver's fp32 directions on the
// packed reduced-precision blocks; reads the fp32 copy s->v32 of v that tsx_k_pupdate / supdate / residual0 -- or
// tsx_pc_narrow -- left there)
int tsx_pc_apply(tsx_solver *s, const double *v, void *z, bool z_is_float, bool in_solve);
int tsx_pc_ensure_buffers(tsx_solver *s);
int tsx_pc_ensure_half(tsx_solver *s);
int tsx_pc_narrow(tsx_solver *s, const double *a);
int tsx_pc_widen(tsx_solver *s, const float *a, double *o);  // o = (double) a over the N unknowns
int tsx_cell_samples(tsx_solver *s, const double *kabs, const double *ksca, const double *g, const double *dz, double dx);  // tsx_api.hip
// shared storage of bit-identical blocks (tsx_dedup.hip)
int tsx_dedup_ensure(tsx_solver *s);
struct TsxLutDev;
int tsx_dedup_from_coords(tsx_solver *s, const TsxLutDev &L, bool *built);  // LUT path: sharing keyed on the cells' LUT coordinates
int tsx_coef_ensure_dense(tsx_solver *s);  // dense per-cell planes, expanded from the shared entries if the LUT path skipped them
// the red-black preconditioner of 3_10 as a segmented scan over the levels (tsx_pcs.hip); packed layout "S16" in s->coef_h
bool tsx_pcs_eligible(const tsx_solver *s);
int tsx_pc_global_agree(tsx_solver *s);       // tsx_pcs.hip: collective, see there
int tsx_allreduce_host(tsx_solver *s, double *v, int n);  // tsx_api.hip (tsx_pipeline_api.inc)
int tsx_pcs_pack(tsx_solver *s);
int tsx_pcs_apply(tsx_solver *s, float *z, const int *done);
int tsx_pcs_pass(tsx_solver *s, int pass, int mode, float *zfin, const int *done, int rq, int part = 0);
bool tsx_pcs_rhs16(const tsx_solver *s);
// passes [p0, p1) of an application (all of them intermediate Gauss-Seidel passes on the bf16 right-hand side words) as ONE
// launch of tsx_k_pcs_flow (tsx_pcs_flow.hip); tsx_pcs_flow_ok: can this solver's current scan configuration take it
// solves in flight in this process (several solver instances on streams of their own: config 4's spectral loop)
int tsx_active_solves();
bool tsx_pcs_flow_ok(tsx_solver *s, int lseg, int nseg, int cw, bool faces = false);
int tsx_pcs_flow(tsx_solver *s, int cw, int p0, int p1, const int *done, bool faces = false);
unsigned *tsx_pcs_words(const tsx_solver *s);
int tsx_records_share(tsx_solver *s, int R, const uint4 *P);
// the red-black passes on the exact blocks with fp64 iterates, as a segmented scan (tsx_pcx.hip): what fp32_directions = 0 gets
bool tsx_pcx_eligible(const tsx_solver *s);
int tsx_pcx_apply(tsx_solver *s, const double *v, double *z, const int *done);
int tsx_dedup_hash_buffer(tsx_solver *s, unsigned long long **h);
