/*
 * tsx.h -- C-ABI of libtsx: the MI355X-native pprts diffuse/direct back-end.
 *
 * Drop-in boundary for TenStream's pprts hot path.  Every entry point cites the reference
 * interface (tenstream/tenstream, file:line) it replaces or is called from.  Plain C: pointers,
 * sizes, no torch / HIP types in the signatures (streams travel as void*).
 *
 * The seam: inside `pprts()` the reference branches
 *     if (lexplicit_diff) call explicit_ediff(solver, prefix, solver%b, solution%ediff, solution, ierr)
 *     else                call ediff(A, Aperm, ksp, prefix)                   (src/pprts.F90:2794-2813)
 * A third branch (option `-<prefix>hip`) calls tsx_diff_solve through the ISO_C_BINDING shim in
 * tenstream_amd/fortran/m_pprts_hip.F90.
 *
 * Conventions
 *  - Arrays handed over use the reference's own layouts and kinds (ireals = real64):
 *      diffuse vectors  (0:D-1, zs:ze, xs:xe, ys:ye)   dof fastest        src/pprts_base.F90:140,256
 *      diff2diff        (1:D*D, zs:ze-1, xs:xe, ys:ye) c(src,dst), src fastest   src/pprts.F90:3471
 *      a11/a12/kabs/... (zs:ze-1, xs:xe, ys:ye);  albedo (xs:xe, ys:ye);  l1d (zs:ze-1)
 *    Only the rank-local (owned, un-ghosted) part is passed; ghosts/halos are the library's job.
 *  - `where`: TSX_HOST pointers are copied with hipMemcpy; TSX_DEVICE pointers are used in place.
 *  - Return value: 0 = ok, >0 = usage/runtime error (tsx_last_error() has the text).  Solver outcome
 *    is reported PETSc-style in tsx_ksp_result.reason (2 rtol, 3 atol, -3 its, -4 dtol, -5 breakdown,
 *    -9 nan) exactly as MyKSPConverged does (src/pprts.F90:4437-4486); the Fortran shim turns a
 *    negative reason into CHKERR like `solve` does (src/pprts.F90:4298-4302).
 *  - One handle per solver per rank; calls on one handle are not concurrent.  With nranks > 1 every
 *    rank of the solver's communicator must call set_coeffs/apply/solve collectively.
 *  - No CPU fallback exists: without a HIP device every compute entry point fails with
 *    TSX_ERR_NO_DEVICE.
 */
#ifndef TSX_H
#define TSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSX_VERSION 100

enum { TSX_HOST = 0, TSX_DEVICE = 1 };

enum {
  TSX_OK = 0,
  TSX_ERR_ARG = 1,
  TSX_ERR_NO_DEVICE = 2,
  TSX_ERR_HIP = 3,
  TSX_ERR_STATE = 4,
  TSX_ERR_UNSUPPORTED = 5,
  TSX_ERR_COMM = 6
};

/* solver ids: c_wrapper/f2c_solver_ids.h (SOLVER_ID_PPRTS_3_10 = 310, _8_16 = 816) */
enum { TSX_SOLVER_3_10 = 310, TSX_SOLVER_8_16 = 816 };

/* preconditioners for the flexible BiCGStab (reference default: PCILU / PCBJACOBI+ILU(0),
 * src/pprts.F90:4350-4371, 4415-4425; here GPU-native equivalents) */
/* NONE: bare operator.  COLUMN: exact column-block solves, block-Jacobi over the columns.  ZEBRA: the same blocks in
 * Gauss-Seidel order over even / odd rows (+ lagged x coupling).  REDBLACK (default): checkerboard order, every pass a
 * Gauss-Seidel step in x and y; 3_10 with fp32 directions and an even number of columns per row, else ZEBRA is used. */
enum { TSX_PC_NONE = 0, TSX_PC_COLUMN = 1, TSX_PC_ZEBRA = 2, TSX_PC_REDBLACK = 3 };

typedef struct tsx_solver tsx_solver; /* opaque */

/* Mirrors t_coord for C_diff (src/pprts_base.F90:92-109) plus the solver's stream layout
 * (src/pprts.F90:332-349, 413-425 are implied by solver_id). */
typedef struct {
  int32_t solver_id;          /* TSX_SOLVER_3_10 | TSX_SOLVER_8_16 */
  int32_t Nz;                 /* layers of the diffuse grid: C_diff%zm - 1 */
  int32_t xm, ym;             /* owned columns: C%xm, C%ym */
  int32_t xs, ys;             /* global start of the owned block: C%xs, C%ys (0-based) */
  int32_t glob_xm, glob_ym;   /* C%glob_xm, C%glob_ym */
  int32_t rank, nranks;       /* solver%myid, size of solver%comm */
  int32_t neigh_w, neigh_e;   /* C%neighbors(10), (16)  (src/pprts_base.F90:812-825) */
  int32_t neigh_s, neigh_n;   /* C%neighbors(4),  (22) */
  int32_t device;             /* HIP device ordinal, <0 = current device */
  int32_t force_halo;         /* testing: route even self-neighbour faces through halo buffers */
} tsx_grid;

/* tolerances as determine_ksp_tolerances + -<prefix>ksp_* overrides deliver them
 * (src/pprts_base.F90:1097-1142, src/pprts.F90:4245-4260) */
typedef struct {
  double rtol, atol, dtol;
  int32_t maxit;
  int32_t pc;                 /* TSX_PC_* */
  int32_t pc_sweeps;          /* ZEBRA / REDBLACK: pc_sweeps + 1 half-grid passes per application; COLUMN: Jacobi sweeps (1..32);
                                 0 (default) = automatic: 27 (28 passes) where the scan kernels run, else 9 */
  int32_t check_every;        /* host looks at the device convergence flag every n iterations; 0 (default) = automatic: every 2, and
                                 the first look where the previous solve of this handle from the same kind of guess ended, less one */
  int32_t fp32_directions;    /* 2 (default; with a preconditioner and rtol >= 1e-7, else like 1): as 1, and the recurrence vectors r, s, v, t are fp32
                                 too.  The iterate x, b, every dot product and the stop rule stay fp64 on the exact blocks; the
                                 recurrence residual is REPLACED by b - A x evaluated in fp64 whenever it has fallen four orders of
                                 magnitude since the last replacement and before convergence is declared, so the reference's
                                 criterion (MyKSPConverged) is decided on the true fp64 residual.  (The reference's ireals is
                                 real32 or real64 by build option.)
                                 1: the directions p, p-hat, s-hat and the shadow residual are stored in fp32 --
                                 flexible BiCGStab accepts any direction: x and r are updated consistently with A p-hat,
                                 A s-hat whatever they are; x, r, s, v, t, the dots and the stop rule stay fp64.  0: all fp64 */
  int32_t pc_coeff_fp16;      /* 1 (default): with fp32 directions the preconditioner reads a packed reduced-precision copy
                                 of the transport blocks (fp16; for 3_10 the couplings to neighbouring columns fp8 e4m3) and
                                 keeps its sweep temporaries in fp32 -- the preconditioner is an approximation anyway; the
                                 operator itself always uses the exact blocks.  0: exact blocks in the preconditioner too */
  int32_t skip_complete_initial_run; /* tsx_pprts_solve only.  0 (default): the first solve of a solution uid runs with tolerances
                                 at least as tight as tsx_determine_ksp_tolerances gives (-ksp_complete_initial_run,
                                 src/pprts.F90:4245-4256); the caller's looser values apply to the warm-started ones */
  int32_t explicit_solver;    /* 0 (default): flexible BiCGStab.  1: the explicit (stationary) solver, -<prefix>explicit of
                                 src/pprts.F90:2799 / explicit_ediff (src/pprts_explicit.F90:461-713): x += M^-1 (b - A x) with the
                                 red-black sweeps (pc_sweeps + 1 half-grid passes per outer iteration = -pc_sub_it) until the
                                 2-norm of the change (mean over ranks) is < atol or < rtol times the first iteration's; the
                                 result's res_hist holds those norms, rnorm0 = residual(1); maxit as given */
  int32_t accept_incomplete_solve; /* 0 (default): a solve that ends with reason <= 0 is repeated once from a zero guess with the
                                 conservative solver (src/pprts.F90:4277-4302).  1: -accept_incomplete_solve (:4271-4273) -- the
                                 reference returns BEFORE that retry; the (warm-started) partial iterate and the negative
                                 reason are the result, e.g. with -<prefix>ksp_max_it N as a fixed work budget per call */
  int32_t initial_guess_zero; /* tsx_diff_solve only.  0 (default): x holds the initial guess (KSPSetInitialGuessNonzero(TRUE),
                                 src/pprts.F90:4343).  1: the caller states that the guess is zero -- x is not read at all (the
                                 import of a zero guess otherwise costs a pass over x to find out), the solve starts from
                                 r = b.  On several ranks every rank must pass the same value */
} tsx_ksp_opts;

/* what `solve` stores on the solution: Niter_diff, diff_ksp_residual_history(100)
 * (src/pprts.F90:4232-4236, 4266; src/pprts_base.F90:163-166) */
typedef struct {
  int32_t reason;
  int32_t niter;
  double rnorm0, rnorm;
  double res_hist[100];
  int32_t nhist;
  float solve_ms;             /* device time of the Krylov loop (HIP events on the solver stream) */
  float import_ms, export_ms; /* layout conversion of b,x0 / x */
} tsx_ksp_result;

const char *tsx_last_error(void);
int tsx_version(void);
int tsx_device_count(void);
/* sizeof(tsx_grid), sizeof(tsx_ksp_opts), sizeof(tsx_ksp_result) as this library was built: what a binding in another language
 * (the Fortran shim's bind(C) types, the ctypes mirror) checks its own declarations against */
int tsx_abi_sizes(int32_t *sizes3);

/* ---- lifetime: init_pprts/setup_grid allocate C_diff and the vectors (src/pprts.F90:213, 830-1097);
 *      destroy_pprts frees them (src/pprts_base.F90:877) */
int tsx_create(const tsx_grid *grid, tsx_solver **out);
int tsx_destroy(tsx_solver *s);
void tsx_default_ksp_opts(tsx_ksp_opts *o);
/* determine_ksp_tolerances (src/pprts_base.F90:1097-1142): rtol 1e-5, atol max(1e-8, 1e-4*Nx*Ny*(Nz+1)*f);
 * f = unconstrained_fraction (share of layers that are not 1-D, src/pprts.F90:716-724); f < 0: the solver's own count */
int tsx_determine_ksp_tolerances(const tsx_solver *s, double unconstrained_fraction, double *rtol,
                                 double *atol, int32_t *maxit);
/* run on an existing HIP stream (hipStream_t as void*), e.g. the caller's current stream; NULL = own stream */
int tsx_set_stream(tsx_solver *s, void *hip_stream);

/* ---- communicator (nranks > 1): RCCL over xGMI replaces the MPI point-to-point of
 *      halo_fill_5pt / exchange_diffuse_boundary and imp_allreduce (src/pprts_base.F90:1622-1731,
 *      src/pprts_explicit.F90:715-848, 626).  Rank 0 creates the 128-byte id, the host broadcasts it
 *      (MPI_Bcast in TenStream, torch.distributed in bench.py), every rank calls tsx_comm_init. */
int tsx_comm_unique_id(void *id128);
int tsx_comm_init(tsx_solver *s, const void *id128);

/* ---- device-resident peer transport (node-local; replaces MPI_Isend/Irecv of exchange_diffuse_boundary,
 *      src/pprts_explicit.F90:715-848, and the MPI_Allreduce of the Krylov dots): every rank owns a mailbox in fine-grained
 *      device memory that its peers map through HIP IPC (xGMI between the GPUs of a node; rank processes sharing one device
 *      work too).  Halos are written by the sender's kernel straight into the receiver's mailbox and signalled with a
 *      sequence word; the 3-double all-reduces are all-to-all stores summed in rank order.  Only kernels on the solver's
 *      streams: no library call and no host round trip per exchange.
 *      Bootstrap like tsx_comm_unique_id / tsx_comm_init: every rank exports a blob of TSX_PEER_BLOB_BYTES, the host
 *      all-gathers them in rank order (MPI_Allgather in the reference's world) and hands the nranks blobs to attach.
 *      Takes precedence over RCCL and over the callbacks once attached.  At most 16 ranks. */
#define TSX_PEER_BLOB_BYTES 192
int tsx_comm_peer_export(tsx_solver *s, void *blob);
int tsx_comm_peer_attach(tsx_solver *s, const void *blobs);
/* collective self test (patterned face exchanges of varying length, verified on the receiver, and all-reduces); *failed = 0 if this
 * rank saw nothing wrong.  The caller reduces *failed over the ranks by other means and, if any rank failed, calls
 * tsx_comm_peer_disable on every rank: the solver then uses RCCL / the callbacks. */
int tsx_comm_peer_selftest(tsx_solver *s, int rounds, double *failed);
int tsx_comm_peer_disable(tsx_solver *s);
/* The mailboxes are uncached memory, so by default the kernels order payload and sequence word with s_waitcnt and relaxed
 * system-scope accesses only; heavy = 1 (or TSX_PEER_FENCES=1) puts full system-scope release / acquire fences around every
 * flag (each writes back / invalidates the L2: several microseconds per exchange) -- for a platform where the self test fails
 * without them.  tsx_comm_peer_reset returns the transport to its state after attach (counters, flags, recorded error) for a
 * second self test; the caller puts a barrier over all ranks before and after it. */
int tsx_comm_peer_set_fences(tsx_solver *s, int heavy);
int tsx_comm_peer_reset(tsx_solver *s);

/* Alternative transport: host-staged callbacks, for hosts whose communicator is MPI (TenStream's own
 * solver%comm) without GPU-aware transport, and for multi-process tests on one GPU.  The library copies the
 * four face buffers to pinned host memory, calls `exchange`, and copies the received faces back.
 *   send/recv/count/peer are indexed W,E,S,N.  recv[W] must be filled with peer W's send[E], recv[E] with peer
 *   E's send[W], recv[S] with peer S's send[N], recv[N] with peer N's send[S]  (exchange_diffuse_boundary,
 *   src/pprts_explicit.F90:769-843).  `allreduce` sums `n` doubles in place over all ranks (imp_allreduce).
 * Callbacks return 0 on success. */
typedef int (*tsx_exchange_fn)(void *ctx, const double *const send[4], double *const recv[4], const size_t count[4],
                               const int peer[4]);
typedef int (*tsx_allreduce_fn)(void *ctx, double *inout, int n);
int tsx_comm_set_callbacks(tsx_solver *s, tsx_exchange_fn exchange, tsx_allreduce_fn allreduce, void *ctx);

/* ---- operator values.  Replaces set_diff_coeff (src/pprts.F90:5511-5796): instead of one
 *      MatSetValuesStencil per cell the blocks are transposed once into stream-major planes.
 *      coeff_kind: 8 = real64 (ireals), 4 = real32.  Blocks whose real64 values are exactly
 *      representable in real32 (always true for LUT output, src/pprts.F90:3457) are stored as fp32. */
int tsx_diff_set_coeffs(tsx_solver *s, const void *diff2diff, int coeff_kind, const uint8_t *l1d,
                        const double *a11, const double *a12, const double *albedo, int where);

/* ---- coefficient source on the device (replaces the per-cell get_coeff loop of alloc_coeff_diff2diff,
 *      src/pprts.F90:3433-3462).  The table is the reference's LUT payload: real32 (nvec, nentries) column-major
 *      with tau fastest, then w0, aspect_zx, g (src/optprop_base.F90:438-442); axes as in
 *      src/optprop_parameters.F90 (tau31, w020, aspect23, g6 for LUT_3_10 / LUT_8_16).
 *      tsx_lut_load_diffuse_mmap4 reads a `.mmap4` file as written by src/mmap.F90:63-127 (one page of size_t
 *      header [dtype_size, n_elems, n_bytes, dim1, dim2, 0...], then the raw array) and applies the solver's
 *      preset axes. */
int tsx_lut_set_diffuse(tsx_solver *s, const float *table, int32_t nvec, int64_t nentries, int32_t ndim,
                        const int32_t *n, const float *axes_concat, int where);
int tsx_lut_load_diffuse_mmap4(tsx_solver *s, const char *path);
/* optical properties (delta-scaled, as atm%kabs/ksca/g/dz hold them after set_optical_properties,
 * src/pprts.F90:1903-1917) -> coefficient planes by N-linear interpolation with lattice snapping
 * (src/interpolation.F90:317-360).  Arrays (zs:ze-1, xs:xe, ys:ye) real64; dx scalar. */
int tsx_diff_set_optprop(tsx_solver *s, const double *kabs, const double *ksca, const double *g, const double *dz,
                         double dx, const uint8_t *l1d, const double *a11, const double *a12, const double *albedo,
                         int where);
/* read the coefficient blocks back in the reference layout (1:D*D, zs:ze-1, xs:xe, ys:ye), real64 -- what
 * solver%diff2diff holds (needed by calc_flx_div on the host, and by parity tests) */
int tsx_diff_get_coeffs(tsx_solver *s, double *diff2diff, int where);

/* ---- y = (I - T) x : op_mat_mult_ediff (src/pprts_shell.F90:366-541), assembled semantics for the
 *      surface row (albedo/streams on every up/down pair, src/pprts.F90:5755-5794) */
int tsx_diff_apply(tsx_solver *s, const double *x, double *y, int where);

/* ---- solve (I - T) x = b : replaces ediff()/KSPSolve (src/pprts.F90:2931-3031, 4206-4303).
 *      x is in/out (nonzero initial guess, src/pprts.F90:4343). */
int tsx_diff_solve(tsx_solver *s, const double *b, double *x, int where, const tsx_ksp_opts *opts,
                   tsx_ksp_result *res);

/* ---- the same two entries with the vectors in the caller's real kind: TenStream's ireals is real32 or real64 by build
 *      (src/data_parameters.F90; the CI builds both).  vec_kind 8 = real64 (identical to the entries above), 4 = real32:
 *      x, y, b are then float arrays, widened / narrowed on the device -- a real32 build hands its vectors over as they are. */
int tsx_diff_apply_r(tsx_solver *s, const void *x, void *y, int vec_kind, int where);
int tsx_diff_solve_r(tsx_solver *s, const void *b, void *x_inout, int vec_kind, int where, const tsx_ksp_opts *opts,
                     tsx_ksp_result *res);

/* ======================= the direct seam and setup_b on their own =========================================
 * `pprts()` offers the same seam for the direct beam as for the diffuse system (src/pprts.F90:2698-2755):
 *     if (lexplicit_dir) call explicit_edir(solver, prefix, edirTOA, solution%edir, lb, v0, solution, ierr)   else call edir(prefix)
 * with set_dir_coeff (:4493-4630) filling the matrix from solver%dir2dir.  The entries below take what that branch has
 * in hand -- coefficient blocks in the reference's layout, no optical properties, no LUT:
 *   dir2dir  (1:S*S, zs:ze-1, xs:xe, ys:ye)  c(src, dst), src fastest: flat index dst*S + src   (:4523-4590; S = 3 | 8)
 *   dir2diff (1:S*D, zs:ze-1, xs:xe, ys:ye)  c(src, dst), src fastest: flat index dst*S + src   (:4725-4821); NULL if only
 *            the beam is wanted
 *   coeff_kind 8 | 4 like tsx_diff_set_coeffs; the values must be exactly representable in real32 (they are: the tables
 *   are irealLUT = real32 and the reference only widens them, :3121-3143) -- else TSX_ERR_UNSUPPORTED
 *   l1d (zs:ze-1); a33 (beam transmission of the 1-D layers, :4513-4520), a13 / a23 (their sources, :4709-4721):
 *   (zs:ze-1, xs:xe, ys:ye) real64, only read where a layer is 1-D (NULL if none is)
 *   dx, dy: setup_incSolar's edirTOA * dx * dy / area_divider (src/pprts_base.F90:1146-1181)
 * tsx_pprts_set_angles comes first (sweep order xinc / yinc, src/pprts.F90:1157-1167) and invalidates the coefficients
 * when called again, exactly as set_angles does in the reference (:1100-1116). */
int tsx_dir_set_coeffs(tsx_solver *s, const void *dir2dir, const void *dir2diff, int coeff_kind, const uint8_t *l1d,
                       const double *a33, const double *a13, const double *a23, double dx, double dy, int where);
/* explicit_edir (src/pprts_explicit.F90:60-459): edir (0:S-1, zs:ze, xs:xe, ys:ye) [W] of kind vec_kind is the initial
 * iterate on entry (v0 = solution%edir, src/pprts.F90:2746) and the beam on exit.  Forward sweeps in sun order, repeated
 * until the 2-norm of the change (mean over ranks) is < atol or < rtol times the first one's (:168-218); rtol / atol /
 * maxit as explicit_edir has derived them from -solar_dir_ksp_* (:94-121), <= 0: determine_ksp_tolerances / 1000.
 * *converged = 0 when maxit sweeps did not suffice (reported, not an error: -accept_incomplete_solve is the caller's). */
int tsx_dir_solve(tsx_solver *s, double edirTOA, void *edir_inout, int vec_kind, int where, double rtol, double atol,
                  int32_t maxit, int32_t *niter, double *residual, int32_t *converged);
/* setup_b, solar (set_solar_source, src/pprts.F90:4684-4846): b (0:D-1, zs:ze, xs:xe, ys:ye) [W] from the beam: edir in the
 * layout above, or NULL = the beam tsx_dir_solve left on the device (required on several ranks); albedo (xs:xe, ys:ye) or
 * NULL = the one tsx_diff_set_coeffs received.  Needs tsx_dir_set_coeffs with dir2diff. */
int tsx_setup_b_solar(tsx_solver *s, const void *edir, const double *albedo, void *b, int vec_kind, int where);
/* setup_b, thermal (set_thermal_source, src/pprts.F90:4848-4987): planck (zs:ze, xs:xe, ys:ye), planck_srfc (xs:xe, ys:ye)
 * or NULL (atm%Bsrfc, :4958-4970), kabs and dz (zs:ze-1, xs:xe, ys:ye), real64.  Emissivities come from the diffuse blocks,
 * a11 / a12 and the albedo of tsx_diff_set_coeffs / tsx_diff_set_optprop, which comes first. */
int tsx_setup_b_thermal(tsx_solver *s, const double *planck, const double *planck_srfc, const double *kabs, const double *dz,
                        double dx, double dy, void *b, int vec_kind, int where);

/* ======================= whole g-point on the device ===============================
 * The pieces of `pprts()` around the diffuse solve (src/pprts.F90:2668-2820) and of restore_solution /
 * pprts_get_result, so that only optical properties go in and four small flux arrays come out:
 *   alloc_coeff_dir2dir/dir2diff (:3088-3391) -> tsx_lut_set_direct + lookup inside tsx_pprts_solve
 *   explicit_edir (src/pprts_explicit.F90:60-459)   -> column-marching sweeps iterated to the same stop rule
 *   setup_b (:4641-4987)                            -> written directly in dst-owned storage
 *   calc_flx_div (:5152-5504), scale_flx (:3682-3988), pprts_get_result (:5799-5888) -> tsx_pprts_get_result
 * Solvers 3_10 (3 direct streams) and 8_16 (8 direct streams: 4 per top face, 2 per side face, src/pprts.F90:413-425;
 * tables relabelled for the sun's quadrant by dir2dir8_coeff_symmetry / dir8_to_diff16_coeff_symmetry,
 * src/optprop.F90:1186-1302).  On several ranks the direct sweep exchanges one face per sweep with the upwind / downwind neighbours
 * (exchange_direct_boundary, src/pprts_explicit.F90:1076-1140), its residual is the mean over ranks of the local norms
 * (:184), setup_b needs no exchange in dst-owned storage, and the flux divergence reads one halo update of the solution. */
/* set_angles (src/pprts.F90:1100-1183): sun azimuth phi0 / zenith theta0 in degrees as pprts_f2c_init takes them */
int tsx_pprts_set_angles(tsx_solver *s, double phi0, double theta0);
/* stop rule of the direct sweep (explicit_edir reads -solar_dir_ksp_rtol / _atol / _max_it, src/pprts_explicit.F90:94-121);
 * a value <= 0 keeps the default of determine_ksp_tolerances / default_max_it; a uid's first solve never runs looser than
 * the defaults (-ksp_complete_initial_run) */
int tsx_pprts_set_direct_tolerances(tsx_solver *s, double rtol, double atol, int32_t maxit);
/* direct tables Tdir (S*S per entry) and Sdir (S*D per entry; S = 3 / 8 for 3_10 / 8_16), 6 axes [tau, w0, aspect_zx, g, phi, theta]
 * (src/optprop_base.F90:228-240); same payload layout as the diffuse table */
int tsx_lut_set_direct(tsx_solver *s, const float *Tdir, const float *Sdir, int64_t nentries, int32_t ndim,
                       const int32_t *n, const float *axes_concat, int where);
/* the same from `.mmap4` files (LUT_direct_3_10.<dims>.ds1000.nc.{Tdir,Sdir}.mmap4, src/optprop_LUT.F90:505, 1348);
 * axes = the LUT_3_10 preset unless a `<tdir_path>.axes` text sidecar (ndim, then "n v1..vn" per axis) exists */
int tsx_lut_load_direct_mmap4(tsx_solver *s, const char *tdir_path, const char *sdir_path);
/* set_optical_properties (src/pprts.F90:1764): raw kabs/ksca/g/dz (zs:ze-1, xs:xe, ys:ye) real64, albedo (xs:xe, ys:ye),
 * planck (zs:ze, xs:xe, ys:ye) or NULL; planck_srfc (xs:xe, ys:ye) or NULL -- the optional argument of the same name
 * (:1773, 1823-1829): the surface's own Planck emission atm%Bsrfc, which the thermal source then uses at the ground with
 * the emissivity 1 - albedo clamped to [0, 1] (:4958-4970) instead of planck at the lowest level (:4971-4984); the rrtmg
 * driver always passes it (rrtmg/rrtmg/pprts_rrtmg.F90:609-642, 681).  Like atm%Bsrfc it is dropped by a call without it.
 * On the device: delta scaling with f = g**2 when ldelta_scaling (:1903-1917,
 * src/helper_functions.fypp:1622-1666), 1-D layer detection dz/dx > twostr_ratio = 2 (:669-677), eddington_coeff_ec of the
 * 1-D layers (:1962-1992, src/eddington.F90:173-241), diffuse coefficient lookup.  Needs tsx_pprts_set_angles and the
 * diffuse LUT first. */
int tsx_pprts_set_optical_properties(tsx_solver *s, const double *albedo, const double *kabs, const double *ksca,
                                     const double *g, const double *dz, const double *planck, const double *planck_srfc,
                                     double dx, double dy, int ldelta_scaling, int where);
/* optical properties of one g-point, (zs:ze-1, xs:xe, ys:ye) real64, already delta-scaled; a11..a33 only read for
 * 1-D layers (eddington coefficients, src/pprts.F90:1962-1992); planck (zs:ze, xs:xe, ys:ye) or NULL for solar;
 * planck_srfc (xs:xe, ys:ye) or NULL as above */
int tsx_pprts_set_optprop(tsx_solver *s, const double *kabs, const double *ksca, const double *g, const double *dz,
                          double dx, double dy, const double *albedo, const uint8_t *l1d, const double *a11,
                          const double *a12, const double *a13, const double *a23, const double *a33,
                          const double *planck, const double *planck_srfc, int where);
/* solve_pprts/pprts() for one g-point: lsolar = edirTOA > 0 semantics are the caller's (pprts_f2c_solve,
 * c_wrapper/f2c_pprts.F90:340-341).  The previous solution of this handle is the initial guess
 * (src/pprts.F90:2542-2558) unless tsx_pprts_zero_guess was called. */
int tsx_pprts_solve(tsx_solver *s, double edirTOA, int lsolar, const tsx_ksp_opts *opts, tsx_ksp_result *res);
int tsx_pprts_zero_guess(tsx_solver *s);
/* solve_pprts(..., opt_solution_uid) (src/pprts.F90:2487-2558; get_solution_uid): make `uid` the solution the next
 * tsx_pprts_solve works on.  The current solution is parked under its own uid (as real32; the reference's
 * lcompress_solutions does the like); the initial guess becomes uid's previous solution, or, for a uid never solved, the
 * solution of uid - 1 (-initial_guess_from_last_uid, on by default in the reference), or zero.  Read the result of a
 * solve before selecting another uid (as rrtmg/rrtmg/pprts_rrtmg.F90:999-1055 does). */
int tsx_pprts_select_solution(tsx_solver *s, int32_t uid);
/* restore_solution + pprts_get_result: edn, eup, edir (zs:ze, xs:xe, ys:ye), abso (zs:ze-1, xs:xe, ys:ye), W/m2 and
 * W/m3, solar results multiplied by sun%mu (src/pprts.F90:5883-5888).  edir may be NULL. */
int tsx_pprts_get_result(tsx_solver *s, double *edn, double *eup, double *abso, double *edir, int where);
/* parity probes: which = 0 edir [W] (0:S-1, zs:ze, xs:xe, ys:ye); 1 b [W]; 2 ediff [W] (0:D-1, zs:ze, ...);
 * 3 dir2dir (S*S, zs:ze-1, ...); 4 dir2diff (S*D, ...) -- reference layouts, real64;
 * what tsx_pprts_set_optical_properties derived on the device, (zs:ze-1, xs:xe, ys:ye): 5 kabs, 6 ksca, 7 g after delta scaling
 * (src/pprts.F90:1903-1917); 8..12 the Eddington coefficients a11, a12, a13, a23, a33 of the 1-D layers (:1962-1992; NaN in
 * layers that are not 1-D) */
int tsx_pprts_get_field(tsx_solver *s, int which, double *out, int where);

/* ---- coefficient probe: pprts_f2c_opp_get_coeff / _get_info (c_wrapper/f2c_pprts.h:54-83, f2c_pprts.F90:627-760).
 *      One raw table lookup on the device, get_coeff_cube semantics (src/optprop.F90:549-582): imode 1 dir2dir,
 *      2 dir2diff (both with the quadrant relabelling for lswitch_east / lswitch_north), 3 diff2diff; only aspect_zx
 *      is clamped (from below).  Ncoeff must equal S*S, S*D or D*D.  ranges20: [min, max] of diffuse tau, w0, g,
 *      aspect_zx, then direct tau, w0, g, aspect_zx, phi, theta. */
int tsx_opp_get_coeff(tsx_solver *s, float tauz, float w0, float g, float aspect_zx, float phi, float theta, int imode,
                      int lswitch_east, int lswitch_north, int ncoeff, float *coeff);
int tsx_opp_get_info(tsx_solver *s, int32_t *Ndir, int32_t *Ndiff, float *ranges20);

/* ---- z = M^-1 v with the preconditioner the solve uses (exposed for parity tests: M is the column-block
 *      diagonal of the assembled matrix in the dst-owned numbering) */
int tsx_diff_pc_apply(tsx_solver *s, const double *v, double *z, int where, int pc, int pc_sweeps, int mixed);

/* ---- measurement helpers (bench.py): time `reps` launches of the dominant kernel with HIP events on
 *      the solver's stream.  kernel: 0 = SpMV (diffuse operator apply), 1 = one full BiCGStab iteration,
 *      2 = one application of the default preconditioner (pc_sweeps + 1 half-grid passes), 3 = one intermediate
 *      Gauss-Seidel pass of it (scan kernels) as a launch of its own, 4 = the flow kernel (the intermediate passes of an
 *      application in ONE launch; TSX_ERR_UNSUPPORTED where the configuration runs a launch per pass), 5 = kernel 1's iteration
 *      replayed from a hipGraph captured from the solver's stream (one rank; a measurement of what graph replay buys) */
int tsx_bench_kernel(tsx_solver *s, int kernel, int reps, float *avg_ms);
/* algorithmic bytes per launch of that kernel *in the storage format in use* (with shared storage of identical blocks:
 * every distinct block once + a 4-byte index per cell + the vectors; the preconditioner passes: packed records + fp32
 * right-hand side + neighbour records + stores, see the table in tsx_api.hip).  kernel 10 / 11: SURVEY 8(d)'s literal
 * figures for every cell's block stored, Nc*D^2*sc + 2*N*sv for the SpMV and 2*B_spmv + 16*N*sv for an iteration
 * (kernel 1 reports the same literal figure: the reference's estimate of an iteration without a preconditioner) */
int tsx_algorithmic_bytes(const tsx_solver *s, int kernel, double *bytes);
/* shared storage of bit-identical transport blocks (lossless; tsx_dedup.hip): how many distinct blocks the current
 * coefficients hold and whether the operator apply / preconditioner read them through the per-cell index (they do when
 * at most half of the cells need a block of their own; TSX_DEDUP=0 switches it off).  *on bit 0: bit-identical blocks shared
 * (*nent of them); bit 1: blocks that agree to about 1 % are grouped FOR THE PRECONDITIONER ONLY (its per-block records are
 * approximate by design; the operator keeps exact blocks) -- alone where nothing is bit-identical (*nent = the groups), or on top
 * of bit 0 where it halves the preconditioner's table.  TSX_DEDUP_NEAR=0 switches that off.  Bit 2: the grouping (which cells share a
 * block) was taken over from the previous coefficient set of this solver after one validation kernel found it still exact for the new
 * LUT coordinates -- the case of a spectral loop over g-points of one scene; TSX_DEDUP_REUSE=0 rebuilds it every time.  Bit 3: likewise
 * the grouping of the cells by their packed column-recurrence records (tsx_records_share), as of the last packing */
int tsx_dedup_info(tsx_solver *s, int32_t *on, int64_t *nent);
/* the preconditioner the last solve / tsx_bench_kernel actually ran (after the automatic choices: red-black -> zebra rows
 * on odd grids, pc_sweeps 0 -> 27 or 9): TSX_PC_*, pc_sweeps, and scan: 0 = one-lane-per-column kernels, 1 = scan kernels,
 * 3 = scan kernels reading identical recurrence records through a shared table */
int tsx_pc_info(const tsx_solver *s, int32_t *pc, int32_t *pc_sweeps, int32_t *scan);
/* how the last application of M^-1 ran its intermediate passes (tsx_k_pcs_flow, DESIGN.md section 4 "The passes of an application
 * as one launch"): info8 = {flow kernel used (0: a launch per pass), first pass, one past the last pass of the launch, columns per
 * tile, fat body (every neighbour-independent load hoisted, two waves per SIMD), records as granules, tiles per pass, workgroups} */
int tsx_flow_info(const tsx_solver *s, int32_t *info8);
/* device STREAM-like copy bandwidth probe (GB/s) for reporting against the measured peak */
int tsx_probe_copy_bandwidth(tsx_solver *s, size_t bytes, int reps, double *gbps);
/* the same probe in full: out4 = [best copy GB/s (read + written bytes), best read-only GB/s, copy variant, read variant] over
 * streaming kernels with 1 / 4 / 8 sixteen-byte accesses per lane in flight, plain and non-temporal, on capped grids -- the
 * ceiling `roofline.frac_of_achievable` in bench.py is quoted against beside MI355X_MICROARCH.md's 6.3 TB/s */
int tsx_probe_bandwidth(tsx_solver *s, size_t bytes, int reps, double *out4);

/* Log events of this path, named like the reference's (solver%logs, src/pprts_base.F90:176-209; begun / ended at src/pprts.F90:1785-2077
 * set_optprop, :3422-3489 get_coeff_diff2diff, :3116-3392 get_coeff_dir2dir / dir2diff, :2694-2756 compute_Edir, :2903-2912 solve_Mdir,
 * :2760-2818 compute_Ediff, :2952-2954 setup_Mdiff, :3012-3021 solve_Mdiff, :5197-5479 compute_absorption; setup_diff_src, get_result):
 * per event a count and the DEVICE time between two HIP events recorded on the solver's stream around it, and a roctx range of the same
 * name around the host code (rocprofv3 --marker-trace shows a spectral loop g-point by g-point; roctx is bound at run time, absent = no
 * ranges).  Off by default (no event records on the hot path); tsx_log_enable(s, 1) or TSX_LOG=1 at tsx_create switch it on.
 * tsx_log_get synchronises the stream and fills up to *nevents = 11 entries; names[] point to static strings; any array may be null. */
int tsx_log_enable(tsx_solver *s, int on);
int tsx_log_get(tsx_solver *s, int32_t *nevents, const char **names, int64_t *counts, double *ms);

/* libtsx's device memory pool (tsx_pool.hip): driver allocations are taken once per process, held in quarantine until their contents
 * have stayed intact for TSX_POOL_GUARD_US (default 3000) microseconds, and sub-allocated from then on (TSX_POOL=0: straight to
 * hipMalloc, for A/B runs).  out8 = {slabs taken from the driver, their bytes, bytes handed out now, pieces, verifies that found a
 * fresh slab's pattern damaged, words damaged, microseconds after hipMalloc of the first damaged verify (-1: none), microseconds
 * spent in quarantine}.  device < 0: the current device.  The reference allocates its coefficient arrays once per solver
 * (alloc_coeff_diff2diff, src/pprts.F90:3396-3490); here no driver allocation lies on a solver's path after the first set. */
int tsx_pool_stats(int device, int64_t *out8);

/* diagnostics: the code of one of libtsx's eight device code objects AS IT SITS IN DEVICE MEMORY (unit 0..7 = api, spmv310, spmv816,
 * pc, pcs, pcsflow, dedup, peer).  The unit's probe kernel reports its program counter in *pc_out and copies nwords 32-bit words from
 * pc + delta to host_out (nwords = 0: the pc only); scripts/code_verify.py knows the units' ELF images inside libtsx.so and compares
 * every loaded .text byte with the file.  device < 0: the current device.  No reference counterpart. */
int tsx_debug_code_read(int device, int unit, long long delta, long long nwords, void *host_out, unsigned long long *pc_out);

#ifdef __cplusplus
}
#endif
#endif /* TSX_H */
