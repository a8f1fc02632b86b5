/*
 * tsx_f2c.h -- the reference's own C-ABI for pprts, served by the MI355X back-end.
 *
 * libtsx_f2c.so exports exactly the entry points of c_wrapper/f2c_pprts.h:48-52 with the argument meaning of
 * c_wrapper/f2c_pprts.F90:130-478, so a C caller written against TenStream (c_wrapper/pprts.c) links unchanged:
 *   - arrays are Fortran-ordered (Nz[+1], Nx, Ny), z fastest, real32 at the ABI, k = 1 at the top of the atmosphere;
 *   - solver_id / Nz / Nx / Ny / dx / dy / phi0 / theta0 / collapseindex travel by pointer into init (the Fortran
 *     side declares them intent(inout)); Nz/Nx/Ny by value elsewhere; edirTOA and lfinalizepetsc by value
 *     (f2c_pprts.F90:325-327, 458-459 -- the header's `int *lfinalizepetsc` notwithstanding, pprts.c passes 0);
 *   - one global solver instance, not re-entrant (module variable pprts_solver, f2c_pprts.F90:106);
 *   - no return codes: errors print and abort, like CHKERR (src/helper_functions.fypp:888-904).
 *   - every rank of the communicator calls every function; only rank 0's arguments are read (they are broadcast and
 *     written back, f2c_pprts.F90:189-230), rank 0's global arrays are scattered to the ranks' blocks
 *     (set_global_optical_properties, src/pprts.F90:2341-2451) and the results gathered to rank 0
 *     (pprts_get_result_toZero, :6265-6359).
 * Differences: the communicator behind `fcomm` is attached with tsx_f2c_set_comm (this library links no MPI: an MPI
 * host passes two small callbacks over its comm, INTEGRATION.md shows them); without it there is one rank.  solver_id
 * 310 (3_10) or 816 (8_16);
 * collapseindex must be <= 1; look-up tables are read from $LUT_BASENAME (src/tenstream_options.F90:103-105)
 * in `.mmap4` form (src/mmap.F90), file names as gen_lut_basename builds them (src/optprop_LUT.F90:364-374).
 * Options: like the reference's options database (src/options_database.F90:60-100) the library reads ./tenstream.options
 * and then $PETSC_OPTIONS ("-key [value]", later overrides earlier) and acts on the keys of this path:
 * -solar_diff_ksp_rtol / _atol / _max_it, -thermal_diff_ksp_*, -solar_dir_ksp_*, -solar_diff_explicit,
 * -thermal_diff_explicit, -accept_incomplete_solve; everything else is ignored.
 */
#ifndef TSX_F2C_H
#define TSX_F2C_H
#ifdef __cplusplus
extern "C" {
#endif

void pprts_f2c_init(int fcomm, int *solver_id, int *Nz, int *Nx, int *Ny, double *dx, double *dy, float *hhl,
                    float *phi0, float *theta0, int *collapseindex);
void pprts_f2c_set_global_optical_properties(int Nz, int Nx, int Ny, float *albedo, float *kabs, float *ksca, float *g,
                                             float *planck);
void pprts_f2c_solve(int fcomm, float edirTOA);
void pprts_f2c_get_result(int Nz, int Nx, int Ny, float *edn, float *eup, float *abso, float *edir);
void pprts_f2c_destroy(int lfinalizepetsc);

/* coefficient probe, c_wrapper/f2c_pprts.h:54-83 (f2c_pprts.F90:587-760): imode 1 dir2dir, 2 dir2diff, 3 diff2diff;
 * the lookup runs on the device (tsx_opp_get_coeff); solver_id 310 or 816 (the reference serves 310 only) */
void pprts_f2c_opp_init(const int comm, const int solver_id, void **opp, int *ierr);
void pprts_f2c_opp_get_coeff(void *opp, const float tauz, const float w0, const float g, const float aspect_zx, const float phi,
                             const float theta, const int imode, const int lswitch_east, const int lswitch_north,
                             const int Ncoeff, float *coeff, int *ierr);
void pprts_f2c_opp_destroy(void *opp, int *ierr);
void pprts_f2c_opp_get_info(void *opp, int *Ndir, int *Ndiff, float *diff_tauz_range, float *diff_w0_range, float *diff_g_range,
                            float *diff_aspect_zx_range, float *dir_tauz_range, float *dir_w0_range, float *dir_g_range,
                            float *dir_aspect_zx_range, float *dir_phi_range, float *dir_theta_range, int *ierr);

/* this back-end's way to say what `fcomm` is: rank / size and the two collectives of include/tsx.h (tsx_exchange_fn:
 * the face exchange with the W, E, S, N neighbours; tsx_allreduce_fn: sum of n doubles over all ranks).  Call on every
 * rank before pprts_f2c_init.  Rank layout: x fastest, MPI_Dims_create split (src/pprts_base.F90:747-790). */
#include "tsx.h"
void tsx_f2c_set_comm(int rank, int nranks, tsx_exchange_fn exchange, tsx_allreduce_fn allreduce, void *ctx);

#ifdef __cplusplus
}
#endif
#endif
