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
 * Differences: `fcomm` is ignored (one process, one GPU); solver_id 310 (3_10) or 816 (8_16);
 * collapseindex must be <= 1; look-up tables are read from $LUT_BASENAME (src/tenstream_options.F90:103-105)
 * in `.mmap4` form (src/mmap.F90), file names as gen_lut_basename builds them (src/optprop_LUT.F90:364-374).
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

#ifdef __cplusplus
}
#endif
#endif
