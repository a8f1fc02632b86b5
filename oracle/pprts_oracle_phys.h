/*
 * pprts_oracle_phys.h -- CPU restatement of the coefficient / source / post-processing pieces around
 * the pprts solve.  TEST INFRASTRUCTURE ONLY (see pprts_oracle.h).
 */
#ifndef PPRTS_ORACLE_PHYS_H
#define PPRTS_ORACLE_PHYS_H
#include <stddef.h>
#include <stdint.h>

#include "pprts_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* search_sorted_bisection: src/search.fypp:177-228.  Returns the 1-based fractional location. */
double orc_search_sorted_bisection_f64(const double *arr, int n, double val);
float orc_search_sorted_bisection_f32(const float *arr, int n, float val);

/* interp_vec_bilinear_iterative + dim_needs_interpolation (lattice snapping 1e-3):
 * src/interpolation.F90:317-360, 546-556, constant :63.  pti are 1-based fractional indices,
 * db is (nvec, nentries) column-major, db_offsets from ndarray_offsets
 * (src/helper_functions.fypp:2431-2437).  real32 arithmetic like the reference (irealLUT). */
void orc_interp_vec_nd_f32(const float *pti, int ndim, const float *db, int nvec, const int64_t *db_offsets,
                           float *Cres);
void orc_ndarray_offsets(const int *shape, int ndim, int64_t *offsets);

/* LUT description: axes of one table (src/optprop_base.F90:228-240 for LUT_3_10) */
#define ORC_LUT_MAXDIM 8
typedef struct {
  int ndim;
  int n[ORC_LUT_MAXDIM];
  const float *axis[ORC_LUT_MAXDIM]; /* axis values, ascending */
  int nvec;                          /* coefficients per entry: D*D, S*S or S*D */
  const float *table;                /* (nvec, prod(n)) column-major: src/mmap.F90:63-127 payload */
} orc_lut;

/* get_coeff for diffuse coefficients: clamp (src/pprts_base.F90:1517-1533), sample order
 * [tauz, w0, aspect, g] (src/optprop.F90:591), find_real_location per dim, N-linear interpolation
 * (src/optprop_LUT.F90:1560-1596).  out: nvec floats. */
void orc_get_coeff_diff2diff(const orc_lut *lut, double kabs, double ksca, double g, double dz, double dx,
                             float *out);
/* whole field: alloc_coeff_diff2diff (src/pprts.F90:3433-3462); only cells of non-1D layers are written */
void orc_alloc_coeff_diff2diff(const orc_lut *lut, int Nz, int xm, int ym, const double *kabs, const double *ksca,
                               const double *g, const double *dz, double dx, const uint8_t *l1d, double *coeffs);

/* delta_scale (f = g**2 unless given): src/helper_functions.fypp:1622-1666 */
void orc_delta_scale(double *kabs, double *ksca, double *g, int have_f, double f);

/* eddington_coeff_ec: src/eddington.F90:173-241.  out: t(a11), r(a12), rdir(a13), sdir(a23), tdir(a33) */
void orc_eddington_coeff_ec(double dtau, double w0, double g, double mu0, double *t, double *r, double *rdir,
                            double *sdir, double *tdir);

/* B_eff: src/schwarzschild.F90:36-67 (2-point Gauss-Legendre on (0,1), dgauss :173-303) */
double orc_B_eff(double B_far, double B_near, double tau);

#ifdef __cplusplus
}
#endif
#endif
