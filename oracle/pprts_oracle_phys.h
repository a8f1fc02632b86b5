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

/* ============================ direct beam, source term, post-processing ============================== */
#ifndef PPRTS_ORACLE_PIPE_H
#define PPRTS_ORACLE_PIPE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int dtop, dside;          /* dirtop%dof, dirside%dof           src/pprts.F90:344-348 (3_10: 1,1), :420-423 (8_16: 4,2) */
  int top_div, side_div;    /* dirtop/dirside%area_divider */
} orc_dir_layout;
void orc_dir_layout_3_10(orc_dir_layout *d);
void orc_dir_layout_8_16(orc_dir_layout *d);

/* setup_suninfo: src/pprts.F90:1118-1183 (angles in degrees) */
typedef struct {
  double phi, theta, mu, costheta, symmetry_phi;
  int xinc, yinc;
} orc_suninfo;
void orc_setup_suninfo(double phi, double theta, orc_suninfo *sun);

/* dir2dir / dir2diff stream relabelling for a sun in the east / north half, in place on one coefficient vector
 * (flat dst * S + src): src/optprop.F90:1009-1045 (3_10 dir2diff), :1186-1240 (8_16 dir2diff), :1268-1302 (8_16 dir2dir),
 * :1256-1266 (3_10 dir2dir: none).  Only the dst blocks the reference assigns are touched. */
void orc_dir_coeff_symmetry(int is_dir2dir, int S, int D, int lswitch_east, int lswitch_north, float *coeff);

/* get_coeff, direct branch (src/pprts_base.F90:1517-1542 -> src/optprop.F90:568-582 -> LUT_get_dir2dir /
 * LUT_get_dir2diff src/optprop_LUT.F90), sample order [tauz, w0, aspect, g, phi, theta]; for 3_10 dir2dir has no
 * symmetry swap, dir2diff uses dir3_to_diff10_coeff_symmetry (src/optprop.F90:1009-1045). */
void orc_get_coeff_dir(const orc_lut *lut, int is_dir2dir, int S, int D, double kabs, double ksca, double g, double dz,
                       double dx, double sym_phi, double theta, int lswitch_east, int lswitch_north, float *out);
void orc_alloc_coeff_dir(const orc_lut *lut, int is_dir2dir, int S, int D, int Nz, int xm, int ym, const double *kabs,
                         const double *ksca, const double *g, const double *dz, double dx, const orc_suninfo *sun,
                         const uint8_t *l1d, double *coeffs);

/* setup_incSolar + explicit_edir (+ forward sweep, exchange_direct_boundary with self neighbours):
 * src/pprts_base.F90:1146-1181, src/pprts_explicit.F90:60-459.  edir in/out (S, L, xm, ym) [W]. */
int orc_explicit_edir_1rank(const orc_layout *l, const orc_dir_layout *d, const orc_suninfo *sun, const double *dir2dir,
                            const uint8_t *l1d, const double *a33, double edirTOA, double dx, double dy, double rtol,
                            double atol, int maxit, double *edir, int *niter);

/* setup_b: src/pprts.F90:4641-4987 (solar: edir given in W; thermal: planck (L, xm, ym) at levels) */
void orc_setup_b_solar_1rank(const orc_layout *l, const orc_dir_layout *d, const orc_suninfo *sun, const double *dir2diff,
                             const uint8_t *l1d, const double *a13, const double *a23, const double *albedo,
                             const double *edir, double *b);
void orc_setup_b_thermal_1rank(const orc_layout *l, const double *diff2diff, const uint8_t *l1d, const double *a11,
                               const double *a12, const double *albedo, const double *planck,
                               const double *planck_srfc /* (xm, ym) = atm%Bsrfc, or NULL */, const double *kabs,
                               const double *dz, double dx, double dy, double *b);

/* gen_scale_*_flx_vec_arr: src/pprts.F90:3901-3987.  to_Wm2 = 1: W -> W/m2 */
void orc_scale_diff(const orc_layout *l, const double *dz, double dx, double dy, int to_Wm2, double *ediff);
void orc_scale_dir(const orc_layout *l, const orc_dir_layout *d, const double *dz, double dx, double dy, int to_Wm2,
                   double *edir);

/* calc_flx_div, default by_coeff_divergence branch: src/pprts.F90:5286-5398, 5477, 5483-5503.  fluxes in W;
 * abso out (Nz, xm, ym) in W/m3.  b_thermal may be NULL (solar) */
void orc_calc_flx_div_1rank(const orc_layout *l, const orc_dir_layout *d, const orc_suninfo *sun, const double *dir2dir,
                            const double *dir2diff, const double *diff2diff, const uint8_t *l1d, const double *a11,
                            const double *a12, const double *kabs, const double *dz, double dx, double dy,
                            const double *edir /* NULL if thermal */, const double *ediff, const double *b_thermal,
                            double *abso);

/* pprts_get_result: src/pprts.F90:5871-5888 (inputs in W/m2) */
void orc_get_result(const orc_layout *l, const orc_dir_layout *d, const orc_suninfo *sun, int lsolar, const double *edir,
                    const double *ediff, const double *abso, double *redir, double *redn, double *reup, double *rabso);

#ifdef __cplusplus
}
#endif
#endif
