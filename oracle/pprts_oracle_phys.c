/*
 * pprts_oracle_phys.c -- see pprts_oracle_phys.h.  TEST INFRASTRUCTURE ONLY (parity oracle).
 * Compiled with -ffp-contract=off so that real32 arithmetic follows the reference's operation order.
 */
#include "pprts_oracle_phys.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- search_sorted_bisection: src/search.fypp:177-228 (1-based indices kept) ----------------- */
#define DEFINE_BISECTION(NAME, T)                                                   \
  T NAME(const T *arr1, int n, T val) {                                             \
    const T *arr = arr1 - 1; /* arr[1..n] */                                        \
    int i = 1, j = n, k;                                                            \
    T loc_increment, res;                                                           \
    if (arr[i] <= arr[j]) { /* ascending */                                         \
      for (;;) {                                                                    \
        k = (i + j) / 2;                                                            \
        if (val < arr[k]) j = k;                                                    \
        else i = k;                                                                 \
        if (i + 1 >= j) {                                                           \
          if (i == j) loc_increment = 0;                                            \
          else loc_increment = (val - arr[i]) / (arr[j] - arr[i]);                  \
          res = (T)i + loc_increment;                                               \
          if (res < (T)1) res = (T)1;                                               \
          if (res > (T)n) res = (T)n;                                               \
          return res;                                                               \
        }                                                                           \
      }                                                                             \
    } else { /* descending */                                                       \
      for (;;) {                                                                    \
        k = (i + j) / 2;                                                            \
        if (val > arr[k]) j = k;                                                    \
        else i = k;                                                                 \
        if (i + 1 >= j) {                                                           \
          if (i == j) loc_increment = 0;                                            \
          else loc_increment = (val - arr[j]) / (arr[i] - arr[j]);                  \
          res = (T)j - loc_increment;                                               \
          if (res < (T)1) res = (T)1;                                               \
          if (res > (T)n) res = (T)n;                                               \
          return res;                                                               \
        }                                                                           \
      }                                                                             \
    }                                                                               \
  }
DEFINE_BISECTION(orc_search_sorted_bisection_f64, double)
DEFINE_BISECTION(orc_search_sorted_bisection_f32, float)

/* ndarray_offsets: src/helper_functions.fypp:2431-2437 */
void orc_ndarray_offsets(const int *shape, int ndim, int64_t *offsets) {
  offsets[0] = 1;
  for (int d = 1; d < ndim; ++d) offsets[d] = offsets[d - 1] * shape[d - 1];
}

/* dim_needs_interpolation: src/interpolation.F90:546-556, snapping constant :63 */
static int dim_needs_interpolation(float pti) {
  const float snap = 1e-3f;
  const float frac = pti - (float)(int)pti;
  if (frac < snap) return 0;
  if (frac > 1.0f - snap) return 0;
  return 1;
}

/* interp_vec_bilinear_iterative: src/interpolation.F90:317-360 */
void orc_interp_vec_nd_f32(const float *pti, int N, const float *db, int nvec, const int64_t *db_offsets,
                           float *Cres) {
  int64_t ioff_lo[ORC_LUT_MAXDIM], ioff_hi[ORC_LUT_MAXDIM];
  float wlo[ORC_LUT_MAXDIM], whi[ORC_LUT_MAXDIM];
  int Ninterp = 0;
  int64_t ofs_base = 1;
  for (int d = 0; d < N; ++d) {
    if (dim_needs_interpolation(pti[d])) {
      const int b = (int)pti[d];
      whi[Ninterp] = pti[d] - (float)b;
      wlo[Ninterp] = 1.0f - whi[Ninterp];
      ioff_lo[Ninterp] = db_offsets[d] * (b - 1);
      ioff_hi[Ninterp] = db_offsets[d] * b;
      Ninterp++;
    } else {
      ofs_base += db_offsets[d] * ((int64_t)lroundf(pti[d]) - 1); /* nint */
    }
  }
  for (int v = 0; v < nvec; ++v) Cres[v] = 0.0f;
  for (int b = 0; b < (1 << Ninterp); ++b) {
    int64_t ofs = ofs_base;
    float w = 1.0f;
    for (int d = 0; d < Ninterp; ++d) {
      if (b & (1 << d)) {
        ofs += ioff_hi[d];
        w = w * whi[d];
      } else {
        ofs += ioff_lo[d];
        w = w * wlo[d];
      }
    }
    const float *col = db + (size_t)(ofs - 1) * nvec;
    for (int v = 0; v < nvec; ++v) Cres[v] = Cres[v] + w * col[v];
  }
}

/* get_coeff (diffuse branch): src/pprts_base.F90:1517-1533 -> get_coeff_cube src/optprop.F90:583-593
 * -> LUT_get_diff2diff src/optprop_LUT.F90:1560-1596 */
void orc_get_coeff_diff2diff(const orc_lut *lut, double kabs, double ksca, double g, double dz, double dx,
                             float *out) {
  float aspect_zx = (float)(dz / dx);
  float w0 = (float)(ksca / fmax(kabs + ksca, DBL_EPSILON));
  float tauz = (float)((kabs + ksca) * dz);
  /* dims: 1 tau, 2 w0, 3 aspect_zx, 4 g */
  const float a_lo = lut->axis[2][0];
  aspect_zx = fmaxf(a_lo, aspect_zx);
  tauz = fmaxf(lut->axis[0][0], fminf(lut->axis[0][lut->n[0] - 1], tauz));
  w0 = fmaxf(lut->axis[1][0], fminf(lut->axis[1][lut->n[1] - 1], w0));
  const float sample[4] = {tauz, w0, aspect_zx, (float)g};
  float pti[4];
  int64_t offs[4];
  for (int d = 0; d < 4; ++d) pti[d] = orc_search_sorted_bisection_f32(lut->axis[d], lut->n[d], sample[d]);
  orc_ndarray_offsets(lut->n, 4, offs);
  orc_interp_vec_nd_f32(pti, 4, lut->table, lut->nvec, offs, out);
}

/* alloc_coeff_diff2diff: src/pprts.F90:3433-3462 (coeffs(:,k,i,j) = real(v, ireals)) */
void orc_alloc_coeff_diff2diff(const orc_lut *lut, int Nz, int xm, int ym, const double *kabs, const double *ksca,
                               const double *g, const double *dz, double dx, const uint8_t *l1d, double *coeffs) {
  float *v = (float *)malloc(sizeof(float) * (size_t)lut->nvec);
  for (int k = 0; k < Nz; ++k)
    for (int j = 0; j < ym; ++j)
      for (int i = 0; i < xm; ++i) {
        if (l1d[k]) continue;
        const size_t c3 = (size_t)k + (size_t)Nz * ((size_t)i + (size_t)xm * j);
        orc_get_coeff_diff2diff(lut, kabs[c3], ksca[c3], g[c3], dz[c3], dx, v);
        for (int q = 0; q < lut->nvec; ++q) coeffs[(size_t)lut->nvec * c3 + q] = (double)v[q];
      }
  free(v);
}

/* delta_scale / delta_scale_optprop: src/helper_functions.fypp:1622-1666 */
void orc_delta_scale(double *kabs, double *ksca, double *g, int have_f, double f_in) {
  double f = have_f ? f_in : (*g) * (*g);
  double dtau = *kabs + *ksca;
  if (dtau < DBL_EPSILON) return;
  double w0 = *ksca / dtau;
  double gg = *g;
  if (gg >= 1.0 - DBL_EPSILON * 10) {
    dtau = dtau * (1.0 - w0);
    w0 = 0.0;
    gg = 0.0;
  } else {
    dtau = dtau * (1.0 - w0 * f);
    gg = (gg - f) / (1.0 - f);
    w0 = w0 * (1.0 - f) / (1.0 - f * w0);
  }
  *g = gg;
  *kabs = dtau * (1.0 - w0);
  *ksca = dtau * w0;
}

/* approx(): src/helper_functions.fypp:1272-1287 */
static int approx_d(double a, double b) {
  const double factor = 10.0 * DBL_EPSILON;
  return a <= b + factor && a >= b - factor;
}

/* eddington_coeff_ec: src/eddington.F90:173-241 (irealeddington = real64) */
void orc_eddington_coeff_ec(double dtau, double w0, double g, double mu0, double *t, double *r, double *rdir,
                            double *sdir, double *tdir) {
  const double f = 0.75 * g;
  const double g1 = 2.0 - w0 * (1.25 + f);
  const double g2 = w0 * (0.75 - f);
  const double g3 = 0.5 - mu0 * f;
  const double dtau_slant = fmax(dtau / fmax(sqrt(DBL_MIN), mu0), 0.0);
  if (dtau_slant > 1e-6) {
    const double g4 = 1.0 - g3;
    const double alpha1 = g1 * g4 + g2 * g3;
    const double alpha2 = g1 * g3 + g2 * g4;
    const double A = sqrt(fmax((g1 - g2) * (g1 + g2), 1e-12));
    double k_mu0 = A * mu0;
    const double k_g3 = A * g3, k_g4 = A * g4;
    const double e0 = exp(-dtau_slant);
    *tdir = e0;
    const double e = exp(-A * dtau);
    const double e2 = e * e;
    const double k_2_e = 2 * A * e;
    if (approx_d(k_mu0, 1.0)) k_mu0 = 1 - 10 * DBL_EPSILON;
    double beta = 1 / (A + g1 + (A - g1) * e2);
    *r = g2 * (1 - e2) * beta;
    *t = k_2_e * beta;
    beta = w0 * beta / (1 - k_mu0 * k_mu0);
    *sdir = beta * (k_2_e * (g4 + alpha1 * mu0) - e0 * ((1 + k_mu0) * (alpha1 + k_g4) - (1 - k_mu0) * (alpha1 - k_g4) * e2));
    *rdir = beta * ((1 - k_mu0) * (alpha2 + k_g3) - (1 + k_mu0) * (alpha2 - k_g3) * e2 - k_2_e * (g3 - alpha2 * mu0) * e0);
  } else {
    *t = 1.0 - g1 * dtau;
    *r = g2 * dtau;
    *sdir = (1.0 - g3) * (w0 * dtau);
    *rdir = g3 * (w0 * dtau);
    *tdir = 1.0 - dtau_slant;
  }
}

/* B_eff: src/schwarzschild.F90:36-67; dgauss(2) on (0,1): nodes 1/2 -+ 1/(2 sqrt 3), weights 1/2 */
double orc_B_eff(double B_far, double B_near, double tau) {
  const double pt[2] = {0.5 - 0.5 / sqrt(3.0), 0.5 + 0.5 / sqrt(3.0)};
  const double wi[2] = {0.5, 0.5};
  double B = 0;
  for (int q = 0; q < 2; ++q) {
    const double mu = pt[q];
    const double dtau = tau / mu;
    double bmu;
    if (dtau < 1e-3) {
      bmu = (B_far + B_near) * .5;
    } else {
      const double tm1 = expm1(-dtau);
      bmu = (-B_near + B_far * (tm1 + 1)) / (tm1) + ((B_far - B_near) * mu) / tau;
    }
    B += bmu * mu * wi[q];
  }
  return B * 2;
}
