/*
 * pprts_oracle_pipe.c -- direct beam, source term and post-processing around the diffuse solve.
 * TEST INFRASTRUCTURE ONLY (parity oracle); see pprts_oracle.h.  One periodic rank.
 */
#include <float.h>
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <stdlib.h>
#include <string.h>

#include "pprts_oracle_phys.h"

#define GI(D, L, gxm, d, k, i, j) \
  ((size_t)(d) + (size_t)(D) * ((size_t)(k) + (size_t)(L) * ((size_t)((i) + 1) + (size_t)(gxm) * (size_t)((j) + 1))))
#define OI(D, L, xm, d, k, i, j) \
  ((size_t)(d) + (size_t)(D) * ((size_t)(k) + (size_t)(L) * ((size_t)(i) + (size_t)(xm) * (size_t)(j))))
#define C3(Nz, xm, k, i, j) ((size_t)(k) + (size_t)(Nz) * ((size_t)(i) + (size_t)(xm) * (size_t)(j)))

void orc_dir_layout_3_10(orc_dir_layout *d) {
  d->dtop = 1;
  d->dside = 1;
  d->top_div = 1;
  d->side_div = 1;
}

/* t_solver_8_16: dirtop 4 streams (area_divider 4), dirside 2 (area_divider 2), src/pprts.F90:413-425 */
void orc_dir_layout_8_16(orc_dir_layout *d) {
  d->dtop = 4;
  d->dside = 2;
  d->top_div = 4;
  d->side_div = 2;
}

static double deg2rad(double x) { return x * M_PI / 180.0; }
static double rad2deg(double x) { return x * 180.0 / M_PI; }

/* setup_suninfo: src/pprts.F90:1118-1183 */
void orc_setup_suninfo(double phi, double theta, orc_suninfo *sun) {
  sun->phi = phi;
  sun->theta = theta;
  sun->mu = fmax(cos(deg2rad(theta)), 0.0);
  if (sun->theta >= 90.0) sun->theta = -1.0;
  sun->costheta = fmax(cos(deg2rad(sun->theta)), 0.0);
  double s = acos(cos(deg2rad(phi)));
  sun->symmetry_phi = fmin(90.0, fmax(0.0, rad2deg(asin(sin(s)))));
  sun->xinc = sin(deg2rad(phi)) > 0.0 ? 0 : 1;
  sun->yinc = cos(deg2rad(phi)) < 0.0 ? 1 : 0;
}

/* The stream relabelling for a sun in the east / north half (get_coeff_cube, src/optprop.F90:571-576), statement by
   statement as the reference's routines assign it -- `coeff(dst block) = newcoeff(src permutation + other dst block)`,
   east first, then north.  A dst block without an (uncommented) assignment in the reference keeps its values,
   source order included.  Pinned to tests/golden/coeff_symmetry.json (the reference's text, interpreted).
     3_10 dir2dir : dir2dir_coeff_symmetry_none        (:1256-1266)   nothing
     3_10 dir2diff: dir3_to_diff10_coeff_symmetry      (:1009-1045)   east: dst 3<->4, 5<->6; north: 7<->8, 9<->10
     8_16 dir2dir : dir2dir8_coeff_symmetry            (:1268-1302)   all 8 dst blocks assigned, src [2,1,4,3,..] / [3,4,1,2,..]
     8_16 dir2diff: dir8_to_diff16_coeff_symmetry      (:1186-1240)   east: dst 3,4,7,8,9..12 assigned; north: 1,2,5,6,13..16 */
void orc_dir_coeff_symmetry(int is_dir2dir, int S, int D, int lswitch_east, int lswitch_north, float *coeff) {
  if (S == 3) {
    if (is_dir2dir || D != 10) return;
    /* from[b] = 1-based block read for dst block b+1; 0 = no assignment */
    static const int e3[10] = {0, 0, 4, 3, 6, 5, 0, 0, 0, 0}, n3[10] = {0, 0, 0, 0, 0, 0, 8, 7, 10, 9};
    float nw[30];
    for (int pass = 0; pass < 2; ++pass) {
      if (!(pass == 0 ? lswitch_east : lswitch_north)) continue;
      const int *from = pass == 0 ? e3 : n3;
      memcpy(nw, coeff, sizeof(nw));
      for (int b = 0; b < 10; ++b)
        if (from[b])
          for (int q = 0; q < 3; ++q) coeff[b * 3 + q] = nw[(from[b] - 1) * 3 + q];
    }
    return;
  }
  if (S != 8) return;
  const int nb = is_dir2dir ? 8 : 16;
  static const int src_e[8] = {2, 1, 4, 3, 5, 6, 7, 8}, src_n[8] = {3, 4, 1, 2, 5, 6, 7, 8};
  static const int t_e[8] = {2, 1, 4, 3, 5, 6, 7, 8}, t_n[8] = {3, 4, 1, 2, 5, 6, 7, 8};
  static const int s_e[16] = {0, 0, 7, 8, 0, 0, 3, 4, 10, 9, 12, 11, 0, 0, 0, 0};
  static const int s_n[16] = {5, 6, 0, 0, 1, 2, 0, 0, 0, 0, 0, 0, 14, 13, 16, 15};
  float nw[128];
  for (int pass = 0; pass < 2; ++pass) {
    if (!(pass == 0 ? lswitch_east : lswitch_north)) continue;
    const int *from = is_dir2dir ? (pass == 0 ? t_e : t_n) : (pass == 0 ? s_e : s_n);
    const int *sp = pass == 0 ? src_e : src_n;
    memcpy(nw, coeff, sizeof(float) * (size_t)nb * 8);
    for (int b = 0; b < nb; ++b)
      if (from[b])
        for (int q = 0; q < 8; ++q) coeff[b * 8 + q] = nw[(from[b] - 1) * 8 + (sp[q] - 1)];
  }
}

/* get_coeff direct branch; clamps on dirconfig dims (src/pprts_base.F90:1521-1526) */
void orc_get_coeff_dir(const orc_lut *lut, int is_dir2dir, int S, int D, double kabs, double ksca, double g, double dz,
                       double dx, double sym_phi, double theta, int lswitch_east, int lswitch_north, float *out) {
  float aspect_zx = (float)(dz / dx);
  float w0 = (float)(ksca / fmax(kabs + ksca, DBL_EPSILON));
  float tauz = (float)((kabs + ksca) * dz);
  aspect_zx = fmaxf(lut->axis[2][0], aspect_zx);
  tauz = fmaxf(lut->axis[0][0], fminf(lut->axis[0][lut->n[0] - 1], tauz));
  w0 = fmaxf(lut->axis[1][0], fminf(lut->axis[1][lut->n[1] - 1], w0));
  const float sample[6] = {tauz, w0, aspect_zx, (float)g, (float)sym_phi, (float)theta};
  float pti[6];
  int64_t offs[6];
  for (int d = 0; d < 6; ++d) pti[d] = orc_search_sorted_bisection_f32(lut->axis[d], lut->n[d], sample[d]);
  orc_ndarray_offsets(lut->n, 6, offs);
  orc_interp_vec_nd_f32(pti, 6, lut->table, lut->nvec, offs, out);
  orc_dir_coeff_symmetry(is_dir2dir, S, D, lswitch_east, lswitch_north, out);
}

/* alloc_coeff_dir2dir / dir2diff: src/pprts.F90:3129-3184, 3280-3391 */
void orc_alloc_coeff_dir(const orc_lut *lut, int is_dir2dir, int S, int D, int Nz, int xm, int ym, const double *kabs,
                         const double *ksca, const double *g, const double *dz, double dx, const orc_suninfo *sun,
                         const uint8_t *l1d, double *coeffs) {
  float *v = (float *)malloc(sizeof(float) * (size_t)lut->nvec);
  for (int k = 0; k < Nz; ++k)
    for (int j = 0; j < ym; ++j)
      for (int i = 0; i < xm; ++i) {
        if (l1d[k]) continue;
        const size_t c3 = C3(Nz, xm, k, i, j);
        orc_get_coeff_dir(lut, is_dir2dir, S, D, kabs[c3], ksca[c3], g[c3], dz[c3], dx, sun->symmetry_phi, sun->theta,
                          sun->xinc == 0, sun->yinc == 0, v);
        for (int q = 0; q < lut->nvec; ++q) coeffs[(size_t)lut->nvec * c3 + q] = (double)v[q];
      }
  free(v);
}

/* exchange_direct_boundary with self neighbours: src/pprts_explicit.F90:232-328 */
static void exchange_direct_boundary_1rank(int S, int dtop, int dside, int L, int xm, int ym, int lsun_north, int lsun_east,
                                           double *x0) {
  const int gxm = xm + 2;
  for (int j = 0; j < ym; ++j)
    for (int k = 0; k < L; ++k)
      for (int q = 0; q < dside; ++q) {
        const int dof = dtop + q;
        if (lsun_east) x0[GI(S, L, gxm, dof, k, xm, j)] = x0[GI(S, L, gxm, dof, k, 0, j)];
        else x0[GI(S, L, gxm, dof, k, 0, j)] = x0[GI(S, L, gxm, dof, k, xm, j)];
      }
  for (int i = 0; i < xm; ++i)
    for (int k = 0; k < L; ++k)
      for (int q = 0; q < dside; ++q) {
        const int dof = dtop + dside + q;
        if (lsun_north) x0[GI(S, L, gxm, dof, k, i, ym)] = x0[GI(S, L, gxm, dof, k, i, 0)];
        else x0[GI(S, L, gxm, dof, k, i, 0)] = x0[GI(S, L, gxm, dof, k, i, ym)];
      }
}

/* explicit_edir_forward_sweep: src/pprts_explicit.F90:330-459 (lopen_bc = false) */
static void edir_forward_sweep(int S, int dtop, int dside, int Nz, int xm, int ym, int xinc, int yinc, const int dx[3],
                               const int dy[3], const double *coeffs, const uint8_t *l1d, const double *a33,
                               const double *xb, double *x0) {
  const int L = Nz + 1, gxm = xm + 2;
  for (int j = 0; j < ym; ++j)
    for (int i = 0; i < xm; ++i)
      for (int d = 0; d < dtop; ++d) x0[GI(S, L, gxm, d, 0, i, j)] = xb[GI(S, L, gxm, d, 0, i, j)];
  for (int k = 0; k < Nz; ++k) {
    if (l1d[k]) {
      for (int j = dy[0]; j != dy[1] + dy[2]; j += dy[2])
        for (int i = dx[0]; i != dx[1] + dx[2]; i += dx[2])
          for (int d = 0; d < dtop; ++d) x0[GI(S, L, gxm, d, k + 1, i, j)] = x0[GI(S, L, gxm, d, k, i, j)] * a33[C3(Nz, xm, k, i, j)];
    } else {
      for (int j = dy[0]; j != dy[1] + dy[2]; j += dy[2])
        for (int i = dx[0]; i != dx[1] + dx[2]; i += dx[2]) {
          const double *v = coeffs + (size_t)S * S * C3(Nz, xm, k, i, j); /* v(src,dst) = v[dst*S+src] */
          int dst = 0;
          for (int part = 0; part < 3; ++part) {
            const int nd = part == 0 ? dtop : dside;
            for (int q = 0; q < nd; ++q, ++dst) {
              size_t ob;
              if (part == 0) ob = GI(S, L, gxm, dst, k + 1, i, j);
              else if (part == 1) ob = GI(S, L, gxm, dst, k, i + xinc, j);
              else ob = GI(S, L, gxm, dst, k, i, j + yinc);
              x0[ob] = 0;
              int src = 0;
              for (int s = 0; s < dtop; ++s, ++src) x0[ob] += x0[GI(S, L, gxm, src, k, i, j)] * v[dst * S + src];
              for (int s = 0; s < dside; ++s, ++src) x0[ob] += x0[GI(S, L, gxm, src, k, i + 1 - xinc, j)] * v[dst * S + src];
              for (int s = 0; s < dside; ++s, ++src) x0[ob] += x0[GI(S, L, gxm, src, k, i, j + 1 - yinc)] * v[dst * S + src];
            }
          }
        }
    }
  }
}

/* setup_incSolar (src/pprts_base.F90:1146-1181) + explicit_edir (src/pprts_explicit.F90:60-229) */
int orc_explicit_edir_1rank(const orc_layout *l, const orc_dir_layout *d, const orc_suninfo *sun, const double *dir2dir,
                            const uint8_t *l1d, const double *a33, double edirTOA, double dxm, double dym, double rtol,
                            double atol, int maxit, double *vedir, int *niter) {
  const int S = d->dtop + 2 * d->dside, Nz = l->Nz, L = Nz + 1, xm = l->xm, ym = l->ym, gxm = xm + 2;
  const size_t ng = (size_t)S * L * gxm * (ym + 2);
  double *lb = (double *)calloc(ng, sizeof(double));
  double *v0 = (double *)calloc(ng, sizeof(double));
  const double fac = edirTOA * dxm * dym / (double)d->top_div;
  for (int j = 0; j < ym; ++j)
    for (int i = 0; i < xm; ++i) {
      for (int s = 0; s < d->dtop; ++s) lb[GI(S, L, gxm, s, 0, i, j)] = fac;
      memcpy(v0 + GI(S, L, gxm, 0, 0, i, j), vedir + OI(S, L, xm, 0, 0, i, j), sizeof(double) * (size_t)S * L);
    }
  const int lsun_north = sun->yinc == 0, lsun_east = sun->xinc == 0;
  int dx[3] = {0, xm - 1, 1}, dy[3] = {0, ym - 1, 1};
  if (lsun_east) { dx[0] = xm - 1; dx[1] = 0; dx[2] = -1; }   /* -explicit_edir_permute default true :149-156 */
  if (lsun_north) { dy[0] = ym - 1; dy[1] = 0; dy[2] = -1; }
  exchange_direct_boundary_1rank(S, d->dtop, d->dside, L, xm, ym, lsun_north, lsun_east, v0);
  double res1 = 0;
  int iter, converged = 0;
  for (iter = 1; iter <= maxit; ++iter) {
    edir_forward_sweep(S, d->dtop, d->dside, Nz, xm, ym, sun->xinc, sun->yinc, dx, dy, dir2dir, l1d, a33, lb, v0);
    exchange_direct_boundary_1rank(S, d->dtop, d->dside, L, xm, ym, lsun_north, lsun_east, v0);
    double s = 0;
    for (int j = 0; j < ym; ++j)
      for (int i = 0; i < xm; ++i)
        for (int q = 0; q < S * L; ++q) {
          const size_t og = OI(S, L, xm, 0, 0, i, j) + q, gg = GI(S, L, gxm, 0, 0, i, j) + q;
          const double df = vedir[og] - v0[gg];
          s += df * df;
          vedir[og] = v0[gg];
        }
    double res = fmax(DBL_MIN, sqrt(s));
    if (iter == 1) res1 = res;
    const double rel = res1 <= sqrt(DBL_MIN) ? 0.0 : res / res1;
    if (res < atol || rel < rtol) {
      converged = 1;
      break;
    }
  }
  if (niter) *niter = iter <= maxit ? iter : maxit;
  free(lb);
  free(v0);
  return converged ? 0 : 1;
}

/* setup_b solar: src/pprts.F90:4684-4846 + halo_fill(edir), halo_reduce(b) with self neighbours */
void orc_setup_b_solar_1rank(const orc_layout *l, const orc_dir_layout *d, const orc_suninfo *sun, const double *dir2diff,
                             const uint8_t *l1d, const double *a13, const double *a23, const double *albedo,
                             const double *edir, double *b) {
  const int D = orc_D(l), S = d->dtop + 2 * d->dside, Nz = l->Nz, L = Nz + 1, xm = l->xm, ym = l->ym, gxm = xm + 2;
  const int ntop = l->ntop, nside = l->nside;
  const double streams = (double)(ntop / 2);
  double *xs = (double *)calloc((size_t)D * L * gxm * (ym + 2), sizeof(double));
  double *xe = (double *)calloc((size_t)S * L * gxm * (ym + 2), sizeof(double));
  orc_owned_to_ghosted(S, L, xm, ym, edir, xe);
  orc_halo_fill_1rank(S, L, xm, ym, xe);
  for (int j = 0; j < ym; ++j)
    for (int i = 0; i < xm; ++i)
      for (int k = 0; k < Nz; ++k) {
        int any = 0;
        for (int s = 0; s < S; ++s) any |= xe[GI(S, L, gxm, s, k, i, j)] > DBL_EPSILON;
        if (!any) continue;
        const size_t c3 = C3(Nz, xm, k, i, j);
        if (l1d[k]) {
          for (int src = 0; src < d->dtop; ++src)
            for (int q = 0; q < ntop; ++q) {
              if (l->top_inward[q]) xs[GI(D, L, gxm, q, k + 1, i, j)] += xe[GI(S, L, gxm, src, k, i, j)] * a23[c3] / streams;
              else xs[GI(D, L, gxm, q, k, i, j)] += xe[GI(S, L, gxm, src, k, i, j)] * a13[c3] / streams;
            }
        } else {
          const double *c = dir2diff + (size_t)S * D * c3; /* dir2diff(dst*S + src) */
          int dst = 0;
          for (int part = 0; part < 3; ++part) {
            const int nd = part == 0 ? ntop : nside;
            for (int q = 0; q < nd; ++q, ++dst) {
              size_t ob;
              if (part == 0) ob = GI(D, L, gxm, dst, l->top_inward[q] ? k + 1 : k, i, j);
              else if (part == 1) ob = GI(D, L, gxm, dst, k, l->side_inward[q] ? i + 1 : i, j);
              else ob = GI(D, L, gxm, dst, k, i, l->side_inward[q] ? j + 1 : j);
              int src = 0;
              for (int s = 0; s < d->dtop; ++s, ++src) xs[ob] += xe[GI(S, L, gxm, src, k, i, j)] * c[dst * S + src];
              for (int s = 0; s < d->dside; ++s, ++src) xs[ob] += xe[GI(S, L, gxm, src, k, i + 1 - sun->xinc, j)] * c[dst * S + src];
              for (int s = 0; s < d->dside; ++s, ++src) xs[ob] += xe[GI(S, L, gxm, src, k, i, j + 1 - sun->yinc)] * c[dst * S + src];
            }
          }
        }
      }
  for (int j = 0; j < ym; ++j) /* ground albedo reflecting direct radiation :4829-4843 */
    for (int i = 0; i < xm; ++i)
      for (int q = 0; q < ntop; ++q)
        if (!l->top_inward[q])
          for (int src = 0; src < d->dtop; ++src)
            xs[GI(D, L, gxm, q, Nz, i, j)] += xe[GI(S, L, gxm, src, Nz, i, j)] * albedo[i + (size_t)xm * j] / streams;
  orc_halo_reduce_1rank(D, L, xm, ym, xs);
  orc_ghosted_to_owned(D, L, xm, ym, xs, b);
  free(xs);
  free(xe);
}

/* setup_b thermal: src/pprts.F90:4848-4987 (no collapse, planck at levels).  Surface emission (:4958-4985): from
 * planck_srfc (xm, ym) = atm%Bsrfc when the caller of set_optical_properties gave it (src/pprts.F90:1823-1829; the rrtmg
 * driver always does, rrtmg/rrtmg/pprts_rrtmg.F90:609-642, 681) with the emissivity 1 - albedo clamped to [0, 1] (:4965-4966),
 * else (planck_srfc == NULL) from planck(ze) with the unclamped 1 - albedo (:4971-4984) */
void orc_setup_b_thermal_1rank(const orc_layout *l, const double *diff2diff, const uint8_t *l1d, const double *a11,
                               const double *a12, const double *albedo, const double *planck, const double *planck_srfc,
                               const double *kabs, const double *dz, double dxm, double dym, double *b) {
  const int D = orc_D(l), Nz = l->Nz, L = Nz + 1, xm = l->xm, ym = l->ym, gxm = xm + 2;
  const int ntop = l->ntop, nside = l->nside;
  const double tstreams = (double)(ntop / 2), sstreams = (double)(nside / 2);
  const double Az = dxm * dym; /* difftop%area_divider = 1 */
  double *xs = (double *)calloc((size_t)D * L * gxm * (ym + 2), sizeof(double));
  for (int j = 0; j < ym; ++j)
    for (int i = 0; i < xm; ++i)
      for (int k = 0; k < Nz; ++k) {
        const size_t c3 = C3(Nz, xm, k, i, j);
        const double b0 = planck[(size_t)k + (size_t)L * (i + (size_t)xm * j)];
        const double b1 = planck[(size_t)k + 1 + (size_t)L * (i + (size_t)xm * j)];
        const double tauz = kabs[c3] * dz[c3];
        const double btop0 = orc_B_eff(b1, b0, tauz), bbot0 = orc_B_eff(b0, b1, tauz);
        if (l1d[k]) {
          const double bfac = M_PI * Az / tstreams;
          double emis = 1.0 - a11[c3] - a12[c3];
          emis = fmax(0.0, fmin(1.0, emis));
          for (int q = 0; q < ntop; ++q) {
            if (l->top_inward[q]) xs[GI(D, L, gxm, q, k + 1, i, j)] += bbot0 * bfac * emis;
            else xs[GI(D, L, gxm, q, k, i, j)] += btop0 * bfac * emis;
          }
        } else {
          const double Ax = dym * dz[c3], Ay = dxm * dz[c3];
          const double *c = diff2diff + (size_t)D * D * c3; /* c(src,dst) = c[dst*D+src] */
          int src = 0;
          double bfac = M_PI * Az / tstreams;
          for (int q = 0; q < ntop; ++q, ++src) {
            double sum = 0;
            for (int dd = 0; dd < D; ++dd) sum += c[dd * D + src];
            double emis = fmax(0.0, fmin(1.0, 1.0 - sum));
            if (!l->top_inward[q]) xs[GI(D, L, gxm, src, k, i, j)] += btop0 * bfac * emis;
            else xs[GI(D, L, gxm, src, k + 1, i, j)] += bbot0 * bfac * emis;
          }
          for (int part = 1; part < 3; ++part) {
            bfac = M_PI * (part == 1 ? Ax : Ay) / sstreams;
            for (int q = 0; q < nside; ++q, ++src) {
              double sum = 0;
              for (int dd = 0; dd < D; ++dd) sum += c[dd * D + src];
              double emis = fmax(0.0, fmin(1.0, 1.0 - sum));
              emis = (q + 1 > nside / 2 ? btop0 : bbot0) * emis;
              if (!l->side_inward[q]) xs[GI(D, L, gxm, src, k, i, j)] += emis * bfac;
              else if (part == 1) xs[GI(D, L, gxm, src, k, i + 1, j)] += emis * bfac;
              else xs[GI(D, L, gxm, src, k, i, j + 1)] += emis * bfac;
            }
          }
        }
      }
  for (int j = 0; j < ym; ++j)
    for (int i = 0; i < xm; ++i)
      for (int q = 0; q < ntop; ++q)
        if (!l->top_inward[q]) {
          if (planck_srfc) { /* allocated(atm%Bsrfc) :4960-4970 */
            double emis = 1.0 - albedo[i + (size_t)xm * j];
            emis = fmax(0.0, fmin(1.0, emis));
            xs[GI(D, L, gxm, q, Nz, i, j)] += planck_srfc[i + (size_t)xm * j] * Az * emis * M_PI / tstreams;
          } else /* Bsrfc not allocated: planck(ze) :4971-4984 */
            xs[GI(D, L, gxm, q, Nz, i, j)] += planck[(size_t)Nz + (size_t)L * (i + (size_t)xm * j)] * Az *
                                              (1.0 - albedo[i + (size_t)xm * j]) * M_PI / tstreams;
        }
  orc_halo_reduce_1rank(D, L, xm, ym, xs);
  orc_ghosted_to_owned(D, L, xm, ym, xs, b);
  free(xs);
}

/* gen_scale_diff_flx_vec_arr: src/pprts.F90:3945-3987 (note: Ay is divided by difftop%area_divider there) */
void orc_scale_diff(const orc_layout *l, const double *dz, double dxm, double dym, int to_Wm2, double *e) {
  const int D = orc_D(l), Nz = l->Nz, L = Nz + 1, xm = l->xm, ym = l->ym, ntop = l->ntop, nside = l->nside;
  for (int j = 0; j < ym; ++j)
    for (int i = 0; i < xm; ++i)
      for (int k = 0; k < L; ++k)
        for (int dd = 0; dd < D; ++dd) {
          double v;
          if (dd < ntop) v = dxm * dym;
          else if (k == Nz) v = 1.0;
          else if (dd < ntop + nside) v = dym * dz[C3(Nz, xm, k, i, j)];
          else v = dxm * dz[C3(Nz, xm, k, i, j)];
          const size_t o = OI(D, L, xm, dd, k, i, j);
          e[o] = to_Wm2 ? (v != 0.0 ? e[o] * (1.0 / v) : 0.0) : e[o] * v;
        }
}
void orc_scale_dir(const orc_layout *l, const orc_dir_layout *d, const double *dz, double dxm, double dym, int to_Wm2,
                   double *e) {
  const int S = d->dtop + 2 * d->dside, Nz = l->Nz, L = Nz + 1, xm = l->xm, ym = l->ym;
  for (int j = 0; j < ym; ++j)
    for (int i = 0; i < xm; ++i)
      for (int k = 0; k < L; ++k)
        for (int dd = 0; dd < S; ++dd) {
          double v;
          if (dd < d->dtop) v = dxm * dym / (double)d->top_div;
          else if (k == Nz) v = 1.0;
          else if (dd < d->dtop + d->dside) v = dym * dz[C3(Nz, xm, k, i, j)] / (double)d->side_div;
          else v = dxm * dz[C3(Nz, xm, k, i, j)] / (double)d->side_div;
          const size_t o = OI(S, L, xm, dd, k, i, j);
          e[o] = to_Wm2 ? e[o] * (1.0 / v) : e[o] * v;
        }
}

/* calc_flx_div, by_coeff_divergence: src/pprts.F90:5286-5398; volume scaling :5477, 5483-5503 */
void orc_calc_flx_div_1rank(const orc_layout *l, const orc_dir_layout *d, const orc_suninfo *sun, const double *dir2dir,
                            const double *dir2diff, const double *diff2diff, const uint8_t *l1d, const double *a11,
                            const double *a12, const double *kabs, const double *dz, double dxm, double dym,
                            const double *edir, const double *ediff, const double *b_thermal, double *abso) {
  const int D = orc_D(l), Nz = l->Nz, L = Nz + 1, xm = l->xm, ym = l->ym, gxm = xm + 2;
  const int ntop = l->ntop, nside = l->nside;
  const int S = d ? d->dtop + 2 * d->dside : 0;
  double *led = (double *)calloc((size_t)D * L * gxm * (ym + 2), sizeof(double));
  orc_owned_to_ghosted(D, L, xm, ym, ediff, led);
  orc_halo_fill_1rank(D, L, xm, ym, led);
  double *ledir = NULL, *lsrc = NULL;
  if (edir) {
    ledir = (double *)calloc((size_t)S * L * gxm * (ym + 2), sizeof(double));
    orc_owned_to_ghosted(S, L, xm, ym, edir, ledir);
    orc_halo_fill_1rank(S, L, xm, ym, ledir);
  }
  if (b_thermal) {
    lsrc = (double *)calloc((size_t)D * L * gxm * (ym + 2), sizeof(double));
    orc_owned_to_ghosted(D, L, xm, ym, b_thermal, lsrc);
    orc_halo_fill_1rank(D, L, xm, ym, lsrc);
  }
  for (int j = 0; j < ym; ++j)
    for (int i = 0; i < xm; ++i)
      for (int k = 0; k < Nz; ++k) {
        const size_t c3 = C3(Nz, xm, k, i, j);
        double a = 0;
        if (edir) {
          if (l1d[k]) {
            for (int s = 0; s < d->dtop; ++s) {
              const double cdiv = kabs[c3] * dz[c3] / sun->costheta;
              a += ledir[GI(S, L, gxm, s, k, i, j)] * (-expm1(-cdiv));
            }
          } else {
            const double *t = dir2dir + (size_t)S * S * c3;  /* (src,dst): t[dst*S+src] */
            const double *sd = dir2diff + (size_t)S * D * c3;
            int idof = 0;
            for (int part = 0; part < 3; ++part) {
              const int ns = part == 0 ? d->dtop : d->dside;
              for (int q = 0; q < ns; ++q, ++idof) {
                double cdiv = 1.0, s1 = 0, s2 = 0;
                for (int dd = 0; dd < S; ++dd) s1 += t[dd * S + idof];
                for (int dd = 0; dd < D; ++dd) s2 += sd[dd * S + idof];
                cdiv = cdiv - s1 - s2;
                double e;
                if (part == 0) e = ledir[GI(S, L, gxm, idof, k, i, j)];
                else if (part == 1) e = ledir[GI(S, L, gxm, idof, k, i + 1 - sun->xinc, j)];
                else e = ledir[GI(S, L, gxm, idof, k, i, j + 1 - sun->yinc)];
                a += e * cdiv;
              }
            }
          }
        }
        if (!l1d[k]) {
          const double *c = diff2diff + (size_t)D * D * c3;
          int idof = 0;
          for (int part = 0; part < 3; ++part) {
            const int ns = part == 0 ? ntop : nside;
            for (int q = 0; q < ns; ++q, ++idof) {
              double sum = 0;
              for (int dd = 0; dd < D; ++dd) sum += c[dd * D + idof];
              const double cdiv = 1.0 - sum;
              double e;
              if (part == 0) e = led[GI(D, L, gxm, idof, l->top_inward[q] ? k : k + 1, i, j)];
              else if (part == 1) e = led[GI(D, L, gxm, idof, k, l->side_inward[q] ? i : i + 1, j)];
              else e = led[GI(D, L, gxm, idof, k, i, l->side_inward[q] ? j : j + 1)];
              a += e * cdiv;
            }
          }
        } else {
          const double cdiv = fmax(0.0, 1.0 - a11[c3] - a12[c3]);
          for (int q = 0; q < ntop; ++q) a += led[GI(D, L, gxm, q, l->top_inward[q] ? k : k + 1, i, j)] * cdiv;
        }
        if (b_thermal) {
          int idof = 0;
          for (int q = 0; q < ntop; ++q, ++idof) a -= lsrc[GI(D, L, gxm, idof, l->top_inward[q] ? k + 1 : k, i, j)];
          for (int q = 0; q < nside; ++q, ++idof) a -= lsrc[GI(D, L, gxm, idof, k, l->side_inward[q] ? i + 1 : i, j)];
          for (int q = 0; q < nside; ++q, ++idof) a -= lsrc[GI(D, L, gxm, idof, k, i, l->side_inward[q] ? j + 1 : j)];
        }
        abso[c3] = a * (1.0 / (dxm * dym * dz[c3]));
      }
  free(led);
  free(ledir);
  free(lsrc);
}

/* pprts_get_result: src/pprts.F90:5850-5888.  edir (L,xm,ym), edn/eup (L,xm,ym), abso (Nz,xm,ym) */
void orc_get_result(const orc_layout *l, const orc_dir_layout *d, const orc_suninfo *sun, int lsolar, const double *edir,
                    const double *ediff, const double *abso, double *redir, double *redn, double *reup, double *rabso) {
  const int D = orc_D(l), Nz = l->Nz, L = Nz + 1, xm = l->xm, ym = l->ym;
  const int S = d ? d->dtop + 2 * d->dside : 0;
  const double mu = lsolar ? sun->mu : 1.0;
  for (int j = 0; j < ym; ++j)
    for (int i = 0; i < xm; ++i) {
      for (int k = 0; k < L; ++k) {
        const size_t o = (size_t)k + (size_t)L * (i + (size_t)xm * j);
        double dn = 0, up = 0, di = 0;
        for (int q = 0; q < l->ntop; ++q) {
          if (l->top_inward[q]) dn += ediff[OI(D, L, xm, q, k, i, j)];
          else up += ediff[OI(D, L, xm, q, k, i, j)];
        }
        if (lsolar && redir) {
          for (int q = 0; q < d->dtop; ++q) di += edir[OI(S, L, xm, q, k, i, j)];
          redir[o] = di / (double)d->top_div * mu;
        } else if (redir) {
          redir[o] = 0;
        }
        redn[o] = dn * mu; /* difftop%area_divider = 1 */
        reup[o] = up * mu;
      }
      for (int k = 0; k < Nz; ++k) rabso[C3(Nz, xm, k, i, j)] = abso[C3(Nz, xm, k, i, j)] * mu;
    }
}
