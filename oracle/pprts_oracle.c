/*
 * pprts_oracle.c -- see pprts_oracle.h.  TEST INFRASTRUCTURE ONLY (parity oracle).
 * Every function cites the reference file:line it restates (paths relative to the
 * tenstream/tenstream tree).  Plain C99, scalar, double precision (ireals = real64).
 */
#include "pprts_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------ */
/* stream layouts: pprts.F90:332-349 (3_10), :413-425 (8_16)                                   */
void orc_layout_3_10(orc_layout *l, int Nz, int xm, int ym) {
  memset(l, 0, sizeof(*l));
  l->ntop = 2;
  l->nside = 4;
  int t[2] = {0, 1};
  int s[4] = {0, 1, 0, 1};
  memcpy(l->top_inward, t, sizeof(t));
  memcpy(l->side_inward, s, sizeof(s));
  l->Nz = Nz;
  l->xm = xm;
  l->ym = ym;
}
void orc_layout_8_16(orc_layout *l, int Nz, int xm, int ym) {
  memset(l, 0, sizeof(*l));
  l->ntop = 8;
  l->nside = 4;
  for (int i = 0; i < 8; ++i) l->top_inward[i] = i & 1;
  for (int i = 0; i < 4; ++i) l->side_inward[i] = i & 1;
  l->Nz = Nz;
  l->xm = xm;
  l->ym = ym;
}
int orc_D(const orc_layout *l) { return l->ntop + 2 * l->nside; }

/* inv_dof: pprts_shell.F90:527-540 (same in pprts.F90:5739-5752, pprts_explicit.F90:1001-1014) */
int orc_inv_dof(const orc_layout *l, int dof) {
  int inc = l->top_inward[0] ? 1 : -1;
  return l->top_inward[dof] ? dof + inc : dof - inc;
}
#define inv_dof orc_inv_dof

/* ghosted index helper: (d, k, i, j) with i,j in [-1, xm] / [-1, ym] */
#define GIDX(D, L, gxm, d, k, i, j) \
  ((size_t)(d) + (size_t)(D) * ((size_t)(k) + (size_t)(L) * ((size_t)((i) + 1) + (size_t)(gxm) * (size_t)((j) + 1))))
#define OIDX(D, L, xm, d, k, i, j) \
  ((size_t)(d) + (size_t)(D) * ((size_t)(k) + (size_t)(L) * ((size_t)(i) + (size_t)(xm) * (size_t)(j))))

/* ------------------------------------------------------------------------------------------ */
/* op_mat_mult_ediff, the cell loop: pprts_shell.F90:413-508                                    */
void orc_op_mat_mult_ediff_local(const orc_layout *l, const double *diff2diff, const uint8_t *l1d,
                                 const double *a11, const double *a12, const double *albedo,
                                 const double *xx, double *xb) {
  const int D = orc_D(l), L = l->Nz + 1, Nz = l->Nz, xm = l->xm, ym = l->ym, gxm = xm + 2;
  const int ntop = l->ntop, nside = l->nside;
  for (int j = 0; j < ym; ++j) {
    for (int i = 0; i < xm; ++i) {
      for (int k = 0; k < Nz; ++k) {
        const size_t c3 = (size_t)k + (size_t)Nz * ((size_t)i + (size_t)xm * j);
        if (l1d[k]) { /* :417-427 */
          for (int idst = 0; idst < ntop; ++idst) {
            if (l->top_inward[idst]) { /* edn */
              xb[GIDX(D, L, gxm, idst, k + 1, i, j)] -= xx[GIDX(D, L, gxm, idst, k, i, j)] * a11[c3];
              xb[GIDX(D, L, gxm, idst, k + 1, i, j)] -=
                  xx[GIDX(D, L, gxm, inv_dof(l, idst), k + 1, i, j)] * a12[c3];
            } else { /* eup */
              xb[GIDX(D, L, gxm, idst, k, i, j)] -= xx[GIDX(D, L, gxm, idst, k + 1, i, j)] * a11[c3];
              xb[GIDX(D, L, gxm, idst, k, i, j)] -= xx[GIDX(D, L, gxm, inv_dof(l, idst), k, i, j)] * a12[c3];
            }
          }
        } else { /* :429-497 */
          const double *v = diff2diff + (size_t)D * D * c3; /* v(src,dst) = v[dst*D+src] */
          int dst = 0;
          for (int part = 0; part < 3; ++part) { /* top dsts, x-side dsts, y-side dsts */
            const int nd = part == 0 ? ntop : nside;
            for (int idst = 0; idst < nd; ++idst, ++dst) {
              size_t ob;
              if (part == 0) {
                int mdst = l->top_inward[idst] ? k + 1 : k;
                ob = GIDX(D, L, gxm, dst, mdst, i, j);
              } else if (part == 1) {
                int mdst = l->side_inward[idst] ? i + 1 : i;
                ob = GIDX(D, L, gxm, dst, k, mdst, j);
              } else {
                int mdst = l->side_inward[idst] ? j + 1 : j;
                ob = GIDX(D, L, gxm, dst, k, i, mdst);
              }
              int src = 0;
              for (int isrc = 0; isrc < ntop; ++isrc, ++src) {
                int msrc = l->top_inward[isrc] ? k : k + 1;
                xb[ob] -= xx[GIDX(D, L, gxm, src, msrc, i, j)] * v[dst * D + src];
              }
              for (int isrc = 0; isrc < nside; ++isrc, ++src) {
                int msrc = l->side_inward[isrc] ? i : i + 1;
                xb[ob] -= xx[GIDX(D, L, gxm, src, k, msrc, j)] * v[dst * D + src];
              }
              for (int isrc = 0; isrc < nside; ++isrc, ++src) {
                int msrc = l->side_inward[isrc] ? j : j + 1;
                xb[ob] -= xx[GIDX(D, L, gxm, src, k, i, msrc)] * v[dst * D + src];
              }
            }
          }
        }
      }
      /* Albedo: :500-505 */
      for (int idst = 0; idst < ntop; ++idst) {
        if (!l->top_inward[idst]) {
          xb[GIDX(D, L, gxm, idst, Nz, i, j)] -=
              xx[GIDX(D, L, gxm, inv_dof(l, idst), Nz, i, j)] * albedo[i + (size_t)xm * j];
        }
      }
    }
  }
}

/* halo_fill_5pt with every neighbour == self (periodic single rank): pprts_base.F90:1622-1671.
 * x-faces are sent including the y-ghost rows (message shape (dof, zm, gym)). */
void orc_halo_fill_1rank(int D, int L, int xm, int ym, double *v) {
  const int gxm = xm + 2;
  const size_t col = (size_t)D * L;
  /* y faces first or x first does not matter for owned->ghost copies of owned data; the reference
   * posts all four at once from owned data, so corner ghosts receive (stale) ghost rows of x-faces. */
  for (int i = 0; i < xm; ++i) {
    memcpy(v + GIDX(D, L, gxm, 0, 0, i, -1), v + GIDX(D, L, gxm, 0, 0, i, ym - 1), col * sizeof(double));
    memcpy(v + GIDX(D, L, gxm, 0, 0, i, ym), v + GIDX(D, L, gxm, 0, 0, i, 0), col * sizeof(double));
  }
  for (int j = 0; j < ym; ++j) {
    memcpy(v + GIDX(D, L, gxm, 0, 0, -1, j), v + GIDX(D, L, gxm, 0, 0, xm - 1, j), col * sizeof(double));
    memcpy(v + GIDX(D, L, gxm, 0, 0, xm, j), v + GIDX(D, L, gxm, 0, 0, 0, j), col * sizeof(double));
  }
}

/* halo_reduce_5pt (ghost -> owner ADD, then ghosts zeroed): pprts_base.F90:1676-1731 */
void orc_halo_reduce_1rank(int D, int L, int xm, int ym, double *v) {
  const int gxm = xm + 2;
  const size_t col = (size_t)D * L;
  for (int j = 0; j < ym; ++j) {
    double *gw = v + GIDX(D, L, gxm, 0, 0, -1, j), *ge = v + GIDX(D, L, gxm, 0, 0, xm, j);
    double *ow = v + GIDX(D, L, gxm, 0, 0, 0, j), *oe = v + GIDX(D, L, gxm, 0, 0, xm - 1, j);
    for (size_t q = 0; q < col; ++q) {
      oe[q] += gw[q]; /* my west ghost belongs to the west neighbour's east column */
      ow[q] += ge[q];
      gw[q] = 0;
      ge[q] = 0;
    }
  }
  for (int i = 0; i < xm; ++i) {
    double *gs = v + GIDX(D, L, gxm, 0, 0, i, -1), *gn = v + GIDX(D, L, gxm, 0, 0, i, ym);
    double *os = v + GIDX(D, L, gxm, 0, 0, i, 0), *on = v + GIDX(D, L, gxm, 0, 0, i, ym - 1);
    for (size_t q = 0; q < col; ++q) {
      on[q] += gs[q];
      os[q] += gn[q];
      gs[q] = 0;
      gn[q] = 0;
    }
  }
}

void orc_owned_to_ghosted(int D, int L, int xm, int ym, const double *v, double *vg) {
  const int gxm = xm + 2, gym = ym + 2;
  memset(vg, 0, sizeof(double) * (size_t)D * L * gxm * gym);
  for (int j = 0; j < ym; ++j)
    memcpy(vg + GIDX(D, L, gxm, 0, 0, 0, j), v + OIDX(D, L, xm, 0, 0, 0, j), sizeof(double) * (size_t)D * L * xm);
}
void orc_ghosted_to_owned(int D, int L, int xm, int ym, const double *vg, double *v) {
  const int gxm = xm + 2;
  for (int j = 0; j < ym; ++j)
    memcpy(v + OIDX(D, L, xm, 0, 0, 0, j), vg + GIDX(D, L, gxm, 0, 0, 0, j), sizeof(double) * (size_t)D * L * xm);
}

/* op_mat_mult_ediff end to end on one periodic rank: pprts_shell.F90:402-519
 * (GlobalToLocal(x) -> cell loop -> LocalToGlobal(ADD) -> y += x) */
void orc_diff_apply_1rank(const orc_layout *l, const double *diff2diff, const uint8_t *l1d,
                          const double *a11, const double *a12, const double *albedo,
                          const double *x, double *y) {
  const int D = orc_D(l), L = l->Nz + 1, xm = l->xm, ym = l->ym;
  const size_t ng = (size_t)D * L * (xm + 2) * (ym + 2), n = (size_t)D * L * xm * ym;
  double *lx = (double *)malloc(ng * sizeof(double));
  double *lb = (double *)calloc(ng, sizeof(double));
  orc_owned_to_ghosted(D, L, xm, ym, x, lx);
  orc_halo_fill_1rank(D, L, xm, ym, lx);
  orc_op_mat_mult_ediff_local(l, diff2diff, l1d, a11, a12, albedo, lx, lb);
  orc_halo_reduce_1rank(D, L, xm, ym, lb);
  orc_ghosted_to_owned(D, L, xm, ym, lb, y);
  for (size_t q = 0; q < n; ++q) y[q] += x[q]; /* VecAXPY(b, one, x) :519 */
  free(lx);
  free(lb);
}

/* ------------------------------------------------------------------------------------------ */
/* set_diff_coeff as CSR: pprts.F90:5511-5796; diagonal pprts.F90:1294-1308.
 * Global natural (DMDA) ordering row = d + D*(k + L*(i + Nx*j)); periodic wrap in i,j
 * (DM_BOUNDARY_PERIODIC pprts.F90:846).  MatSetValuesStencil INSERT_VALUES semantics. */
typedef struct {
  int32_t col;
  double val;
} ent_t;
static int ent_cmp(const void *a, const void *b) {
  int32_t ca = ((const ent_t *)a)->col, cb = ((const ent_t *)b)->col;
  return (ca > cb) - (ca < cb);
}

int orc_diff_assemble_csr_1rank(const orc_layout *l, const double *diff2diff, const uint8_t *l1d,
                                const double *a11, const double *a12, const double *albedo,
                                orc_csr *A) {
  const int D = orc_D(l), L = l->Nz + 1, Nz = l->Nz, xm = l->xm, ym = l->ym;
  const int ntop = l->ntop, nside = l->nside;
  const int64_t n = (int64_t)D * L * xm * ym;
  if (n > INT32_MAX) return 1;
  /* every row: diagonal + at most max(D, ntop) entries from the one cell it leaves (+ albedo) */
  const int maxrow = 1 + (D > ntop ? D : ntop) + ntop;
  ent_t *ents = (ent_t *)malloc(sizeof(ent_t) * (size_t)n * maxrow);
  int *cnt = (int *)calloc((size_t)n, sizeof(int));
  if (!ents || !cnt) return 2;
#define ROW(d, k, i, j) ((int64_t)(d) + (int64_t)D * ((k) + (int64_t)L * (((i) + xm) % xm + (int64_t)xm * (((j) + ym) % ym))))
#define PUT(r, c, v)                                                           \
  do {                                                                         \
    int64_t r_ = (r);                                                          \
    int32_t c_ = (int32_t)(c);                                                 \
    int f_ = 0;                                                                \
    for (int q_ = 0; q_ < cnt[r_]; ++q_)                                       \
      if (ents[r_ * maxrow + q_].col == c_) {                                  \
        ents[r_ * maxrow + q_].val = (v);                                      \
        f_ = 1;                                                                \
        break;                                                                 \
      }                                                                        \
    if (!f_) {                                                                 \
      if (cnt[r_] >= maxrow) return 3;                                         \
      ents[r_ * maxrow + cnt[r_]].col = c_;                                    \
      ents[r_ * maxrow + cnt[r_]].val = (v);                                   \
      cnt[r_]++;                                                               \
    }                                                                          \
  } while (0)

  for (int64_t r = 0; r < n; ++r) PUT(r, r, 1.0); /* mat_set_diagonal :1294-1308 */

  int64_t row[ORC_MAXDOF], col[ORC_MAXDOF];
  for (int j = 0; j < ym; ++j)
    for (int i = 0; i < xm; ++i)
      for (int k = 0; k < Nz; ++k) {
        const size_t c3 = (size_t)k + (size_t)Nz * ((size_t)i + (size_t)xm * j);
        if (l1d[k]) { /* set_eddington_coeff :5675-5738 */
          int colk[ORC_MAXDOF], rowk[ORC_MAXDOF];
          for (int s = 0; s < ntop; ++s) {
            colk[s] = l->top_inward[s] ? k : k + 1;
            col[s] = ROW(s, colk[s], i, j);
          }
          for (int d = 0; d < ntop; ++d) {
            rowk[d] = l->top_inward[d] ? k + 1 : k;
            row[d] = ROW(d, rowk[d], i, j);
          }
          for (int d = 0; d < ntop; ++d)
            for (int s = 0; s < ntop; ++s) {
              double v = 0; /* v(:) = zero; explicit zeros are inserted like the reference does */
              if (colk[s] == rowk[d]) {
                if (s == inv_dof(l, d)) v = a12[c3];
              } else {
                if (s == d) v = a11[c3];
              }
              PUT(row[d], col[s], -v);
            }
        } else { /* set_pprts_coeff :5546-5651 */
          int s = 0;
          for (int q = 0; q < ntop; ++q, ++s) col[s] = ROW(s, l->top_inward[q] ? k : k + 1, i, j);
          for (int q = 0; q < nside; ++q, ++s) col[s] = ROW(s, k, l->side_inward[q] ? i : i + 1, j);
          for (int q = 0; q < nside; ++q, ++s) col[s] = ROW(s, k, i, l->side_inward[q] ? j : j + 1);
          int d = 0;
          for (int q = 0; q < ntop; ++q, ++d) row[d] = ROW(d, l->top_inward[q] ? k + 1 : k, i, j);
          for (int q = 0; q < nside; ++q, ++d) row[d] = ROW(d, k, l->side_inward[q] ? i + 1 : i, j);
          for (int q = 0; q < nside; ++q, ++d) row[d] = ROW(d, k, i, l->side_inward[q] ? j + 1 : j);
          const double *v = diff2diff + (size_t)D * D * c3;
          for (d = 0; d < D; ++d)
            for (s = 0; s < D; ++s) PUT(row[d], col[s], -v[d * D + s]); /* row-major [dst][src] :5650 */
        }
      }
  /* set_albedo_coeff :5755-5794: every (dst not inward, src inward) pair, -albedo/streams */
  const double streams = (double)(ntop / 2);
  for (int j = 0; j < ym; ++j)
    for (int i = 0; i < xm; ++i)
      for (int d = 0; d < ntop; ++d)
        if (!l->top_inward[d])
          for (int s = 0; s < ntop; ++s)
            if (l->top_inward[s]) PUT(ROW(d, Nz, i, j), ROW(s, Nz, i, j), -albedo[i + (size_t)xm * j] / streams);
#undef PUT
#undef ROW
  int64_t nnz = 0;
  for (int64_t r = 0; r < n; ++r) nnz += cnt[r];
  A->n = n;
  A->nnz = nnz;
  A->rowptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
  A->col = (int32_t *)malloc(sizeof(int32_t) * (size_t)nnz);
  A->val = (double *)malloc(sizeof(double) * (size_t)nnz);
  if (!A->rowptr || !A->col || !A->val) return 2;
  int64_t p = 0;
  for (int64_t r = 0; r < n; ++r) {
    A->rowptr[r] = p;
    qsort(ents + r * maxrow, (size_t)cnt[r], sizeof(ent_t), ent_cmp);
    for (int q = 0; q < cnt[r]; ++q, ++p) {
      A->col[p] = ents[r * maxrow + q].col;
      A->val[p] = ents[r * maxrow + q].val;
    }
  }
  A->rowptr[n] = p;
  free(ents);
  free(cnt);
  return 0;
}

void orc_csr_free(orc_csr *A) {
  free(A->rowptr);
  free(A->col);
  free(A->val);
  memset(A, 0, sizeof(*A));
}

void orc_csr_matvec(const orc_csr *A, const double *x, double *y) {
  for (int64_t r = 0; r < A->n; ++r) {
    double s = 0;
    for (int64_t p = A->rowptr[r]; p < A->rowptr[r + 1]; ++p) s += A->val[p] * x[A->col[p]];
    y[r] = s;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* ILU(0), natural ordering, on the pattern of A: the PETSc default that pprts.F90:4350-4371
 * selects for one rank (PCILU) and per block under PCBJACOBI (levels 0, fill 1, :4415-4425).
 * PETSc itself is not in the reference tree; this is the textbook IKJ ILU(0) (Saad, Alg. 10.4). */
int orc_ilu0_factor(const orc_csr *A, orc_ilu0 *F) {
  const int64_t n = A->n;
  F->A = A;
  F->lu = (double *)malloc(sizeof(double) * (size_t)A->nnz);
  F->diag = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
  int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
  if (!F->lu || !F->diag || !pos) return 2;
  memcpy(F->lu, A->val, sizeof(double) * (size_t)A->nnz);
  for (int64_t r = 0; r < n; ++r) pos[r] = -1;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t b = A->rowptr[i], e = A->rowptr[i + 1];
    F->diag[i] = -1;
    for (int64_t p = b; p < e; ++p) {
      pos[A->col[p]] = p;
      if (A->col[p] == i) F->diag[i] = p;
    }
    if (F->diag[i] < 0) return 4;
    for (int64_t p = b; p < e && A->col[p] < i; ++p) {
      const int64_t k = A->col[p];
      const double lik = F->lu[p] / F->lu[F->diag[k]];
      F->lu[p] = lik;
      for (int64_t q = F->diag[k] + 1; q < A->rowptr[k + 1]; ++q) {
        const int64_t pj = pos[A->col[q]];
        if (pj >= 0) F->lu[pj] -= lik * F->lu[q];
      }
    }
    if (F->lu[F->diag[i]] == 0.0) return 5;
    for (int64_t p = b; p < e; ++p) pos[A->col[p]] = -1;
  }
  free(pos);
  return 0;
}

void orc_ilu0_solve(const orc_ilu0 *F, const double *b, double *x) {
  const orc_csr *A = F->A;
  const int64_t n = A->n;
  for (int64_t i = 0; i < n; ++i) { /* L y = b, unit diagonal */
    double s = b[i];
    for (int64_t p = A->rowptr[i]; p < F->diag[i]; ++p) s -= F->lu[p] * x[A->col[p]];
    x[i] = s;
  }
  for (int64_t i = n - 1; i >= 0; --i) { /* U x = y */
    double s = x[i];
    for (int64_t p = F->diag[i] + 1; p < A->rowptr[i + 1]; ++p) s -= F->lu[p] * x[A->col[p]];
    x[i] = s / F->lu[F->diag[i]];
  }
}

void orc_ilu0_free(orc_ilu0 *F) {
  free(F->lu);
  free(F->diag);
  memset(F, 0, sizeof(*F));
}

/* ------------------------------------------------------------------------------------------ */
/* determine_ksp_tolerances: pprts_base.F90:1097-1142 */
void orc_determine_ksp_tolerances(int glob_xm, int glob_ym, int glob_zm, double unconstrained_fraction,
                                  double *rtol, double *atol, int *maxit) {
  const double rel_atol = 1e-4;
  *maxit = 1000;
  *rtol = 1e-5;
  *atol = rel_atol * (double)((int64_t)glob_xm * glob_ym * glob_zm) * unconstrained_fraction;
  if (*atol < 1e-8) *atol = 1e-8;
}

/* MyKSPConverged: pprts.F90:4437-4486.  n==0 stores the initial norm and continues. */
static int my_ksp_converged(int n, double rnorm, double *initial_rnorm, const orc_ksp_tol *t) {
  if (n == 0) {
    *initial_rnorm = rnorm > DBL_MIN ? rnorm : DBL_MIN;
    return 0;
  }
  if (rnorm / *initial_rnorm <= t->rtol) return 2;
  if (rnorm <= t->atol) return 3;
  if (n > t->maxit) return -3;
  if (rnorm / *initial_rnorm >= t->dtol) return -4;
  if (isnan(rnorm)) return -9;
  return 0;
}

static double vdot(int64_t n, const double *a, const double *b) {
  double s = 0;
  for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

/* KSPFBCGS (flexible = right-preconditioned BiCGStab), selected at pprts.F90:4342 with
 * KSPSetInitialGuessNonzero(TRUE) :4343.  PETSc is a third-party dependency absent from the
 * reference tree (unpinned, branch main/release: misc/build_dependencies.sh:39-40); this restates
 * the published algorithm (van der Vorst 1992; right-preconditioned form, Saad Alg. 9.? / PETSc
 * manual "KSPFBCGS") anchored on the reference's call sites and its convergence callback. */
int orc_fbcgs(int64_t n, orc_apply_fn A, void *Actx, orc_apply_fn Minv, void *Mctx,
              const double *b, double *x, const orc_ksp_tol *tol, int *niter, double *res_hist,
              int nhist) {
  double *r = (double *)malloc(sizeof(double) * (size_t)n * 8);
  if (!r) return -100;
  double *rp = r + n, *p = rp + n, *v = p + n, *s = v + n, *t = s + n, *p2 = t + n, *s2 = p2 + n;
  int reason = 0, its = 0;
  double initial = 0;

  A(Actx, x, s2); /* nonzero initial guess */
  for (int64_t i = 0; i < n; ++i) r[i] = b[i] - s2[i];
  double dp = sqrt(vdot(n, r, r));
  if (res_hist && nhist > 0) res_hist[0] = dp;
  reason = my_ksp_converged(0, dp, &initial, tol);
  if (reason) goto done;

  memcpy(rp, r, sizeof(double) * (size_t)n);
  memcpy(p, r, sizeof(double) * (size_t)n);
  double rho = vdot(n, r, rp), rhoold, alpha, omega, beta;
  if (rho == 0.0) {
    reason = -5; /* KSP_DIVERGED_BREAKDOWN */
    goto done;
  }
  for (int i = 0; i < tol->maxit; ++i) {
    if (Minv) Minv(Mctx, p, p2); else memcpy(p2, p, sizeof(double) * (size_t)n);
    A(Actx, p2, v);
    rhoold = rho;
    double d1 = vdot(n, v, rp);
    if (d1 == 0.0) {
      reason = -5;
      break;
    }
    alpha = rho / d1;
    for (int64_t q = 0; q < n; ++q) s[q] = r[q] - alpha * v[q];
    if (Minv) Minv(Mctx, s, s2); else memcpy(s2, s, sizeof(double) * (size_t)n);
    A(Actx, s2, t);
    d1 = vdot(n, s, t);
    double d2 = vdot(n, t, t);
    if (d2 == 0.0) {
      if (vdot(n, s, s) != 0.0) {
        reason = -5;
        break;
      }
      for (int64_t q = 0; q < n; ++q) x[q] += alpha * p2[q];
      its++;
      if (res_hist && its < nhist) res_hist[its] = 0.0;
      reason = 2;
      break;
    }
    omega = d1 / d2;
    for (int64_t q = 0; q < n; ++q) x[q] += alpha * p2[q] + omega * s2[q];
    for (int64_t q = 0; q < n; ++q) r[q] = s[q] - omega * t[q];
    dp = sqrt(vdot(n, r, r));
    rho = vdot(n, r, rp);
    its++;
    if (res_hist && its < nhist) res_hist[its] = dp;
    reason = my_ksp_converged(i + 1, dp, &initial, tol);
    if (reason) break;
    if (rho == 0.0) {
      reason = -5;
      break;
    }
    beta = (rho / rhoold) * (alpha / omega);
    for (int64_t q = 0; q < n; ++q) p[q] = r[q] - omega * beta * v[q] + beta * p[q];
  }
  if (!reason) reason = -3; /* KSP_DIVERGED_ITS */
done:
  if (niter) *niter = its;
  free(r);
  return reason;
}

typedef struct {
  const orc_layout *l;
  const double *c, *a11, *a12, *albedo;
  const uint8_t *l1d;
} mf_ctx;
static void mf_apply(void *ctx, const double *x, double *y) {
  mf_ctx *m = (mf_ctx *)ctx;
  orc_diff_apply_1rank(m->l, m->c, m->l1d, m->a11, m->a12, m->albedo, x, y);
}
static void csr_apply(void *ctx, const double *x, double *y) { orc_csr_matvec((const orc_csr *)ctx, x, y); }
static void ilu_apply(void *ctx, const double *x, double *y) { orc_ilu0_solve((const orc_ilu0 *)ctx, x, y); }

int orc_diff_solve_matfree(const orc_layout *l, const double *diff2diff, const uint8_t *l1d,
                           const double *a11, const double *a12, const double *albedo,
                           const double *b, double *x, const orc_ksp_tol *tol, int *niter,
                           double *res_hist, int nhist) {
  mf_ctx m = {l, diff2diff, a11, a12, albedo, l1d};
  const int64_t n = (int64_t)orc_D(l) * (l->Nz + 1) * l->xm * l->ym;
  return orc_fbcgs(n, mf_apply, &m, NULL, NULL, b, x, tol, niter, res_hist, nhist);
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* the reference's default diffuse path on one rank: set_diff_coeff -> KSPFBCGS + PCILU
 * (pprts.F90:2950-2953, 3009-3013, 4350-4360) */
int orc_diff_solve_ilu(const orc_layout *l, const double *diff2diff, const uint8_t *l1d,
                       const double *a11, const double *a12, const double *albedo, const double *b,
                       double *x, const orc_ksp_tol *tol, int *niter, double *res_hist, int nhist,
                       double *t_assemble, double *t_factor, double *t_solve) {
  orc_csr A;
  orc_ilu0 F;
  double t0 = now_s();
  int rc = orc_diff_assemble_csr_1rank(l, diff2diff, l1d, a11, a12, albedo, &A);
  if (rc) return -100 - rc;
  double t1 = now_s();
  rc = orc_ilu0_factor(&A, &F);
  if (rc) return -200 - rc;
  double t2 = now_s();
  int reason = orc_fbcgs(A.n, csr_apply, &A, ilu_apply, &F, b, x, tol, niter, res_hist, nhist);
  double t3 = now_s();
  if (t_assemble) *t_assemble = t1 - t0;
  if (t_factor) *t_factor = t2 - t1;
  if (t_solve) *t_solve = t3 - t2;
  orc_ilu0_free(&F);
  orc_csr_free(&A);
  return reason;
}

/* ------------------------------------------------------------------------------------------ */
/* explicit_ediff_sor_sweep: pprts_explicit.F90:849-1015 on ghosted arrays                      */
static void sor_sweep(const orc_layout *l, const double *coeffs, const uint8_t *l1d, const double *a11,
                      const double *a12, const double *albedo, const int dx[3], const int dy[3],
                      const int dz[3], double omega, const double *xb, double *x0) {
  const int D = orc_D(l), L = l->Nz + 1, Nz = l->Nz, xm = l->xm, gxm = xm + 2;
  const int ntop = l->ntop, nside = l->nside;
  if (dz[2] < 0) { /* :871-884 */
    for (int j = dy[0]; j != dy[1] + dy[2]; j += dy[2])
      for (int i = dx[0]; i != dx[1] + dx[2]; i += dx[2])
        for (int idst = 0; idst < ntop; ++idst)
          if (!l->top_inward[idst])
            x0[GIDX(D, L, gxm, idst, Nz, i, j)] = xb[GIDX(D, L, gxm, idst, Nz, i, j)] +
                                                  x0[GIDX(D, L, gxm, inv_dof(l, idst), Nz, i, j)] * albedo[i + (size_t)xm * j];
  }
  for (int j = dy[0]; j != dy[1] + dy[2]; j += dy[2])
    for (int i = dx[0]; i != dx[1] + dx[2]; i += dx[2])
      for (int k = dz[0]; k != dz[1] + dz[2]; k += dz[2]) {
        const size_t c3 = (size_t)k + (size_t)Nz * ((size_t)i + (size_t)xm * j);
        if (l1d[k]) { /* :890-903 */
          for (int idst = 0; idst < ntop; ++idst) {
            if (l->top_inward[idst]) {
              x0[GIDX(D, L, gxm, idst, k + 1, i, j)] = xb[GIDX(D, L, gxm, idst, k + 1, i, j)] +
                                                       x0[GIDX(D, L, gxm, idst, k, i, j)] * a11[c3] +
                                                       x0[GIDX(D, L, gxm, inv_dof(l, idst), k + 1, i, j)] * a12[c3];
            } else {
              x0[GIDX(D, L, gxm, idst, k, i, j)] = xb[GIDX(D, L, gxm, idst, k, i, j)] +
                                                   x0[GIDX(D, L, gxm, idst, k + 1, i, j)] * a11[c3] +
                                                   x0[GIDX(D, L, gxm, inv_dof(l, idst), k, i, j)] * a12[c3];
            }
          }
        } else { /* :905-978 */
          const double *v = coeffs + (size_t)D * D * c3;
          int dst = 0;
          for (int part = 0; part < 3; ++part) {
            const int nd = part == 0 ? ntop : nside;
            for (int idst = 0; idst < nd; ++idst, ++dst) {
              size_t ob;
              if (part == 0)
                ob = GIDX(D, L, gxm, dst, l->top_inward[idst] ? k + 1 : k, i, j);
              else if (part == 1)
                ob = GIDX(D, L, gxm, dst, k, l->side_inward[idst] ? i + 1 : i, j);
              else
                ob = GIDX(D, L, gxm, dst, k, i, l->side_inward[idst] ? j + 1 : j);
              double sigma = 0;
              int src = 0;
              for (int isrc = 0; isrc < ntop; ++isrc, ++src)
                sigma += x0[GIDX(D, L, gxm, src, l->top_inward[isrc] ? k : k + 1, i, j)] * v[dst * D + src];
              for (int isrc = 0; isrc < nside; ++isrc, ++src)
                sigma += x0[GIDX(D, L, gxm, src, k, l->side_inward[isrc] ? i : i + 1, j)] * v[dst * D + src];
              for (int isrc = 0; isrc < nside; ++isrc, ++src)
                sigma += x0[GIDX(D, L, gxm, src, k, i, l->side_inward[isrc] ? j : j + 1)] * v[dst * D + src];
              x0[ob] = (1.0 - omega) * x0[ob] + omega * (xb[ob] + sigma);
            }
          }
        }
      }
  if (dz[2] > 0) { /* :983-994 */
    for (int j = dy[0]; j != dy[1] + dy[2]; j += dy[2])
      for (int i = dx[0]; i != dx[1] + dx[2]; i += dx[2])
        for (int idst = 0; idst < ntop; ++idst)
          if (!l->top_inward[idst])
            x0[GIDX(D, L, gxm, idst, Nz, i, j)] = xb[GIDX(D, L, gxm, idst, Nz, i, j)] +
                                                  x0[GIDX(D, L, gxm, inv_dof(l, idst), Nz, i, j)] * albedo[i + (size_t)xm * j];
  }
}

/* fill_ghost with self neighbours: pprts_explicit.F90:1019-1073 (x faces (dof,zm,ym), y faces (dof,zm,xm)) */
static void fill_ghost_1rank(int D, int L, int xm, int ym, double *v) {
  const int gxm = xm + 2;
  const size_t col = (size_t)D * L;
  for (int j = 0; j < ym; ++j) {
    memcpy(v + GIDX(D, L, gxm, 0, 0, -1, j), v + GIDX(D, L, gxm, 0, 0, xm - 1, j), col * sizeof(double));
    memcpy(v + GIDX(D, L, gxm, 0, 0, xm, j), v + GIDX(D, L, gxm, 0, 0, 0, j), col * sizeof(double));
  }
  for (int i = 0; i < xm; ++i) {
    memcpy(v + GIDX(D, L, gxm, 0, 0, i, -1), v + GIDX(D, L, gxm, 0, 0, i, ym - 1), col * sizeof(double));
    memcpy(v + GIDX(D, L, gxm, 0, 0, i, ym), v + GIDX(D, L, gxm, 0, 0, i, 0), col * sizeof(double));
  }
}

/* exchange_diffuse_boundary with self neighbours: pprts_explicit.F90:715-848 */
static void exchange_diffuse_boundary_1rank(const orc_layout *l, double *x0) {
  const int D = orc_D(l), L = l->Nz + 1, xm = l->xm, ym = l->ym, gxm = xm + 2;
  const int ntop = l->ntop, nside = l->nside;
  const int nh = nside / 2;
  double *se = (double *)malloc(sizeof(double) * (size_t)nh * L * ym * 2);
  double *sw = se + (size_t)nh * L * ym;
  double *sn = (double *)malloc(sizeof(double) * (size_t)nh * L * xm * 2);
  double *ss = sn + (size_t)nh * L * xm;
  for (int j = 0; j < ym; ++j)
    for (int k = 0; k < L; ++k) {
      int d1 = 0, d2 = 0;
      for (int idof = 0; idof < nside; ++idof) {
        int dof = ntop + idof;
        if (l->side_inward[idof]) se[(d1++) + nh * (k + (size_t)L * j)] = x0[GIDX(D, L, gxm, dof, k, xm, j)];
        else sw[(d2++) + nh * (k + (size_t)L * j)] = x0[GIDX(D, L, gxm, dof, k, 0, j)];
      }
    }
  for (int i = 0; i < xm; ++i)
    for (int k = 0; k < L; ++k) {
      int d1 = 0, d2 = 0;
      for (int idof = 0; idof < nside; ++idof) {
        int dof = ntop + nside + idof;
        if (l->side_inward[idof]) sn[(d1++) + nh * (k + (size_t)L * i)] = x0[GIDX(D, L, gxm, dof, k, i, ym)];
        else ss[(d2++) + nh * (k + (size_t)L * i)] = x0[GIDX(D, L, gxm, dof, k, i, 0)];
      }
    }
  /* self neighbours: recv_w == send_e (tag_e), recv_e == send_w, recv_s == send_n, recv_n == send_s */
  for (int j = 0; j < ym; ++j)
    for (int k = 0; k < L; ++k) {
      int d1 = 0, d2 = 0;
      for (int idof = 0; idof < nside; ++idof) {
        int dof = ntop + idof;
        if (l->side_inward[idof]) x0[GIDX(D, L, gxm, dof, k, 0, j)] = se[(d1++) + nh * (k + (size_t)L * j)];
        else x0[GIDX(D, L, gxm, dof, k, xm, j)] = sw[(d2++) + nh * (k + (size_t)L * j)];
      }
    }
  for (int i = 0; i < xm; ++i)
    for (int k = 0; k < L; ++k) {
      int d1 = 0, d2 = 0;
      for (int idof = 0; idof < nside; ++idof) {
        int dof = ntop + nside + idof;
        if (l->side_inward[idof]) x0[GIDX(D, L, gxm, dof, k, i, 0)] = sn[(d1++) + nh * (k + (size_t)L * i)];
        else x0[GIDX(D, L, gxm, dof, k, i, ym)] = ss[(d2++) + nh * (k + (size_t)L * i)];
      }
    }
  free(se);
  free(sn);
}

/* explicit_ediff: pprts_explicit.F90:461-713 (pc_sub_it = 1, no option overrides) */
int orc_explicit_ediff_1rank(const orc_layout *l, const double *diff2diff, const uint8_t *l1d,
                             const double *a11, const double *a12, const double *albedo,
                             const double *b, double *vediff, const orc_sor_opts *o, int *niter,
                             double *res_hist, int nhist) {
  const int D = orc_D(l), L = l->Nz + 1, Nz = l->Nz, xm = l->xm, ym = l->ym, gxm = xm + 2;
  const size_t ng = (size_t)D * L * gxm * (ym + 2);
  double *v0 = (double *)malloc(ng * sizeof(double));
  double *lvb = (double *)malloc(ng * sizeof(double));
  double *residual = (double *)calloc((size_t)o->maxit + 3, sizeof(double));
  orc_owned_to_ghosted(D, L, xm, ym, vediff, v0);
  fill_ghost_1rank(D, L, xm, ym, v0);
  orc_owned_to_ghosted(D, L, xm, ym, b, lvb);
  fill_ghost_1rank(D, L, xm, ym, lvb);

  double omega_adaptive = o->omega, omega_min = 1.0, omega_max = 1.25, omega_increment = 0.1;
  double omega_dir = 1.0, omega_step = omega_increment * 0.5, log_rate_prev = 0.0;
  double best_residual = DBL_MAX;
  int iter_at_best = 1, frozen = 0, converged = 0, iter;
  const int stagnation_window = 50;
  const int fdx[3] = {0, xm - 1, 1}, fdy[3] = {0, ym - 1, 1}, fdz[3] = {0, Nz - 1, 1};
  const int bdx[3] = {xm - 1, 0, -1}, bdy[3] = {ym - 1, 0, -1}, bdz[3] = {Nz - 1, 0, -1};

  for (iter = 1; iter <= o->maxit; ++iter) {
    if ((iter + 1) % 2 == 0) /* modulo(iter + isub, 2) == 0 with isub = 1: :591 */
      sor_sweep(l, diff2diff, l1d, a11, a12, albedo, fdx, fdy, fdz, omega_adaptive, lvb, v0);
    else
      sor_sweep(l, diff2diff, l1d, a11, a12, albedo, bdx, bdy, bdz, omega_adaptive, lvb, v0);
    exchange_diffuse_boundary_1rank(l, v0);

    double s = 0; /* residual = norm2(xg - x0(owned)); xg = x0(owned)  :621-626 */
    for (int j = 0; j < ym; ++j)
      for (int i = 0; i < xm; ++i)
        for (int q = 0; q < D * L; ++q) {
          const size_t og = OIDX(D, L, xm, 0, 0, i, j) + q, gg = GIDX(D, L, gxm, 0, 0, i, j) + q;
          const double d = vediff[og] - v0[gg];
          s += d * d;
          vediff[og] = v0[gg];
        }
    residual[iter] = sqrt(s);
    double rel = residual[1] <= sqrt(DBL_MIN) ? 0.0 : residual[iter] / residual[1];
    if (res_hist) res_hist[(iter < nhist ? iter : nhist) - 1] = residual[iter];
    if (residual[iter] < o->atol || rel < o->rtol) {
      converged = 1;
      break;
    }
    if (residual[iter] < best_residual) {
      best_residual = residual[iter];
      iter_at_best = iter;
    }
    if (o->adaptive_omega && iter >= 3) { /* :662-686 */
      if (!frozen && omega_adaptive > omega_min && (iter - iter_at_best) > stagnation_window) {
        omega_adaptive = omega_min;
        frozen = 1;
      }
      if (!frozen && residual[iter] > 0 && residual[iter - 2] > 0) {
        double log_rate = 0.5 * log(residual[iter] / residual[iter - 2]);
        if (log_rate < log_rate_prev) {
          omega_step = fmin(omega_step * 1.3, omega_max - omega_min);
        } else {
          omega_dir = -omega_dir;
          omega_step = fmax(omega_step * 0.5, 0.01);
        }
        log_rate_prev = log_rate;
        omega_adaptive = fmin(fmax(omega_adaptive + omega_dir * omega_step, omega_min), omega_max);
      }
    }
  }
  if (niter) *niter = iter <= o->maxit ? iter : o->maxit;
  orc_ghosted_to_owned(D, L, xm, ym, v0, vediff);
  free(v0);
  free(lvb);
  free(residual);
  return converged ? 0 : 1;
}
