// tsx_pc.hip -- host side of the column-block preconditioner (kernels: tsx_kernels_pc.hpp).  Its own translation unit so
// that the sweep kernels compile next to the operator's.
#include "tsx_host.hpp"
#include "tsx_kernels_pc.hpp"

// ------------------------------------------------------------------------------------------------
// z = M^-1 v with the column preconditioner; sweeps > 1 adds stationary refinement sweeps
//   z <- z + M^-1 (v - A z)     (block-Jacobi iteration on column blocks)
// one pass of the column preconditioner: writes rows of colour ROWS into z; GS: +-y coupling from zy (other colour);
// XL: lagged +-x coupling from zx (own colour, previous pass, a different buffer than z)
template <int NTOP, int NSIDE, int ROWS, bool GS, bool XL, typename ZT>
static int pc_column_launch(tsx_solver *s, const double *v, ZT *z, const ZT *zy, const ZT *zx, const int *done) {
  const TsxGeo &g = s->geo;
  const int ncols = ROWS == 0 ? g.ncol : (ROWS == 1 ? (g.ym + 1) / 2 : g.ym / 2) * g.xm;
  if (ncols == 0) return TSX_OK;
  const int nb = (ncols + 63) / 64;
  if constexpr (std::is_same<ZT, float>::value) {
    // fp32 directions: always the packed reduced-precision blocks (tsx_pc_ensure_half) and the fp32 right-hand side s->v32
    if (!s->pc_half || !s->pc_rhs) {
      tsx_set_error("preconditioner: fp32 directions need the packed blocks (internal state error)");
      return TSX_ERR_STATE;
    }
    if constexpr (NTOP == 2) {
      // sweep temporaries in LDS when a block's share (Nz x 64 columns x 16 B) fits the per-block limit
      const size_t lds = (size_t)g.Nz * 64 * sizeof(float4);
      static int use_lds = -1;  // TSX_PC_LDS=0: keep them in global memory (A/B knob)
      if (use_lds < 0) {
        const char *e = getenv("TSX_PC_LDS");
        use_lds = e ? atoi(e) : 1;
      }
#define TSX_P16_LAUNCH(HAS, LDST, BYTES)                                                                                        \
  hipLaunchKernelGGL((tsx_k_pc_column_p16<ROWS, GS, HAS, XL, LDST>), dim3(nb), dim3(64), BYTES, s->stream, g,                    \
                     (const uint4 *)s->coef_h, s->l1d, s->a11, s->a12, s->albedo, (const float *)s->pc_rhs, z, zy, zx,             \
                     (float4 *)s->pc_tmp, done)
      if (use_lds && lds <= (size_t)s->max_lds) {
        if (s->any_l1d) TSX_P16_LAUNCH(true, true, lds);
        else TSX_P16_LAUNCH(false, true, lds);
      } else {
        if (s->any_l1d) TSX_P16_LAUNCH(true, false, 0);
        else TSX_P16_LAUNCH(false, false, 0);
      }
#undef TSX_P16_LAUNCH
    } else {
      if (s->any_l1d)
        hipLaunchKernelGGL((tsx_k_pc_column_p16h<ROWS, GS, true, XL>), dim3(nb), dim3(64), 0, s->stream, g,
                           (const uint4 *)s->coef_h, s->l1d, s->a11, s->a12, s->albedo, (const float *)s->pc_rhs, z, zy, zx,
                           (float4 *)s->pc_tmp, done);
      else
        hipLaunchKernelGGL((tsx_k_pc_column_p16h<ROWS, GS, false, XL>), dim3(nb), dim3(64), 0, s->stream, g,
                           (const uint4 *)s->coef_h, s->l1d, s->a11, s->a12, s->albedo, (const float *)s->pc_rhs, z, zy, zx,
                           (float4 *)s->pc_tmp, done);
    }
    HIPCHK(hipGetLastError());
    return TSX_OK;
  } else {
    // fp64 directions: the exact blocks
    static int use_h1 = -1;  // TSX_PC_PREFETCH=0 selects the generic kernel for 3_10 as well (A/B knob)
    if (use_h1 < 0) {
      const char *e = getenv("TSX_PC_PREFETCH");
      use_h1 = e ? atoi(e) : 1;
    }
    if constexpr (NTOP == 2) {
      if (use_h1) {
#define TSX_H1_LAUNCH(CTYPE, HAS)                                                                                              \
  hipLaunchKernelGGL((tsx_k_pc_column_h1<CTYPE, ROWS, GS, ZT, HAS, XL>), dim3(nb), dim3(64), 0, s->stream, g,                     \
                     (const CTYPE *)s->coef, s->l1d, s->a11, s->a12, s->albedo, v, z, zy, zx, (void *)s->pc_tmp, done)
        if (s->coef_bytes == 4) {
          if (s->any_l1d) TSX_H1_LAUNCH(float, true);
          else TSX_H1_LAUNCH(float, false);
        } else {
          if (s->any_l1d) TSX_H1_LAUNCH(double, true);
          else TSX_H1_LAUNCH(double, false);
        }
#undef TSX_H1_LAUNCH
        HIPCHK(hipGetLastError());
        return TSX_OK;
      }
    }
    // generic kernel (8_16, or A/B): y coupling only
    if (s->coef_bytes == 4)
      hipLaunchKernelGGL((tsx_k_pc_column<NTOP, NSIDE, float, ROWS, GS, ZT>), dim3(nb), dim3(64), 0, s->stream, g,
                         (const float *)s->coef, s->l1d, s->a11, s->a12, s->albedo, v, z, zy, s->pc_tmp, done);
    else
      hipLaunchKernelGGL((tsx_k_pc_column<NTOP, NSIDE, double, ROWS, GS, ZT>), dim3(nb), dim3(64), 0, s->stream, g,
                         (const double *)s->coef, s->l1d, s->a11, s->a12, s->albedo, v, z, zy, s->pc_tmp, done);
    HIPCHK(hipGetLastError());
    return TSX_OK;
  }
}

// z = M^-1 v.
//  TSX_PC_COLUMN: block-Jacobi over columns; sweeps > 1 adds stationary refinement  z <- z + M^-1 (v - A z)  (fp64 only)
//  TSX_PC_ZEBRA:  pc_sweeps + 1 half-grid passes over the column blocks, even rows / odd rows alternately.  From the
//                 second pass on the +-y streams of the other colour (latest values) are on the right-hand side
//                 (line Gauss-Seidel in y); from the third pass on also the +-x streams of the same rows with the
//                 values of that colour's previous pass (Jacobi in x).  Each colour alternates between z and a
//                 scratch buffer so that a pass never reads what it writes; the last pass of each colour lands in z.
// ZT = float stores the preconditioned direction in fp32 (legitimate in *flexible* BiCGStab, see tsx_k_spmv_w).
template <int NTOP, int NSIDE, typename ZT>
static int apply_pc(tsx_solver *s, const double *v, ZT *z, bool in_solve) {
  const TsxGeo &g = s->geo;
  const int *done = in_solve ? &s->scal->done : nullptr;
  int rc;
  if constexpr (std::is_same<ZT, float>::value) {
    if (s->pc == TSX_PC_REDBLACK) {
      // pc_sweeps + 1 passes, colours alternately; the iterate lives colour-split in s->vw; the last pass also writes the
      // Krylov-layout result z for both colours
      if (!s->pc_half || !s->pc_rhs || !s->coef_h_split) {
        tsx_set_error("preconditioner: red-black ordering needs the colour-split packed blocks (internal state error)");
        return TSX_ERR_STATE;
      }
      if (s->coef_h_scan) return tsx_pcs_apply(s, (float *)z, done);
      const int P = s->pc_sweeps + 1;
      float *zs = (float *)s->vw;                                   // fp32 iterate (written by the last pass of colour P % 2)
      unsigned short *zb = (unsigned short *)(zs + (size_t)g.N);   // bf16 neighbour values of the intermediate passes
      const int nb = (g.ym * (g.xm / 2) + 63) / 64;
      const size_t lds = (size_t)g.Nz * 64 * sizeof(float4);
      static int use_lds = -1;
      if (use_lds < 0) {
        const char *e = getenv("TSX_PC_LDS");
        use_lds = e ? atoi(e) : 1;
      }
      const bool ld = NTOP == 2 && use_lds && lds <= (size_t)s->max_lds;
      // 3_10: pass modes 0 (intermediate: bf16 side streams only), 1 (first colour's last pass: fp32), 2 (last pass: pairs).
      // 8_16 keeps fp32 iterates throughout: its kernel knows FINAL (= mode 2) only.
#define TSX_RB_LAUNCH(GSV, HAS, LDSV, MODEV)                                                                                    \
  do {                                                                                                                          \
    if constexpr (NTOP == 2)                                                                                                    \
      hipLaunchKernelGGL((tsx_k_pc_column_rb<GSV, HAS, LDSV, MODEV>), dim3(nb), dim3(64), LDSV ? lds : 0, s->stream, g,          \
                         (const uint4 *)s->coef_h, s->l1d, s->a11, s->a12, s->albedo, (const float *)s->pc_rhs, zs, zb,            \
                         (float *)z, (float4 *)s->pc_tmp, done, pass & 1);                                                      \
    else                                                                                                                        \
      hipLaunchKernelGGL((tsx_k_pc_column_rbh<GSV, HAS, (MODEV == 2)>), dim3(nb), dim3(64), 0, s->stream, g,                     \
                         (const uint4 *)s->coef_h, s->l1d, s->a11, s->a12, s->albedo, (const float *)s->pc_rhs, zs,             \
                         (const float *)zs, (float *)z, (float4 *)s->pc_tmp, done, pass & 1);                                   \
  } while (0)
#define TSX_RB_L2(GSV, HAS, MODEV)                                                                                              \
  do {                                                                                                                          \
    if (ld) TSX_RB_LAUNCH(GSV, HAS, true, MODEV);                                                                               \
    else TSX_RB_LAUNCH(GSV, HAS, false, MODEV);                                                                                 \
  } while (0)
#define TSX_RB_L1(GSV, MODEV)                                                                                                   \
  do {                                                                                                                          \
    if (s->any_l1d) TSX_RB_L2(GSV, true, MODEV);                                                                                \
    else TSX_RB_L2(GSV, false, MODEV);                                                                                          \
  } while (0)
      for (int pass = 0; pass < P; ++pass) {
        const int mode = pass == P - 1 ? 2 : (pass == P - 2 ? 1 : 0);
        if (pass == 0) {
          if (mode == 1) TSX_RB_L1(false, 1);
          else TSX_RB_L1(false, 0);
        } else {
          if (mode == 2) TSX_RB_L1(true, 2);
          else if (mode == 1) TSX_RB_L1(true, 1);
          else TSX_RB_L1(true, 0);
        }
      }
#undef TSX_RB_L1
#undef TSX_RB_L2
#undef TSX_RB_LAUNCH
      HIPCHK(hipGetLastError());
      return TSX_OK;
    }
  }
  if constexpr (std::is_same<ZT, double>::value && NTOP == 2) {
    if (s->pc == TSX_PC_REDBLACK) return tsx_pcx_apply(s, v, z, done);  // exact blocks, fp64 iterates, scan over the levels (tsx_pcx.hip)
  }
  if (s->pc == TSX_PC_ZEBRA) {
    const int P = s->pc_sweeps + 1;
    ZT *alt = (ZT *)s->vw;
    // lagged x coupling: 3_10 kernels and the packed 8_16 kernel; the generic (exact) 8_16 kernel couples in y only
    const bool xl = g.ym >= 2 && (NTOP == 2 || std::is_same<ZT, float>::value);
    auto buf = [&](int pass) {  // buffer a pass writes: its colour's last pass writes z, alternating backwards
      const int last = ((P - 1) % 2 == pass % 2) ? P - 1 : P - 2;
      return (((last - pass) / 2) % 2 == 0 || !xl) ? z : alt;
    };
    for (int pass = 0; pass < P; ++pass) {
      ZT *out = buf(pass);
      const ZT *zy = pass > 0 ? buf(pass - 1) : (const ZT *)out;
      const ZT *zx = pass > 1 ? buf(pass - 2) : (const ZT *)out;
      if (pass == 0) rc = pc_column_launch<NTOP, NSIDE, 1, false, false, ZT>(s, v, out, zy, zx, done);
      else if (pass == 1) rc = pc_column_launch<NTOP, NSIDE, 2, true, false, ZT>(s, v, out, zy, zx, done);
      else if (!xl) rc = (pass & 1) ? pc_column_launch<NTOP, NSIDE, 2, true, false, ZT>(s, v, out, zy, zx, done)
                                    : pc_column_launch<NTOP, NSIDE, 1, true, false, ZT>(s, v, out, zy, zx, done);
      else rc = (pass & 1) ? pc_column_launch<NTOP, NSIDE, 2, true, true, ZT>(s, v, out, zy, zx, done)
                           : pc_column_launch<NTOP, NSIDE, 1, true, true, ZT>(s, v, out, zy, zx, done);
      if (rc) return rc;
    }
    return TSX_OK;
  }
  if ((rc = pc_column_launch<NTOP, NSIDE, 0, false, false, ZT>(s, v, z, (const ZT *)z, (const ZT *)z, done))) return rc;
  if constexpr (std::is_same<ZT, double>::value) {
    const long long n2 = g.N / 2;
    const int nbv = grid_for(n2);
    for (int sw = 1; sw < s->pc_sweeps; ++sw) {
      if ((rc = launch_spmv<NTOP, NSIDE, 0>(s, (const double *)z, s->vt, (const double *)nullptr, in_solve))) return rc;
      hipLaunchKernelGGL(tsx_k_sub, dim3(nbv), dim3(TSX_BLOCK), 0, s->stream, n2, (const double2 *)v, (const double2 *)s->vt,
                         (double2 *)s->vt, done);
      if ((rc = pc_column_launch<NTOP, NSIDE, 0, false, false, double>(s, s->vt, s->vw, (const double *)s->vw,
                                                                      (const double *)s->vw, done)))
        return rc;
      hipLaunchKernelGGL(tsx_k_addto, dim3(nbv), dim3(TSX_BLOCK), 0, s->stream, n2, (const double2 *)s->vw, (double2 *)z, done);
    }
  }
  HIPCHK(hipGetLastError());
  return TSX_OK;
}

template <int NTOP>
static int ensure_pc_buffers_t(tsx_solver *s) {
  const TsxGeo &g = s->geo;
  const size_t nb = (size_t)g.N * sizeof(double);
  if (!s->pc_tmp) HIPCHK(tsx_dev_malloc((void **)&s->pc_tmp, sizeof(double) * (size_t)tsx_pc_ntmp<NTOP>() * g.Nc));
  if (s->vph == s->vp || !s->vph) HIPCHK(tsx_dev_malloc((void **)&s->vph, nb));  // fp64-sized: also holds the fp32 form
  if (s->vsh == s->vs || !s->vsh) HIPCHK(tsx_dev_malloc((void **)&s->vsh, nb));
  if (!s->vw) HIPCHK(tsx_dev_malloc((void **)&s->vw, nb));
  if (!s->v32) HIPCHK(tsx_dev_malloc((void **)&s->v32, (size_t)g.N * sizeof(float)));  // fp32 right-hand sides of the mixed path
  if (!s->p32) HIPCHK(tsx_dev_malloc((void **)&s->p32, (size_t)g.N * sizeof(float)));
  if (!s->pc_rhs) s->pc_rhs = s->v32;
  return TSX_OK;
}

// packed fp16 copy of the blocks for the preconditioner (tsx_k_pack_p16), rebuilt when the coefficients changed
int tsx_pc_ensure_half(tsx_solver *s) {
  const bool h1 = s->geo.ntop == 2;
  // 3_10: 8 groups, + 1 for the second half of record 1 in fp16 (tsx_k_pcs_pack_rec1h)
  const long long n = (long long)(h1 ? TSX_P16_GROUPS + 1 : 34 /* max(TSX_P16H_GROUPS, 14 + 20 of the scan layout) */) * s->geo.Nc;
  if (!s->coef_h) HIPCHK(tsx_dev_malloc((void **)&s->coef_h, sizeof(tsx_h8) * (size_t)n));
  const bool scan = s->pc_split && tsx_pcs_eligible(s);
  if (scan && (!s->coef_h_valid || !s->coef_h_scan || s->coef_h_dd != ((s->dd_on || s->dd_pc) && s->coef_bytes == 4))) {
    int rc = tsx_pcs_pack(s);
    if (rc) return rc;
    s->coef_h_valid = true;
    s->coef_h_scan = true;
    s->coef_h_split = true;
  }
  if (!scan && (!s->coef_h_valid || s->coef_h_scan || s->coef_h_split != s->pc_split)) {
    s->coef_h_scan = false;
    const int sx = s->pc_split ? s->geo.xm : 0, sy = s->pc_split ? s->geo.ym : 0;
#define TSX_PACK(CTYPE, NT)                                                                                        \
  hipLaunchKernelGGL((tsx_k_pack_p16<CTYPE, NT>), dim3(grid_for(n)), dim3(TSX_BLOCK), 0, s->stream, s->geo.Nc,      \
                     (const CTYPE *)s->coef, (uint4 *)s->coef_h, sx, sy)
    if (s->coef_bytes == 4) {
      if (h1) TSX_PACK(float, 2);
      else TSX_PACK(float, 8);
    } else {
      if (h1) TSX_PACK(double, 2);
      else TSX_PACK(double, 8);
    }
#undef TSX_PACK
    HIPCHK(hipGetLastError());
    s->coef_h_valid = true;
    s->coef_h_split = s->pc_split;
  }
  s->pc_half = true;
  return TSX_OK;
}


int tsx_pc_ensure_buffers(tsx_solver *s) { return s->geo.ntop == 2 ? ensure_pc_buffers_t<2>(s) : ensure_pc_buffers_t<8>(s); }

int tsx_pc_apply(tsx_solver *s, const double *v, void *z, bool z_is_float, bool in_solve) {
  if (s->geo.ntop == 2)
    return z_is_float ? apply_pc<2, 4, float>(s, v, (float *)z, in_solve) : apply_pc<2, 4, double>(s, v, (double *)z, in_solve);
  return z_is_float ? apply_pc<8, 4, float>(s, v, (float *)z, in_solve) : apply_pc<8, 4, double>(s, v, (double *)z, in_solve);
}

int tsx_pc_narrow(tsx_solver *s, const double *a) {  // s->v32 = (float) a: the mixed path's right-hand side
  hipLaunchKernelGGL(tsx_k_narrow, dim3(grid_for(s->geo.N)), dim3(TSX_BLOCK), 0, s->stream, s->geo, (int)s->pc_split, a, s->v32);
  s->pc_rhs = s->v32;
  HIPCHK(hipGetLastError());
  return TSX_OK;
}

int tsx_pc_widen(tsx_solver *s, const float *a, double *o) {
  hipLaunchKernelGGL(tsx_k_widen, dim3(grid_for(s->geo.N)), dim3(TSX_BLOCK), 0, s->stream, s->geo.N, a, o);
  HIPCHK(hipGetLastError());
  return TSX_OK;
}

TSX_CODE_PROBE(pc)  // tsx_host.hpp: this unit's code object as it sits in device memory (diagnostics)
