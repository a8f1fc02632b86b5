// tsx_kernels.hpp -- Krylov BLAS-1 stages, scalar stage, layout conversion, coefficient import and LUT kernels
// (operator: tsx_kernels_spmv.hpp, preconditioner: tsx_kernels_pc.hpp, helpers: tsx_dev.hpp)
#pragma once
#include "tsx_dev.hpp"
#include "tsx_peer_dev.hpp"

// ------------------------------------------------------------------------------------------------
// BLAS-1 stages of the flexible BiCGStab (KSPFBCGS, selected at src/pprts.F90:4342), fused so that a
// full iteration moves 19 N-vectors besides the two operator applications.
// r = b - y (y = A x0); rhat = r; p = r; slot0 = (r,r)
template <typename RT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_residual0(long long n, const double *__restrict__ b,
                                                             const double *__restrict__ y, double *__restrict__ r,
                                                             RT *__restrict__ rhat, double *__restrict__ p,
                                                             float *__restrict__ p32, double *__restrict__ partials, TsxGeo g,
                                                             int split, int yzero) {
  // yzero: the initial guess is known to be zero -- y = A x0 was not computed and is not read: r = b
  double sum[2] = {0.0, 0.0};  // slot0 = (rhat, r) with rhat as stored, slot1 = (r, r)
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * TSX_BLOCK) {
    const double v = yzero ? b[q] : b[q] - y[q];
    r[q] = v;
    rhat[q] = (RT)v;
    if (p32) p32[split ? tsx_split_pos(q, g) : q] = (float)v;  // fp32 directions with a preconditioner: p lives in fp32 only
    else p[q] = v;
    sum[0] += (double)(RT)v * v;
    sum[1] += v * v;
  }
  tsx_block_reduce_store<2>(sum, partials);
}

// p = r + beta (p - omega v).  p32 (nullable): an fp32 copy for the preconditioner, which reads its right-hand side three
// times per colour -- the preconditioner is an approximation anyway, the Krylov recurrence keeps the fp64 p
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_pupdate(long long n2, const TsxScalars *__restrict__ sc,
                                                           const double2 *__restrict__ r, double2 *__restrict__ p,
                                                           const double2 *__restrict__ v, float2 *__restrict__ p32, TsxGeo g,
                                                           int split) {
  if (sc->done) return;
  const double beta = sc->beta, omega = sc->omega;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n2; q += (long long)gridDim.x * TSX_BLOCK) {
    const double2 rr = r[q], pp = p[q], vv = v[q];
    double2 o;
    o.x = rr.x + beta * (pp.x - omega * vv.x);
    o.y = rr.y + beta * (pp.y - omega * vv.y);
    p[q] = o;
    if (p32) {
      if (split) {  // colour-split order of the red-black preconditioner
        float *f = reinterpret_cast<float *>(p32);
        f[tsx_split_pos(2 * q, g)] = (float)o.x;
        f[tsx_split_pos(2 * q + 1, g)] = (float)o.y;
      } else {
        p32[q] = make_float2((float)o.x, (float)o.y);
      }
    }
  }
}

// p = r + beta (p - omega v) with p stored in fp32 only (fp32 directions + preconditioner): p enters the iteration only
// through p-hat = M^-1 p, and x, r are updated with v = A p-hat whatever p-hat is, so rounding p perturbs the choice of the
// direction, not the consistency of x and r -- the same licence flexible BiCGStab gives the preconditioner.  p32 may be in
// the colour-split order of the red-black preconditioner.
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_pupdate32(long long n2, const TsxScalars *__restrict__ sc,
                                                             const double2 *__restrict__ r, const double2 *__restrict__ v,
                                                             float *__restrict__ p32, TsxGeo g, int split) {
  if (sc->done) return;
  const double beta = sc->beta, omega = sc->omega;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n2; q += (long long)gridDim.x * TSX_BLOCK) {
    const double2 rr = r[q], vv = v[q];
    const long long i0 = split ? tsx_split_pos(2 * q, g) : 2 * q, i1 = split ? tsx_split_pos(2 * q + 1, g) : 2 * q + 1;
    const double p0 = (double)p32[i0], p1 = (double)p32[i1];
    p32[i0] = (float)(rr.x + beta * (p0 - omega * vv.x));
    p32[i1] = (float)(rr.y + beta * (p1 - omega * vv.y));
  }
}

// s = r - alpha v
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_supdate(long long n2, const TsxScalars *__restrict__ sc,
                                                           const double2 *__restrict__ r, const double2 *__restrict__ v,
                                                           double2 *__restrict__ s, float2 *__restrict__ s32, TsxGeo g,
                                                           int split, double *__restrict__ partials) {
  if (sc->done) return;
  const double alpha = sc->alpha;
  double sum[1] = {0.0};  // (s, s): the half-step stop test (TSX_STAGE_HALF)
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n2; q += (long long)gridDim.x * TSX_BLOCK) {
    const double2 rr = r[q], vv = v[q];
    double2 o;
    o.x = rr.x - alpha * vv.x;
    o.y = rr.y - alpha * vv.y;
    sum[0] += o.x * o.x + o.y * o.y;
    s[q] = o;
    if (s32) {
      if (split) {
        float *f = reinterpret_cast<float *>(s32);
        f[tsx_split_pos(2 * q, g)] = (float)o.x;
        f[tsx_split_pos(2 * q + 1, g)] = (float)o.y;
      } else {
        s32[q] = make_float2((float)o.x, (float)o.y);
      }
    }
  }
  if (partials) tsx_block_reduce_store<1>(sum, partials);
}

// x += alpha ph + omega sh; r = s - omega t; slot0 = (rhat, r), slot1 = (r, r).  PT: storage of ph/sh, RT: of rhat
template <typename PT, typename RT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_xrupdate(long long n2, const TsxScalars *__restrict__ sc,
                                                            double2 *__restrict__ x, const PT *__restrict__ ph,
                                                            const PT *__restrict__ sh, const double2 *__restrict__ s,
                                                            const double2 *__restrict__ t, const RT *__restrict__ rhat,
                                                            double2 *__restrict__ r, double *__restrict__ partials) {
  if (sc->done) return;
  const double alpha = sc->alpha, omega = sc->omega;
  double sum[2] = {0.0, 0.0};
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n2; q += (long long)gridDim.x * TSX_BLOCK) {
    const double2 xx = x[q], ss = s[q], tt = t[q];
    double pp[2], ss2[2], rh[2];
    TsxVec<2>::ld(ph + 2 * q, pp);
    TsxVec<2>::ld(sh + 2 * q, ss2);
    TsxVec<2>::ld(rhat + 2 * q, rh);
    double2 xo, ro;
    xo.x = xx.x + alpha * pp[0] + omega * ss2[0];
    xo.y = xx.y + alpha * pp[1] + omega * ss2[1];
    ro.x = ss.x - omega * tt.x;
    ro.y = ss.y - omega * tt.y;
    x[q] = xo;
    r[q] = ro;
    sum[0] += rh[0] * ro.x + rh[1] * ro.y;
    sum[1] += ro.x * ro.x + ro.y * ro.y;
  }
  tsx_block_reduce_store<2>(sum, partials);
}

// ------------------------------------------------------------------------------------------------
// fp32 Krylov vectors (tsx_ksp_opts.fp32_directions = 2, the default with a preconditioner).  The reference's `ireals` may be
// real32 or real64 (CI builds both); here the iterate x, the right-hand side, every dot product and the stop rule stay fp64, the
// operator works on the exact blocks, and the recurrence vectors r, s, v, t join p, p-hat, s-hat, r-hat in fp32: 96 instead of
// 140 bytes per unknown and iteration.  A recurrence kept in fp32 drifts from b - A x by about 1e-7 |b|, so the residual is
// REPLACED by the true one, b - A x evaluated in fp64 (van der Vorst & Ye's residual replacement: r, rho and the norm are
// renewed, the search direction is kept), whenever the recurrence has fallen four orders of magnitude since the last
// replacement and -- always -- before convergence is declared: the stop rule of MyKSPConverged (src/pprts.F90:4437-4486) is
// decided on the fp64 norm of the true residual.
// r = b - y (y = A x in fp64; yzero: x = 0, r = b) -> r32; init: also rhat32 = p32 = r.  slot0 = (rhat, r), slot1 = (r, r)
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_residual_k32(long long n, const double *__restrict__ b, const double *__restrict__ y,
                                                                float *__restrict__ r32, float *__restrict__ rhat32,
                                                                float *__restrict__ p32, double *__restrict__ partials, TsxGeo g,
                                                                int split, int yzero, int init, const int *__restrict__ done) {
  if (done && *done) return;
  double sum[2] = {0.0, 0.0};
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * TSX_BLOCK) {
    const double vd = yzero ? b[q] : b[q] - y[q];  // the norm (what the stop rule sees) from the fp64 value
    const float v = (float)vd;
    r32[q] = v;
    float rh;
    if (init) {
      rhat32[q] = v;
      p32[split ? tsx_split_pos(q, g) : q] = v;
      rh = v;
    } else {
      rh = rhat32[q];
    }
    sum[0] += (double)rh * (double)v;
    sum[1] += vd * vd;
  }
  tsx_block_reduce_store<2>(sum, partials);
}
// p = r + beta (p - omega v), p in the preconditioner's order
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_pupdate_k32(long long n2, const TsxScalars *__restrict__ sc,
                                                               const float2 *__restrict__ r, const float2 *__restrict__ v,
                                                               float *__restrict__ p32, TsxGeo g, int split) {
  if (sc->done) return;
  const double beta = sc->beta, omega = sc->omega;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n2; q += (long long)gridDim.x * TSX_BLOCK) {
    const float2 rr = r[q], vv = v[q];
    const long long i0 = split ? tsx_split_pos(2 * q, g) : 2 * q, i1 = split ? tsx_split_pos(2 * q + 1, g) : 2 * q + 1;
    const double p0 = (double)p32[i0], p1 = (double)p32[i1];
    p32[i0] = (float)((double)rr.x + beta * (p0 - omega * (double)vv.x));
    p32[i1] = (float)((double)rr.y + beta * (p1 - omega * (double)vv.y));
  }
}
// s = r - alpha v: natural order (the operator's fused dots and the update read it) and, where the preconditioner works in
// colour-split order, a second copy in that order
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_supdate_k32(long long n2, const TsxScalars *__restrict__ sc,
                                                               const float2 *__restrict__ r, const float2 *__restrict__ v,
                                                               float2 *__restrict__ s, float *__restrict__ ssplit, TsxGeo g,
                                                               double *__restrict__ partials) {
  if (sc->done) return;
  const double alpha = sc->alpha;
  double sum[1] = {0.0};  // (s, s) of the values as stored: the half-step stop test (TSX_STAGE_HALF)
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n2; q += (long long)gridDim.x * TSX_BLOCK) {
    const float2 rr = r[q], vv = v[q];
    const float2 o = make_float2((float)((double)rr.x - alpha * (double)vv.x), (float)((double)rr.y - alpha * (double)vv.y));
    sum[0] += (double)o.x * (double)o.x + (double)o.y * (double)o.y;
    s[q] = o;
    if (ssplit) {
      ssplit[tsx_split_pos(2 * q, g)] = o.x;
      ssplit[tsx_split_pos(2 * q + 1, g)] = o.y;
    }
  }
  if (partials) tsx_block_reduce_store<1>(sum, partials);
}
// The same two updates cell by cell, for the scan preconditioner: the thread of a cell has all D streams of it in hand and also
// leaves the D / 2 bf16-pair words of the right-hand side that the red-black passes read (rb[w * Nc + split cell] = streams
// 2 w, 2 w + 1; tsx_k_pcs_rb "RQ") -- the first two passes of an application then read 20 B per cell like every other
// intermediate pass instead of 40 B of fp32 + writing the words themselves.  MODE 0: p = r + beta (p - omega v) (p in split
// order); MODE 1: s = r - alpha v (natural copy s_nat and split copy s_split).  Tail rows (index >= D * Nc) element-wise.
__device__ __forceinline__ unsigned tsx_bf16pair(float lo, float hi) {
  auto b = [](float x) {  // round to nearest even, as tsx_to_bf16 (tsx_pack.hpp): the words equal the ones a pass would leave
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
  };
  return b(lo) | (b(hi) << 16);
}
template <int D, int MODE>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_psupdate_k32c(TsxGeo g, const TsxScalars *__restrict__ sc, const float *__restrict__ r,
                                                                 const float *__restrict__ v, float *__restrict__ p_or_ssplit,
                                                                 float *__restrict__ s_nat, unsigned *__restrict__ rb,
                                                                 double *__restrict__ partials) {
  if (sc->done) return;
  const double beta = sc->beta, omega = sc->omega, alpha = sc->alpha;
  const long long Nc = g.Nc;
  double sum[1] = {0.0};  // MODE 1: (s, s) of the values as stored, the half-step stop test (TSX_STAGE_HALF)
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const int i = (int)(c % g.xm);
    const long long t = c / g.xm;
    const int j = (int)(t % g.ym);
    const long long cs = (t / g.ym) * g.ncol + tsx_split_col(i, j, g.xm);
    float o[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const size_t in = (size_t)d * Nc + c, sp = (size_t)d * Nc + cs;
      if (MODE == 0) {
        o[d] = (float)((double)r[in] + beta * ((double)p_or_ssplit[sp] - omega * (double)v[in]));
      } else {
        o[d] = (float)((double)r[in] - alpha * (double)v[in]);
        s_nat[in] = o[d];
        sum[0] += (double)o[d] * (double)o[d];
      }
      p_or_ssplit[sp] = o[d];
    }
#pragma unroll
    for (int w = 0; w < D / 2; ++w) rb[(size_t)w * Nc + cs] = tsx_bf16pair(o[2 * w], o[2 * w + 1]);
  }
  const long long body = (long long)D * Nc;
  for (long long q = body + (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < g.N; q += (long long)gridDim.x * TSX_BLOCK) {
    const long long sp = tsx_split_pos(q, g);
    if (MODE == 0) {
      p_or_ssplit[sp] = (float)((double)r[q] + beta * ((double)p_or_ssplit[sp] - omega * (double)v[q]));
    } else {
      const float o = (float)((double)r[q] - alpha * (double)v[q]);
      s_nat[q] = o;
      p_or_ssplit[sp] = o;
      sum[0] += (double)o * (double)o;
    }
  }
  if (MODE == 1 && partials) tsx_block_reduce_store<1>(sum, partials);
}
// x += alpha p-hat: the iterate of a solve that stopped at the half step (TSX_STAGE_HALF; runs although `done` is set)
template <typename PT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_xhalf(long long n, const TsxScalars *__restrict__ sc, double *__restrict__ x,
                                                         const PT *__restrict__ ph) {
  const double alpha = sc->alpha;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * TSX_BLOCK)
    x[q] += alpha * (double)ph[q];
}
// x += alpha ph + omega sh (fp64); r = s - omega t; slot0 = (rhat, r), slot1 = (r, r)
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_xrupdate_k32(long long n2, const TsxScalars *__restrict__ sc,
                                                                double2 *__restrict__ x, const float2 *__restrict__ ph,
                                                                const float2 *__restrict__ sh, const float2 *__restrict__ s,
                                                                const float2 *__restrict__ t, const float2 *__restrict__ rhat,
                                                                float2 *__restrict__ r, double *__restrict__ partials) {
  if (sc->done) return;
  const double alpha = sc->alpha, omega = sc->omega;
  double sum[2] = {0.0, 0.0};
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n2; q += (long long)gridDim.x * TSX_BLOCK) {
    const double2 xx = x[q];
    const float2 pp = ph[q], ss2 = sh[q], ss = s[q], tt = t[q], rh = rhat[q];
    double2 xo;
    xo.x = xx.x + alpha * (double)pp.x + omega * (double)ss2.x;
    xo.y = xx.y + alpha * (double)pp.y + omega * (double)ss2.y;
    const float2 ro = make_float2((float)((double)ss.x - omega * (double)tt.x), (float)((double)ss.y - omega * (double)tt.y));
    x[q] = xo;
    r[q] = ro;
    sum[0] += (double)rh.x * (double)ro.x + (double)rh.y * (double)ro.y;
    sum[1] += (double)ro.x * (double)ro.x + (double)ro.y * (double)ro.y;
  }
  tsx_block_reduce_store<2>(sum, partials);
}

// ------------------------------------------------------------------------------------------------
// The explicit (stationary) solver, explicit_ediff of src/pprts_explicit.F90:461-713 on the device: sweeps until the change
// of the iterate is small.  One outer iteration is a defect correction, x += M^-1 (b - A x), with M^-1 = the red-black passes
// of the preconditioner -- for a stationary method, continuing the sweeps from x and sweeping on the defect from zero are the
// same iterate.  d = b - y (y = A x), in the layout / precision the sweeps read their right-hand side in
template <typename DT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_defect(long long n, const TsxScalars *__restrict__ sc, const double *__restrict__ b,
                                                          const double *__restrict__ y, DT *__restrict__ d, TsxGeo g, int split) {
  if (sc->done) return;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * TSX_BLOCK)
    d[split ? tsx_split_pos(q, g) : q] = (DT)(b[q] - y[q]);
}
// x += z; slot0 = (z, z): the change of the iterate, which is what the reference's stop rule measures (:616-617)
template <typename ZT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_xplus(long long n, const TsxScalars *__restrict__ sc, double *__restrict__ x,
                                                         const ZT *__restrict__ z, double *__restrict__ partials) {
  if (sc->done) return;
  double sum[1] = {0.0};
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * TSX_BLOCK) {
    const double zz = (double)z[q];
    x[q] += zz;
    sum[0] += zz * zz;
  }
  tsx_block_reduce_store<1>(sum, partials);
}

// ------------------------------------------------------------------------------------------------
// Scalar stage: one block.  mode bit0: reduce the per-block partials into sc->red (fixed order);
// mode bit1: run the stage's scalar algebra (after the all-reduce when ranks > 1).
// Stop rule restates MyKSPConverged (src/pprts.F90:4437-4486).
enum { TSX_STAGE_INIT = 0, TSX_STAGE_ALPHA = 1, TSX_STAGE_OMEGA = 2, TSX_STAGE_RHO = 3, TSX_STAGE_EXPLICIT = 4, TSX_STAGE_REPLACE = 5, TSX_STAGE_HALF = 6 };

__global__ __launch_bounds__(1024) void tsx_k_scalar(TsxScalars *__restrict__ sc, const double *__restrict__ partials,
                                                     int nblocks, int nslots, int stage, int mode, TsxPeerArArgs ar) {
  // ar.nranks > 1 (peer transport, mode 3): the sum over the ranks happens here, between the reduction of the partial sums and
  // the scalar algebra -- one kernel per reduction point instead of three.  Once the solve is done the ranks still exchange
  // (whatever red holds): a rank writes all-reduce n only after n - 1 has completed everywhere, which is what lets the slots
  // do without acknowledgements
  __shared__ double sm[TSX_NSLOTS][16];
  const bool finished = sc->done != 0;
  if (finished && ar.nranks <= 1) return;
  if (!finished && (mode & 1)) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int s = 0; s < nslots; ++s) {
      double v = 0.0;
      for (int q = threadIdx.x; q < nblocks; q += 1024) v += partials[(size_t)s * TSX_MAX_PARTIAL_BLOCKS + q];
      v = tsx_wave_sum(v);
      if (lane == 0) sm[s][wv] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int s = 0; s < nslots; ++s) {
        double v = 0.0;
        for (int q = 0; q < 16; ++q) v += sm[s][q];
        sc->red[s] = v;
      }
      // the explicit solver's residual is the *mean over ranks of the local norms* of the change (imp_allreduce_mean,
      // src/pprts_explicit.F90:620): take the root before the sum over ranks
      if (stage == TSX_STAGE_EXPLICIT) sc->red[0] = sqrt(sc->red[0]) / (double)(sc->nranks > 0 ? sc->nranks : 1);
    }
    __syncthreads();
  }
  if (ar.nranks > 1) tsx_peer_allreduce_wg(ar, sc->red);  // red[] was written by lane 0 before the barrier above
  if (finished) return;
  if (!(mode & 2) || threadIdx.x != 0) return;
  const double tiny = 2.2250738585072014e-308;
  switch (stage) {
    case TSX_STAGE_INIT: {
      const double rn = sqrt(sc->red[1]);
      sc->rnorm = rn;
      sc->rnorm_true = rn;
      if (!sc->restart) {
        sc->rnorm0 = rn > tiny ? rn : tiny;  // n == 0: store initial norm, no test (:4455-4458)
        sc->hist[0] = rn;
        sc->nhist = 1;
        sc->its = 0;
      }
      sc->rho = sc->red[0];
      sc->rho_old = 1.0;
      sc->alpha = 1.0;
      sc->omega = 1.0;
      sc->beta = 0.0;
      sc->reason = 0;
      if (sc->red[1] == 0.0) {  // exact initial guess: nothing to do
        sc->reason = 2;
        sc->done = 1;
      } else if (rn != rn) {
        sc->reason = -9;
        sc->done = 1;
      } else if (sc->restart && (rn / sc->rnorm0 <= sc->rtol || rn <= sc->atol)) {
        sc->reason = rn / sc->rnorm0 <= sc->rtol ? 2 : 3;
        sc->done = 1;
      }
    } break;
    case TSX_STAGE_ALPHA: {
      const double d1 = sc->red[0];
      if (d1 == 0.0 || d1 != d1) {
        sc->reason = d1 != d1 ? -9 : -5;
        sc->done = 1;
      } else {
        sc->alpha = sc->rho / d1;
      }
    } break;
    case TSX_STAGE_HALF: {
      // BiCGStab's first half step: x + alpha p-hat has the residual s = r - alpha v.  If s already meets the stop rule
      // (MyKSPConverged's two tests, src/pprts.F90:4437-4486, on |s|), the second half -- M^-1 s, A s-hat, the omega update --
      // is work on a converged system: stop here; the host adds alpha p-hat to x (tsx_k_xhalf).  red[0] = (s, s)
      const double sn = sqrt(sc->red[0]);
      const double m = sc->half_margin > 0.0 ? sc->half_margin : 1.0;
      int reason = 0;
      if (sn / sc->rnorm0 <= m * sc->rtol) reason = 2;
      else if (sn <= m * sc->atol) reason = 3;
      if (reason && sc->its + 1 <= sc->maxit) {
        sc->rnorm = sn;
        sc->its += 1;
        if (sc->nhist < 100) sc->hist[sc->nhist++] = sn;
        sc->reason = reason;
        sc->half = 1;
        sc->done = 1;
      }
    } break;
    case TSX_STAGE_OMEGA: {
      const double ts = sc->red[0], tt = sc->red[2];  // SpMV_2 runs with w = s: slot0 = (s,t), slot2 = (t,t)
      sc->omega = tt == 0.0 ? 0.0 : ts / tt;
    } break;
    case TSX_STAGE_RHO: {
      sc->rho_old = sc->rho;
      sc->rho = sc->red[0];
      const double rn = sqrt(sc->red[1]);
      sc->rnorm = rn;
      sc->its += 1;
      if (sc->nhist < 100) sc->hist[sc->nhist++] = rn;
      int reason = 0;
      if (rn / sc->rnorm0 <= sc->rtol) reason = 2;
      else if (rn <= sc->atol) reason = 3;
      else if (sc->its > sc->maxit) reason = -3;
      else if (rn / sc->rnorm0 >= sc->dtol) reason = -4;
      else if (rn != rn) reason = -9;
      else if (sc->rho == 0.0 || sc->omega == 0.0) reason = -5;
      if (reason) {
        sc->reason = reason;
        sc->done = 1;
      } else {
        sc->beta = (sc->rho / sc->rho_old) * (sc->alpha / sc->omega);
      }
    } break;
    case TSX_STAGE_REPLACE: {  // the residual has been replaced by b - A x (fp64): renew rho and the norm, decide the stop rule on it
      sc->rho = sc->red[0];
      const double rn = sqrt(sc->red[1]);
      sc->rnorm = rn;
      sc->rnorm_true = rn;
      if (sc->nhist > 0 && sc->nhist <= 100) sc->hist[sc->nhist - 1] = rn;  // the history holds true norms where they are known
      int reason = 0;
      if (rn / sc->rnorm0 <= sc->rtol) reason = 2;
      else if (rn <= sc->atol) reason = 3;
      else if (sc->its > sc->maxit) reason = -3;
      else if (rn / sc->rnorm0 >= sc->dtol) reason = -4;
      else if (rn != rn) reason = -9;
      else if (sc->rho == 0.0 || sc->omega == 0.0) reason = -5;
      if (reason) {
        sc->reason = reason;
        sc->done = 1;
      } else {
        sc->beta = (sc->rho / sc->rho_old) * (sc->alpha / sc->omega);
      }
    } break;
    case TSX_STAGE_EXPLICIT: {  // stop rule of explicit_ediff (src/pprts_explicit.F90:616-650)
      const double res = sc->red[0];  // residual(iter) = |x_new - x_old|_2, mean over ranks
      sc->its += 1;
      if (sc->its == 1) sc->rnorm0 = res;  // residual(1)
      sc->rnorm = res;
      sc->hist[(sc->its < 100 ? sc->its : 100) - 1] = res;
      sc->nhist = sc->its < 100 ? sc->its : 100;
      const double rel = sc->rnorm0 <= 1.4916681462400413e-154 ? 0.0 : res / sc->rnorm0;  // sqrt(tiny)
      int reason = 0;
      if (res < sc->atol) reason = 3;        // CONVERGED_ATOL is tested (and reported) first
      else if (rel < sc->rtol) reason = 2;
      else if (res != res) reason = -9;
      else if (sc->its >= sc->maxit) reason = -3;  // "did not converge"
      if (reason) {
        sc->reason = reason;
        sc->done = 1;
      }
    } break;
  }
}

// ------------------------------------------------------------------------------------------------
// Layout conversion reference <-> internal (see tsx_internal.hpp) as a transpose through LDS.  The reference vector is
// (dof fastest, level, i, j): one column is a contiguous run of D x (Nz + 1) doubles; the internal planes run along i.
// A workgroup moves a tile of TI columns of one row x TK levels: on the reference side the lanes walk the contiguous
// (dof, level) runs of each column, on the internal side they walk along i (one thread per (level, i, j) read the reference
// side in 80-byte pieces: 2 TB/s).
// Streams that leave the *neighbouring* cell across the low x / low y face of the owned block belong to
// the neighbour rank's cell: they are routed through the halo buffers.
//   import: ref value of +x stream at face i=0  -> sendW (west rank stores it at its cell xm-1)
//           ref value of +y stream at face j=0  -> sendS
//   export: ref value of +x stream at face i=0  <- recvW (== SpMV halo of x), same for y
constexpr int TSX_CV_TI = 32, TSX_CV_TK = 8;
template <int NTOP, int NSIDE, bool EXPORT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_convert_vec(TsxGeo g, double *__restrict__ ref,
                                                               double *__restrict__ v, double *__restrict__ bufW,
                                                               double *__restrict__ bufS, int *__restrict__ nzflag) {
  // nzflag (import only, nullable): set to 1 if any imported value is nonzero (a zero initial guess spares the solve A x0)
  bool nz = false;
  constexpr int D = NTOP + 2 * NSIDE, TI = TSX_CV_TI, TK = TSX_CV_TK, ROWS = TK * D;
  __shared__ double sm[ROWS][TI + 1];
  const int L = g.Nz + 1, xm = g.xm, ym = g.ym, Nz = g.Nz, ncol = g.ncol;
  const long long Nc = g.Nc;
  const int nti = (xm + TI - 1) / TI, ntk = (L + TK - 1) / TK;
  const long long ntile = (long long)nti * ym * ntk;
  double *__restrict__ vt = v + (size_t)D * Nc;
  for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int ti = (int)(tile % nti);
    const long long t2 = tile / nti;
    const int j = (int)(t2 % ym), tk = (int)(t2 / ym);
    const int i0 = ti * TI, k0 = tk * TK;
    const int ni = xm - i0 < TI ? xm - i0 : TI, nk = L - k0 < TK ? L - k0 : TK;
    // internal location of ref (d, k, i, j)
    auto internal = [&](int d, int k, int i) -> double * {
      const int col = j * xm + i;
      if (d < NTOP) {
        if (tsx_inward(d)) return k >= 1 ? v + (size_t)d * Nc + ((size_t)(k - 1) * ym + j) * xm + i : vt + (size_t)d * ncol + col;
        return k < Nz ? v + (size_t)d * Nc + ((size_t)k * ym + j) * xm + i : vt + (size_t)d * ncol + col;
      }
      if (k == Nz) return vt + (size_t)d * ncol + col;
      if (d < NTOP + NSIDE) {
        const int qd = d - NTOP;
        if (!tsx_inward(qd)) return v + (size_t)d * Nc + ((size_t)k * ym + j) * xm + i;
        if (i > 0) return v + (size_t)d * Nc + ((size_t)k * ym + j) * xm + (i - 1);
        if (g.wrap_x) return v + (size_t)d * Nc + ((size_t)k * ym + j) * xm + (xm - 1);
        return bufW + ((size_t)(qd >> 1) * Nz + k) * ym + j;
      }
      const int qd = d - NTOP - NSIDE;
      if (!tsx_inward(qd)) return v + (size_t)d * Nc + ((size_t)k * ym + j) * xm + i;
      if (j > 0) return v + (size_t)d * Nc + ((size_t)k * ym + (j - 1)) * xm + i;
      if (g.wrap_y) return v + (size_t)d * Nc + ((size_t)k * ym + (ym - 1)) * xm + i;
      return bufS + ((size_t)(qd >> 1) * Nz + k) * xm + i;
    };
    // reference side: column ii is the run ref[D * (k0 + L * (i0 + ii + xm * j)) + 0 .. nk * D)
    auto refp = [&](int ii, int rem) { return ref + (size_t)D * ((size_t)k0 + (size_t)L * ((size_t)(i0 + ii) + (size_t)xm * j)) + rem; };
    if (!EXPORT) {
      for (int e = threadIdx.x; e < TI * ROWS; e += TSX_BLOCK) {
        const int ii = e / ROWS, rem = e - ii * ROWS;
        if (ii < ni && rem < nk * D) {
          const double val = *refp(ii, rem);
          sm[rem][ii] = val;
          nz |= val != 0.0;
        }
      }
      __syncthreads();
      for (int e = threadIdx.x; e < TI * ROWS; e += TSX_BLOCK) {
        const int rem = e / TI, ii = e - rem * TI;
        const int kk = rem / D, d = rem - kk * D;
        if (ii < ni && kk < nk) *internal(d, k0 + kk, i0 + ii) = sm[rem][ii];
      }
    } else {
      for (int e = threadIdx.x; e < TI * ROWS; e += TSX_BLOCK) {
        const int rem = e / TI, ii = e - rem * TI;
        const int kk = rem / D, d = rem - kk * D;
        if (ii < ni && kk < nk) sm[rem][ii] = *internal(d, k0 + kk, i0 + ii);
      }
      __syncthreads();
      for (int e = threadIdx.x; e < TI * ROWS; e += TSX_BLOCK) {
        const int ii = e / ROWS, rem = e - ii * ROWS;
        if (ii < ni && rem < nk * D) *refp(ii, rem) = sm[rem][ii];
      }
    }
    __syncthreads();
  }
  if (!EXPORT && nzflag && nz) *nzflag = 1;  // benign race: every writer stores 1
}

// after an import exchange: +x streams received from the east rank (its face i=0) land on my cells xm-1,
// +y streams received from the north rank land on my cells ym-1.
template <int NTOP, int NSIDE>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_import_unpack(TsxGeo g, double *__restrict__ v,
                                                                 const double *__restrict__ recvE,
                                                                 const double *__restrict__ recvN) {
  const int xm = g.xm, ym = g.ym, Nz = g.Nz;
  const long long Nc = g.Nc;
  const long long nx = (long long)(NSIDE / 2) * Nz * ym, ny = (long long)(NSIDE / 2) * Nz * xm;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < nx + ny; q += (long long)gridDim.x * TSX_BLOCK) {
    if (q < nx) {
      const int j = (int)(q % ym);
      const int k = (int)((q / ym) % Nz);
      const int slot = (int)(q / ((long long)ym * Nz));
      const int d = NTOP + 2 * slot + 1;
      if (recvE) v[(size_t)d * Nc + ((size_t)k * ym + j) * xm + (xm - 1)] = recvE[q];
    } else {
      const long long p = q - nx;
      const int i = (int)(p % xm);
      const int k = (int)((p / xm) % Nz);
      const int slot = (int)(p / ((long long)xm * Nz));
      const int d = NTOP + NSIDE + 2 * slot + 1;
      if (recvN) v[(size_t)d * Nc + ((size_t)k * ym + (ym - 1)) * xm + i] = recvN[p];
    }
  }
}


#include "tsx_lut_dev.hpp"  // TsxLutDev, bisection, N-linear weights, one block's interpolation (shared with tsx_dedup.hip)

// The LUT coordinates of every cell before clamping, (aspect, w0, tauz, g) as float32 exactly as src/pprts_base.F90:1517-1533
// forms them, in CELL order: the optical properties arrive level-fastest, the coefficient kernels run column-fastest, and a lane
// that fetches four doubles 512 bytes apart from its neighbour's moves 64 bytes for every 8 it uses (1 GB for 134 MB at
// 256 x 256 x 64).  A tile of 32 columns x 32 levels goes through LDS: read along the levels, written along the columns.
// Grid: (ceil(ncol / 32), ceil(Nz / 32)), 256 threads.
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_cell_samples(TsxGeo g, const double *__restrict__ kabs, const double *__restrict__ ksca,
                                                                const double *__restrict__ gg, const double *__restrict__ dz, double dx,
                                                                float4 *__restrict__ out) {
  constexpr int TC = 32, TK = 32;
  __shared__ float4 tile[TK][TC + 1];
  const int Nz = g.Nz, ncol = g.ncol;
  const int c0 = blockIdx.x * TC, k0 = blockIdx.y * TK;
  for (int e = threadIdx.x; e < TC * TK; e += TSX_BLOCK) {
    const int kk = e % TK, cc = e / TK;
    const int col = c0 + cc, k = k0 + kk;
    if (col >= ncol || k >= Nz) continue;
    const size_t r = (size_t)k + (size_t)Nz * col;  // col = i + xm * j
    const double ka = kabs[r], ks = ksca[r], dzz = dz[r];
    tile[kk][cc] = make_float4((float)(dzz / dx), (float)(ks / fmax(ka + ks, 2.220446049250313e-16)), (float)((ka + ks) * dzz), (float)gg[r]);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < TC * TK; e += TSX_BLOCK) {
    const int cc = e % TC, kk = e / TC;
    const int col = c0 + cc, k = k0 + kk;
    if (col >= ncol || k >= Nz) continue;
    out[(size_t)k * ncol + col] = tile[kk][cc];
  }
}

// diffuse coefficients for every 3-D cell -> planes C[q*Nc + cell] (float).  Inputs in reference layout (k fastest), or -- samp
// != null -- the cells' LUT coordinates from tsx_k_cell_samples.
template <int DD>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_lut_diff2diff(TsxGeo g, TsxLutDev L, const double *__restrict__ kabs,
                                                                 const double *__restrict__ ksca, const double *__restrict__ gg,
                                                                 const double *__restrict__ dz, double dx,
                                                                 const uint8_t *__restrict__ l1d, float *__restrict__ C,
                                                                 unsigned long long *__restrict__ hash,
                                                                 const float4 *__restrict__ samp) {
  // hash (nullable): the block's 64-bit hash for the shared storage, taken while the block is in registers (tsx_dedup.hip
  // would otherwise read all planes again for it)
  const int xm = g.xm, ym = g.ym, Nz = g.Nz;
  const long long Nc = g.Nc;
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const int i = (int)(c % xm);
    const long long t = c / xm;
    const int j = (int)(t % ym);
    const int k = (int)(t / ym);
    if (l1d[k]) {
      if (hash) hash[c] = TSX_DD_H1D;
      continue;
    }
    // src/pprts_base.F90:1517-1533
    float aspect, w0, tauz, gcell;
    if (samp) {
      const float4 v = samp[c];
      aspect = v.x, w0 = v.y, tauz = v.z, gcell = v.w;
    } else {
      const size_t r = (size_t)k + (size_t)Nz * ((size_t)i + (size_t)xm * j);
      const double ka = kabs[r], ks = ksca[r], dzz = dz[r];
      aspect = (float)(dzz / dx);
      w0 = (float)(ks / fmax(ka + ks, 2.220446049250313e-16));
      tauz = (float)((ka + ks) * dzz);
      gcell = (float)gg[r];
    }
    float acc[DD];
    tsx_lut_diff_block<DD>(L, tsx_lut_diff_clamp(L, make_float4(aspect, w0, tauz, gcell)), acc);
#pragma unroll
    for (int q = 0; q < DD; ++q) C[(size_t)q * Nc + c] = acc[q];
    if (hash) {
      unsigned long long hv = TSX_DD_SEED;
#pragma unroll
      for (int q = 0; q < DD; ++q) hv = tsx_dd_hash_step(hv, q, acc[q]);
      hash[c] = tsx_dd_hash_final(hv);
    }
  }
}

// planes -> reference block layout (c fastest, then k, i, j), as real64: what solver%diff2diff holds
template <typename CT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_export_coeff(TsxGeo g, int DD, const CT *__restrict__ C,
                                                                double *__restrict__ ref) {
  const int xm = g.xm, ym = g.ym, Nz = g.Nz;
  const long long total = g.Nc * DD;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < total; q += (long long)gridDim.x * TSX_BLOCK) {
    const long long c = q % g.Nc;
    const int cc = (int)(q / g.Nc);
    const int i = (int)(c % xm);
    const long long t = c / xm;
    const int j = (int)(t % ym);
    const int k = (int)(t / ym);
    ref[(size_t)cc + (size_t)DD * ((size_t)k + (size_t)Nz * ((size_t)i + (size_t)xm * j))] = (double)C[q];
  }
}

// ------------------------------------------------------------------------------------------------
// Operator values: reference block layout (c = dst*D+src fastest, then k, i, j) -> one plane per c,
// x fastest.  LDS-tiled transpose so both sides coalesce.
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_import_coeff(TsxGeo g, int DD, int TI, const TIN *__restrict__ ref,
                                                                TOUT *__restrict__ C) {
  // one block per (j,k, tile of TI i): tile[TI][DD+1]
  extern __shared__ unsigned char smem_raw[];
  TOUT *tile = reinterpret_cast<TOUT *>(smem_raw);
  const int xm = g.xm, ym = g.ym, Nz = g.Nz;
  const int tiles_x = (xm + TI - 1) / TI;
  const long long nt = (long long)tiles_x * ym * Nz;
  for (long long b = blockIdx.x; b < nt; b += gridDim.x) {
    const int tx = (int)(b % tiles_x);
    const int j = (int)((b / tiles_x) % ym);
    const int k = (int)(b / ((long long)tiles_x * ym));
    const int i0 = tx * TI;
    const int ni = xm - i0 < TI ? xm - i0 : TI;
    for (int q = threadIdx.x; q < ni * DD; q += TSX_BLOCK) {
      const int ii = q / DD, c = q % DD;
      tile[ii * (DD + 1) + c] =
          (TOUT)ref[(size_t)c + (size_t)DD * ((size_t)k + (size_t)Nz * ((size_t)(i0 + ii) + (size_t)xm * j))];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < ni * DD; q += TSX_BLOCK) {
      const int c = q / ni, ii = q % ni;
      C[(size_t)c * g.Nc + ((size_t)k * ym + j) * xm + i0 + ii] = tile[ii * (DD + 1) + c];
    }
    __syncthreads();
  }
}

// flag[0] |= 1 if any value is not exactly representable in fp32
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_check_fp32_lossless(long long n, const double *__restrict__ v,
                                                                       int *__restrict__ flag) {
  int bad = 0;
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * TSX_BLOCK) {
    const double a = v[q];
    if ((double)(float)a != a) bad = 1;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// (k,i,j) reference scalar field (z fastest) -> cell-indexed (i fastest)
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_import_cellfield(TsxGeo g, const double *__restrict__ ref,
                                                                    double *__restrict__ out) {
  const int xm = g.xm, ym = g.ym, Nz = g.Nz;
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < g.Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const int i = (int)(c % xm);
    const long long t = c / xm;
    const int j = (int)(t % ym);
    const int k = (int)(t / ym);
    out[c] = ref[(size_t)k + (size_t)Nz * ((size_t)i + (size_t)xm * j)];
  }
}

__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_copy16(long long n, const float4 *__restrict__ a, float4 *__restrict__ b) {
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * TSX_BLOCK) b[q] = a[q];
}

