// tsx_spmv_impl.hpp -- host launch code of the operator apply for one stream configuration.  Included by
// tsx_spmv_3_10.hip / tsx_spmv_8_16.hip with TSX_SPMV_NTOP and TSX_SPMV_TAG defined: the 96 kernel variants per
// configuration (6 dot/precision combinations x fp32/fp64 blocks x 1|2 cells per thread x halo x 1-D layers) compile in
// parallel, one translation unit each.
#include "tsx_host.hpp"
#include "tsx_kernels_spmv.hpp"

template <int NTOP, int NSIDE, typename XT = double>
static int halo_update_t(tsx_solver *s, const XT *v, bool in_solve) {
  const TsxGeo &g = s->geo;
  if (g.wrap_x && g.wrap_y) return TSX_OK;
  const long long n = (long long)s->halo_x_elems + (long long)s->halo_y_elems;
  hipLaunchKernelGGL((tsx_k_halo_pack<NTOP, NSIDE, XT>), dim3(grid_for(n)), dim3(TSX_BLOCK), 0, s->stream, g, v, (XT *)s->sendW,
                     (XT *)s->sendE, (XT *)s->sendS, (XT *)s->sendN, in_solve ? &s->scal->done : (const int *)nullptr);
  return tsx_face_exchange_elems(s, s->stream, sizeof(XT));
}

// part 0: whole grid; 1: interior (no halo reads); 2: frame, partial sums behind those of part 1
template <int NTOP, int NSIDE, int FUSE, typename CT, int CPT, typename XT, typename WT, bool HALO, bool HAS1D, typename YT = double>
static void launch_spmv_variant(tsx_solver *s, const XT *x, YT *y, const WT *w, const int *done, int part) {
  const TsxGeo &g = s->geo;
  const int nbmain = grid_for(g.Nc / CPT, TSX_MAX_PARTIAL_BLOCKS - TSX_FRAME_BLOCKS);
  const int nb = part == 2 ? grid_for(frame_groups(g, CPT), TSX_FRAME_BLOCKS) : nbmain;
  if constexpr (std::is_same<CT, float>::value) {
    if (s->dd_on) {  // shared storage of identical blocks (tsx_dedup.hip)
      hipLaunchKernelGGL((tsx_k_spmv_w<NTOP, NSIDE, float, FUSE, CPT, XT, WT, HALO, HAS1D, true, YT>), dim3(nb), dim3(TSX_BLOCK), 0,
                         s->stream, g, (const float *)s->dd_coef_e, (const int *)s->dd_cidx, (long long)s->dd_nent, s->l1d, s->a11,
                         s->a12, s->albedo, x, y, (const XT *)s->recvW, (const XT *)s->recvE, (const XT *)s->recvS,
                         (const XT *)s->recvN, w, s->partials + (part == 2 ? nbmain : 0), done, part);
      return;
    }
  }
  hipLaunchKernelGGL((tsx_k_spmv_w<NTOP, NSIDE, CT, FUSE, CPT, XT, WT, HALO, HAS1D, false, YT>), dim3(nb), dim3(TSX_BLOCK), 0, s->stream, g,
                     (const CT *)s->coef, (const int *)nullptr, 0ll, s->l1d, s->a11, s->a12, s->albedo, x, y, (const XT *)s->recvW,
                     (const XT *)s->recvE, (const XT *)s->recvS, (const XT *)s->recvN, w, s->partials + (part == 2 ? nbmain : 0),
                     done, part);
}

template <int NTOP, int NSIDE, int FUSE, typename CT, int CPT, typename XT, typename WT, typename YT = double>
static void launch_spmv_flags(tsx_solver *s, const XT *x, YT *y, const WT *w, const int *done, int part) {
  const bool halo = !(s->geo.wrap_x && s->geo.wrap_y) && part != 1, has1d = s->any_l1d;
  if (halo) {
    if (has1d) launch_spmv_variant<NTOP, NSIDE, FUSE, CT, CPT, XT, WT, true, true, YT>(s, x, y, w, done, part);
    else launch_spmv_variant<NTOP, NSIDE, FUSE, CT, CPT, XT, WT, true, false, YT>(s, x, y, w, done, part);
  } else {
    if (has1d) launch_spmv_variant<NTOP, NSIDE, FUSE, CT, CPT, XT, WT, false, true, YT>(s, x, y, w, done, part);
    else launch_spmv_variant<NTOP, NSIDE, FUSE, CT, CPT, XT, WT, false, false, YT>(s, x, y, w, done, part);
  }
}

template <int NTOP, int NSIDE, int FUSE, typename XT, typename WT, typename YT = double>
static int launch_spmv_t(tsx_solver *s, const XT *x, YT *y, const WT *w, bool in_solve) {
  const int *done = in_solve ? &s->scal->done : nullptr;
  const int cpt = spmv_cpt(s);
  auto launch = [&](int part) {
    if (s->coef_bytes == 4) {
      if (cpt == 2) launch_spmv_flags<NTOP, NSIDE, FUSE, float, 2, XT, WT, YT>(s, x, y, w, done, part);
      else launch_spmv_flags<NTOP, NSIDE, FUSE, float, 1, XT, WT, YT>(s, x, y, w, done, part);
    } else {
      if (cpt == 2) launch_spmv_flags<NTOP, NSIDE, FUSE, double, 2, XT, WT, YT>(s, x, y, w, done, part);
      else launch_spmv_flags<NTOP, NSIDE, FUSE, double, 1, XT, WT, YT>(s, x, y, w, done, part);
    }
  };
  if (!spmv_split(s)) {
    int rc = halo_update_t<NTOP, NSIDE, XT>(s, x, in_solve);
    if (rc) return rc;
    launch(0);
    HIPCHK(hipGetLastError());
    return TSX_OK;
  }
  // overlap: pack -> [exchange on comm_stream || interior cells on stream] -> frame cells
  const TsxGeo &g = s->geo;
  const long long n = (long long)s->halo_x_elems + (long long)s->halo_y_elems;
  hipLaunchKernelGGL((tsx_k_halo_pack<NTOP, NSIDE, XT>), dim3(grid_for(n)), dim3(TSX_BLOCK), 0, s->stream, g, x, (XT *)s->sendW,
                     (XT *)s->sendE, (XT *)s->sendS, (XT *)s->sendN, done);
  HIPCHK(hipEventRecord(s->ev_pack, s->stream));
  HIPCHK(hipStreamWaitEvent(s->comm_stream, s->ev_pack, 0));
  launch(1);  // queued before the (possibly host-synchronous) exchange so that it runs underneath it
  HIPCHK(hipGetLastError());
  int rc = tsx_face_exchange_elems(s, s->comm_stream, sizeof(XT));
  if (rc) return rc;
  HIPCHK(hipEventRecord(s->ev_recv, s->comm_stream));
  HIPCHK(hipStreamWaitEvent(s->stream, s->ev_recv, 0));
  launch(2);
  HIPCHK(hipGetLastError());
  return TSX_OK;
}

#define TSX_CAT2(a, b) a##b
#define TSX_CAT(a, b) TSX_CAT2(a, b)

int TSX_CAT(tsx_spmv_launch_, TSX_SPMV_TAG)(tsx_solver *s, int combo, const void *x, void *yv, const void *w, bool in_solve) {
  constexpr int NT = TSX_SPMV_NTOP, NS = 4;
  double *y = (double *)yv;
  switch (combo) {
    // fp32 Krylov vectors (tsx_ksp_opts.fp32_directions = 2): v = A p-hat and t = A s-hat stored in fp32
    case TSX_SPMV_1FF_YF: return launch_spmv_t<NT, NS, 1, float, float, float>(s, (const float *)x, (float *)yv, (const float *)w, in_solve);
    case TSX_SPMV_5FF_YF: return launch_spmv_t<NT, NS, 5, float, float, float>(s, (const float *)x, (float *)yv, (const float *)w, in_solve);
    case TSX_SPMV_0DD: return launch_spmv_t<NT, NS, 0, double, double>(s, (const double *)x, y, (const double *)w, in_solve);
    case TSX_SPMV_1FF: return launch_spmv_t<NT, NS, 1, float, float>(s, (const float *)x, y, (const float *)w, in_solve);
    case TSX_SPMV_5FD: return launch_spmv_t<NT, NS, 5, float, double>(s, (const float *)x, y, (const double *)w, in_solve);
    case TSX_SPMV_1DF: return launch_spmv_t<NT, NS, 1, double, float>(s, (const double *)x, y, (const float *)w, in_solve);
    case TSX_SPMV_1DD: return launch_spmv_t<NT, NS, 1, double, double>(s, (const double *)x, y, (const double *)w, in_solve);
    case TSX_SPMV_5DD: return launch_spmv_t<NT, NS, 5, double, double>(s, (const double *)x, y, (const double *)w, in_solve);
  }
  tsx_set_error("tsx_spmv_launch: unknown variant");
  return TSX_ERR_ARG;
}

int TSX_CAT(tsx_halo_update_, TSX_SPMV_TAG)(tsx_solver *s, const double *v, bool in_solve) {
  return halo_update_t<TSX_SPMV_NTOP, 4, double>(s, v, in_solve);
}
