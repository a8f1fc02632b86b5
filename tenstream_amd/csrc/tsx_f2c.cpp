// tsx_f2c.cpp -- c_wrapper/f2c_pprts.h served by libtsx (see include/tsx_f2c.h).  Host code only: argument
// conversion (float32 -> ireals, minimal_dimension tiling, dz from hhl), the rank-0 scatter / gather semantics of the
// reference wrapper, and the coefficient probe, around the device pipeline tsx_pprts_*; delta scaling, 1-D layer
// detection and the Eddington coefficients run on the device (tsx_pprts_set_optical_properties).
#include <ctype.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/tsx.h"
#include "../../include/tsx_f2c.h"

namespace {

struct F2cComm {  // the communicator behind `fcomm` (tsx_f2c_set_comm); one rank when nothing was attached
  int rank = 0, nranks = 1;
  tsx_exchange_fn exchange = nullptr;
  tsx_allreduce_fn allreduce = nullptr;
  void *ctx = nullptr;
};
F2cComm g_comm;

struct F2cState {
  tsx_solver *h = nullptr;
  int solver_id = 0, Nz = 0, Nx = 0, Ny = 0, gNx = 0, gNy = 0;  // gN*: after minimal_dimension tiling (src/pprts.F90:205)
  int xs = 0, xm = 0, ys = 0, ym = 0;                            // this rank's block of the (tiled) global domain
  double dx = 0, dy = 0, phi0 = 0, theta0 = 0;
  std::vector<double> dz1d;
  bool have_planck = false;
};
F2cState g_st;

[[noreturn]] void die(const std::string &msg) {  // CHKERR: print + abort (src/helper_functions.fypp:888-904)
  fprintf(stderr, "tsx_f2c: %s\n", msg.c_str());
  abort();
}
void chk(int rc, const char *what) {
  if (rc) die(std::string(what) + ": " + tsx_last_error());
}

bool file_exists(const std::string &p) {
  FILE *f = fopen(p.c_str(), "rb");
  if (!f) return false;
  fclose(f);
  return true;
}

// sum over the ranks, in place (imp_allreduce_sum); with zeros everywhere but on the root this is imp_bcast from rank 0,
// with every rank writing its own block into a zeroed global array it is the gather of pprts_get_result_toZero
void allsum(double *v, size_t n) {
  if (g_comm.nranks <= 1) return;
  const size_t chunk = 1u << 22;
  for (size_t o = 0; o < n; o += chunk) {
    const size_t m = n - o < chunk ? n - o : chunk;
    if (g_comm.allreduce(g_comm.ctx, v + o, (int)m)) die("allreduce callback failed");
  }
}
void bcast0(double *v, size_t n) {
  if (g_comm.nranks <= 1) return;
  if (g_comm.rank != 0) memset(v, 0, sizeof(double) * n);
  allsum(v, n);
}

// ---- the options the reference reads from its options database on this path (src/options_database.F90:60-100: later
// sources override earlier ones; a library has no command line, so: ./tenstream.options, then $PETSC_OPTIONS).  "-key
// [value]" pairs, '#' and '!' start comments.  Only the keys this back-end acts on are looked up:
//   -<solar|thermal>_diff_ksp_rtol / _ksp_atol / _ksp_max_it   (KSPSetFromOptions, src/pprts.F90:4258-4260)
//   -<solar|thermal>_diff_explicit                               (src/pprts.F90:2795-2801)
//   -solar_dir_ksp_rtol / _ksp_atol / _ksp_max_it                (explicit_edir, src/pprts_explicit.F90:94-112)
//   -accept_incomplete_solve                                     (src/pprts.F90:4271-4273)
struct Opt {
  std::string key, val;
};
std::vector<Opt> g_opts;
bool g_opts_loaded = false;

void opts_insert_string(const std::string &text) {
  std::vector<std::string> tok;
  size_t i = 0;
  while (i < text.size()) {
    while (i < text.size() && isspace((unsigned char)text[i])) ++i;
    size_t j = i;
    while (j < text.size() && !isspace((unsigned char)text[j])) ++j;
    if (j > i) tok.push_back(text.substr(i, j - i));
    i = j;
  }
  auto is_key = [](const std::string &t) { return t.size() >= 2 && t[0] == '-' && !(isdigit((unsigned char)t[1]) || t[1] == '.'); };
  for (size_t q = 0; q < tok.size(); ++q) {
    if (!is_key(tok[q])) continue;
    Opt o{tok[q].substr(1), ""};
    if (q + 1 < tok.size() && !is_key(tok[q + 1])) o.val = tok[++q];
    bool found = false;
    for (Opt &e : g_opts)
      if (e.key == o.key) {
        e.val = o.val;
        found = true;
      }
    if (!found) g_opts.push_back(o);
  }
}
void opts_load() {
  if (g_opts_loaded) return;
  g_opts_loaded = true;
  if (FILE *f = fopen("tenstream.options", "r")) {
    char line[4096];
    while (fgets(line, sizeof(line), f)) {
      std::string l(line);
      const size_t c = l.find_first_of("#!");
      if (c != std::string::npos) l.resize(c);
      opts_insert_string(l);
    }
    fclose(f);
  }
  if (const char *e = getenv("PETSC_OPTIONS")) opts_insert_string(e);
}
const Opt *opt_find(const std::string &key) {
  opts_load();
  for (const Opt &e : g_opts)
    if (e.key == key) return &e;
  return nullptr;
}
bool opt_real(const std::string &key, double *v) {
  const Opt *o = opt_find(key);
  if (!o || o->val.empty()) return false;
  *v = atof(o->val.c_str());
  return true;
}
bool opt_bool(const std::string &key, bool dflt) {  // PetscOptionsGetBool: a bare key means true
  const Opt *o = opt_find(key);
  if (!o) return dflt;
  const std::string &v = o->val;
  return v.empty() || v == "1" || v == "true" || v == "TRUE" || v == "yes" || v == "on" || v == ".true.";
}

// setup_coord_native (src/pprts_base.F90:721-828): dims = MPI_Dims_create(nranks, 2) = [nyp, nxp] with nyp >= nxp, ranks
// x-fastest, even split xs = (xi * Nx) / nxp, periodic neighbours
void decompose(int nranks, int *nxp, int *nyp) {
  int best_big = nranks, best_small = 1;
  for (int f = 1; f * f <= nranks; ++f)
    if (nranks % f == 0) {
      best_big = nranks / f;
      best_small = f;
    }
  *nyp = best_big;
  *nxp = best_small;
}

void load_luts(tsx_solver *h, int solver_id) {
  // look-up tables: $LUT_BASENAME + the reference's file names (src/optprop_LUT.F90:364-374, 453, 505, 1348)
  const char *base = getenv("LUT_BASENAME");
  if (!base) die("LUT_BASENAME is not set: cannot find the look-up tables");
  const std::string b(base);
  const bool s310 = solver_id == TSX_SOLVER_3_10;
  const std::string diff = b + (s310 ? "_diffuse_10" : "_diffuse_16") + ".tau31.w020.aspect_zx23.g6.ds1000.nc.Sdiff.mmap4";
  if (!file_exists(diff)) die("LUT Sdiff data not loaded -- mmap4 file missing: " + diff);
  chk(tsx_lut_load_diffuse_mmap4(h, diff.c_str()), "diffuse LUT");
  const char *dirdims = getenv("TSX_LUT_DIRECT_DIMS");  // e.g. "tau11.w05.aspect_zx6.g3.phi3.theta5" for thinned test tables
  const std::string dd = dirdims ? dirdims : "tau31.w020.aspect_zx23.g6.phi19.theta19";
  const std::string dirb = b + (s310 ? "_direct_3_10." : "_direct_8_16.") + dd + ".ds1000.nc.";
  if (file_exists(dirb + "Tdir.mmap4") && file_exists(dirb + "Sdir.mmap4"))
    chk(tsx_lut_load_direct_mmap4(h, (dirb + "Tdir.mmap4").c_str(), (dirb + "Sdir.mmap4").c_str()), "direct LUT");
  // (a missing direct table only matters for solar solves and is reported there)
}

}  // namespace

extern "C" void tsx_f2c_set_comm(int rank, int nranks, tsx_exchange_fn exchange, tsx_allreduce_fn allreduce, void *ctx) {
  if (g_st.h) die("tsx_f2c_set_comm: call before pprts_f2c_init (or after pprts_f2c_destroy)");
  if (nranks < 1 || rank < 0 || rank >= nranks) die("tsx_f2c_set_comm: bad rank / nranks");
  if (nranks > 1 && (!exchange || !allreduce)) die("tsx_f2c_set_comm: several ranks need both callbacks");
  g_comm.rank = rank;
  g_comm.nranks = nranks;
  g_comm.exchange = exchange;
  g_comm.allreduce = allreduce;
  g_comm.ctx = ctx;
}

#ifdef TSX_F2C_MPI
// ---- libtsx_f2c_mpi.so: the same library built with an MPI compiler (make mpi; MPICH or any MPI with MPI_Comm_f2c).  `fcomm` is
// then what the reference says it is -- MPI_Comm_c2f(comm), c_wrapper/f2c_pprts.F90:130-230 -- and a multi-rank C caller written
// against TenStream (c_wrapper/pprts.c) links and runs unchanged: no tsx_f2c_set_comm.  The two collectives of include/tsx.h over
// that communicator: the face exchange as non-blocking receives E, W, N, S against sends W, E, S, N with the sender's face as tag
// (two ranks along a periodic axis are each other's W and E neighbour), the sum as MPI_Allreduce.
#include <mpi.h>
namespace {
MPI_Comm g_mpi_comm = MPI_COMM_NULL;
int mpi_exchange(void *, const double *const send[4], double *const recv[4], const size_t count[4], const int peer[4]) {
  static const int want_tag[4] = {1, 0, 3, 2};  // recv[W] <- peer W's send[E] ...
  MPI_Request req[8];
  int n = 0, me = 0;
  MPI_Comm_rank(g_mpi_comm, &me);
  for (int q = 0; q < 4; ++q)
    if (count[q] && peer[q] != me) MPI_Irecv(recv[q], (int)count[q], MPI_DOUBLE, peer[q], want_tag[q], g_mpi_comm, &req[n++]);
  for (int q = 0; q < 4; ++q)
    if (count[q] && peer[q] != me) MPI_Isend(const_cast<double *>(send[q]), (int)count[q], MPI_DOUBLE, peer[q], q, g_mpi_comm, &req[n++]);
  for (int q = 0; q < 4; ++q)  // a rank that is its own neighbour in a direction
    if (count[q] && peer[q] == me) memcpy(recv[q], send[q ^ 1], count[q] * sizeof(double));
  return MPI_Waitall(n, req, MPI_STATUSES_IGNORE) == MPI_SUCCESS ? 0 : 1;
}
int mpi_allreduce(void *, double *inout, int n) {
  return MPI_Allreduce(MPI_IN_PLACE, inout, n, MPI_DOUBLE, MPI_SUM, g_mpi_comm) == MPI_SUCCESS ? 0 : 1;
}
void mpi_attach(int fcomm) {
  int inited = 0;
  MPI_Initialized(&inited);
  if (!inited) die("pprts_f2c_init: MPI is not initialised (the caller owns MPI_Init, as with the reference)");
  g_mpi_comm = MPI_Comm_f2c((MPI_Fint)fcomm);
  int rank = 0, size = 1;
  MPI_Comm_rank(g_mpi_comm, &rank);
  MPI_Comm_size(g_mpi_comm, &size);
  tsx_f2c_set_comm(rank, size, size > 1 ? mpi_exchange : nullptr, size > 1 ? mpi_allreduce : nullptr, nullptr);
}
}  // namespace
// test hook (tests/c/f2c_mpi_selftest.c; needs no GPU): the two callbacks over `fcomm` on host buffers, every face a pattern that
// names sender and face; returns the number of wrong values (summed over the ranks)
extern "C" int tsx_f2c_mpi_selftest(int fcomm, int nxp, int nyp, int count) {
  mpi_attach(fcomm);
  const int rank = g_comm.rank, size = g_comm.nranks;
  if (nxp * nyp != size) return -1;
  const int xi = rank % nxp, yi = rank / nxp;  // x fastest, periodic (src/pprts_base.F90:747-790)
  const int peer[4] = {yi * nxp + (xi + nxp - 1) % nxp, yi * nxp + (xi + 1) % nxp, ((yi + nyp - 1) % nyp) * nxp + xi, ((yi + 1) % nyp) * nxp + xi};
  std::vector<double> sb[4], rb[4];
  const double *send[4];
  double *recv[4];
  size_t cnt[4];
  for (int q = 0; q < 4; ++q) {
    sb[q].resize(count);
    rb[q].assign(count, -1.0);
    for (int i = 0; i < count; ++i) sb[q][i] = rank * 1000.0 + q * 100.0 + (i % 97);
    send[q] = sb[q].data();
    recv[q] = rb[q].data();
    cnt[q] = (size_t)count;
  }
  double bad = 0;
  if (size > 1) {
    if (g_comm.exchange(nullptr, send, recv, cnt, peer)) return -2;
    for (int q = 0; q < 4; ++q)
      for (int i = 0; i < count; ++i) bad += rb[q][i] != peer[q] * 1000.0 + (q ^ 1) * 100.0 + (i % 97);
    double v[3] = {1.0, (double)rank, bad};
    if (g_comm.allreduce(nullptr, v, 3)) return -3;
    if (v[0] != size || v[1] != size * (size - 1) / 2.0) return -4;
    bad = v[2];
  }
  g_comm = F2cComm();
  return (int)bad;
}
#endif

extern "C" void pprts_f2c_init(int fcomm, int *solver_id, int *Nz, int *Nx, int *Ny, double *dx, double *dy, float *hhl,
                               float *phi0, float *theta0, int *collapseindex) {
#ifdef TSX_F2C_MPI
  if (!g_st.h) mpi_attach(fcomm);  // `fcomm` names the communicator, as in the reference
#else
  (void)fcomm;
#endif
  if (g_st.h) {  // already initialised: only the solver type is checked (f2c_pprts.F90:148-182)
    if (*solver_id != g_st.solver_id)
      die("seems you changed the solver type id in between calls... you must destroy the solver first");
    return;
  }
  g_opts.clear();  // a new solver reads the options anew (the reference reads them once per PetscInitialize)
  g_opts_loaded = false;
  // rank 0's values reach every rank and overwrite the caller's (imp_bcast, f2c_pprts.F90:189-230)
  double head[9] = {(double)*solver_id, (double)*Nz, (double)*Nx, (double)*Ny, *dx, *dy, (double)*phi0, (double)*theta0,
                    (double)*collapseindex};
  bcast0(head, 9);
  *solver_id = (int)head[0];
  *Nz = (int)head[1];
  *Nx = (int)head[2];
  *Ny = (int)head[3];
  *dx = head[4];
  *dy = head[5];
  *phi0 = (float)head[6];
  *theta0 = (float)head[7];
  *collapseindex = (int)head[8];
  std::vector<double> ohhl((size_t)*Nz + 1);
  if (g_comm.rank == 0)
    for (int k = 0; k <= *Nz; ++k) ohhl[k] = (double)hhl[k];
  bcast0(ohhl.data(), ohhl.size());

  if (*solver_id != TSX_SOLVER_3_10 && *solver_id != TSX_SOLVER_8_16)  // SOLVER_ID_PPRTS_3_10 / _8_16, f2c_solver_ids.h
    die("solver_id " + std::to_string(*solver_id) + ": this back-end serves 3_10 (310) and 8_16 (816)");
  if (*collapseindex > 1) die("collapseindex > 1 is not supported by this back-end");
  F2cState &st = g_st;
  st.solver_id = *solver_id;
  st.Nz = *Nz;
  st.Nx = *Nx;
  st.Ny = *Ny;
  st.gNx = st.Nx < 3 ? 3 : st.Nx;  // minimal_dimension = 3, src/pprts.F90:205, 493
  st.gNy = st.Ny < 3 ? 3 : st.Ny;
  st.dx = *dx;
  st.dy = *dy;
  st.phi0 = *phi0;
  st.theta0 = *theta0;
  st.dz1d.resize(st.Nz);
  for (int k = 0; k < st.Nz; ++k) st.dz1d[k] = ohhl[k] - ohhl[k + 1];  // :231-234
  int nxp, nyp;
  decompose(g_comm.nranks, &nxp, &nyp);
  if (nxp > st.gNx || nyp > st.gNy) die("more ranks along an axis than columns");
  const int xi = g_comm.rank % nxp, yi = g_comm.rank / nxp;
  st.xs = (xi * st.gNx) / nxp;
  st.xm = ((xi + 1) * st.gNx) / nxp - st.xs;
  st.ys = (yi * st.gNy) / nyp;
  st.ym = ((yi + 1) * st.gNy) / nyp - st.ys;
  tsx_grid grid;
  memset(&grid, 0, sizeof(grid));
  grid.solver_id = st.solver_id;
  grid.Nz = st.Nz;
  grid.xm = st.xm;
  grid.ym = st.ym;
  grid.xs = st.xs;
  grid.ys = st.ys;
  grid.glob_xm = st.gNx;
  grid.glob_ym = st.gNy;
  grid.rank = g_comm.rank;
  grid.nranks = g_comm.nranks;
  grid.neigh_w = ((xi - 1 + nxp) % nxp) + yi * nxp;
  grid.neigh_e = ((xi + 1) % nxp) + yi * nxp;
  grid.neigh_s = xi + ((yi - 1 + nyp) % nyp) * nxp;
  grid.neigh_n = xi + ((yi + 1) % nyp) * nxp;
  grid.device = -1;
  chk(tsx_create(&grid, &st.h), "tsx_create");
  if (g_comm.nranks > 1) chk(tsx_comm_set_callbacks(st.h, g_comm.exchange, g_comm.allreduce, g_comm.ctx), "tsx_comm_set_callbacks");
  chk(tsx_pprts_set_angles(st.h, st.phi0, st.theta0), "tsx_pprts_set_angles");
  load_luts(st.h, st.solver_id);
}

extern "C" void pprts_f2c_set_global_optical_properties(int Nz, int Nx, int Ny, float *albedo, float *kabs, float *ksca,
                                                        float *g, float *planck) {
  F2cState &st = g_st;
  if (!st.h) die("pprts_f2c_set_global_optical_properties: call pprts_f2c_init first");
  const bool root = g_comm.rank == 0;
  if (root && (Nz != st.Nz || Nx != st.Nx || Ny != st.Ny)) die("pprts_f2c_set_global_optical_properties: shape differs from init");
  Nz = st.Nz;  // only rank 0's arguments mean anything (set_global_optical_properties(solver) on the others, :302-316)
  Nx = st.Nx;
  Ny = st.Ny;
  const int gx = st.gNx, gy = st.gNy;
  // rank 0: global fields as ireals, tiled to the minimal dimension (extend_arr, src/pprts.F90:2453-2483); then to all ranks
  // (bcast_and_slice, :2423-2441) which keep their block
  const size_t ncg = (size_t)Nz * gx * gy, nlg = (size_t)(Nz + 1) * gx * gy;
  double flags[2] = {0.0, 0.0};
  if (root) {
    flags[0] = (double)*albedo;
    if (planck)
      for (size_t q = 0; q < (size_t)(Nz + 1) * Nx * Ny; ++q)
        if (planck[q] > 0.0f) flags[1] = 1.0;  // any(oplanck > 0), f2c_pprts.F90:303
  }
  bcast0(flags, 2);
  st.have_planck = flags[1] != 0.0;
  std::vector<double> G[4];  // kabs, ksca, g, planck on the tiled global domain
  for (int f = 0; f < 4; ++f) {
    if (f == 3 && !st.have_planck) continue;
    const int L = f == 3 ? Nz + 1 : Nz;
    G[f].assign(f == 3 ? nlg : ncg, 0.0);
    if (root) {
      const float *src = f == 0 ? kabs : (f == 1 ? ksca : (f == 2 ? g : planck));
      for (int j = 0; j < gy; ++j)
        for (int i = 0; i < gx; ++i) {
          const int si = i % Nx, sj = j % Ny;
          for (int k = 0; k < L; ++k)
            G[f][(size_t)k + (size_t)L * ((size_t)i + (size_t)gx * j)] = src[(size_t)k + (size_t)L * ((size_t)si + (size_t)Nx * sj)];
        }
    }
    bcast0(G[f].data(), G[f].size());
  }
  const int xm = st.xm, ym = st.ym;
  const size_t nc = (size_t)Nz * xm * ym, nl = (size_t)(Nz + 1) * xm * ym;
  std::vector<double> ka(nc), ks(nc), gg(nc), dz(nc), alb((size_t)xm * ym, flags[0]), pl;
  if (st.have_planck) pl.resize(nl);
  for (int j = 0; j < ym; ++j)
    for (int i = 0; i < xm; ++i) {
      const size_t gi = (size_t)(st.xs + i) + (size_t)gx * (st.ys + j), li = (size_t)i + (size_t)xm * j;
      for (int k = 0; k < Nz; ++k) {
        ka[(size_t)k + (size_t)Nz * li] = G[0][(size_t)k + (size_t)Nz * gi];
        ks[(size_t)k + (size_t)Nz * li] = G[1][(size_t)k + (size_t)Nz * gi];
        gg[(size_t)k + (size_t)Nz * li] = G[2][(size_t)k + (size_t)Nz * gi];
        dz[(size_t)k + (size_t)Nz * li] = st.dz1d[k];
      }
      if (st.have_planck)
        for (int k = 0; k <= Nz; ++k) pl[(size_t)k + (size_t)(Nz + 1) * li] = G[3][(size_t)k + (size_t)(Nz + 1) * gi];
    }
  // delta scaling (-pprts_delta_scale default true), 1-D layers, Eddington coefficients, lookups: on the device
  chk(tsx_pprts_set_optical_properties(st.h, alb.data(), ka.data(), ks.data(), gg.data(), dz.data(),
                                       st.have_planck ? pl.data() : nullptr, /* planck_srfc: set_global_optical_properties has none, src/pprts.F90:2341 */ nullptr,
                                       st.dx, st.dy, 1, TSX_HOST),
      "tsx_pprts_set_optical_properties");
}

extern "C" void pprts_f2c_solve(int fcomm, float edirTOA) {
  (void)fcomm;
  if (!g_st.h) die("pprts_f2c_solve: call pprts_f2c_init first");
  double e = (double)edirTOA;  // "incoming solar radiation need only be set by zeroth node", f2c_pprts.F90:334-338
  bcast0(&e, 1);
  const int lsolar = e > 0;  // lthermal = .not. lsolar, f2c_pprts.F90:340-341
  tsx_ksp_result res;
  // tolerances and solver choice as the options database gives them (defaults: determine_ksp_tolerances, FBCGS)
  const std::string pre = lsolar ? "solar_diff_" : "thermal_diff_";
  tsx_ksp_opts o;
  tsx_default_ksp_opts(&o);
  int32_t mx;
  chk(tsx_determine_ksp_tolerances(g_st.h, -1.0, &o.rtol, &o.atol, &mx), "tsx_determine_ksp_tolerances");
  o.maxit = mx;
  double v;
  if (opt_real(pre + "ksp_rtol", &v)) o.rtol = v;
  if (opt_real(pre + "ksp_atol", &v)) o.atol = v;
  if (opt_real(pre + "ksp_max_it", &v)) o.maxit = (int32_t)v;
  o.explicit_solver = opt_bool(pre + "explicit", false) ? 1 : 0;
  o.accept_incomplete_solve = opt_bool("accept_incomplete_solve", false) ? 1 : 0;  // no retry from zero then, src/pprts.F90:4271-4273
  if (o.explicit_solver && !opt_find(pre + "ksp_max_it")) o.maxit = 10000;  // default_max_it, src/pprts_explicit.F90:474
  if (lsolar) {
    double rt = -1, at = -1, mi = -1;
    opt_real("solar_dir_ksp_rtol", &rt);
    opt_real("solar_dir_ksp_atol", &at);
    opt_real("solar_dir_ksp_max_it", &mi);
    chk(tsx_pprts_set_direct_tolerances(g_st.h, rt, at, (int32_t)mi), "tsx_pprts_set_direct_tolerances");
  }
  chk(tsx_pprts_solve(g_st.h, e, lsolar, &o, &res), "tsx_pprts_solve");
  if (res.reason <= 0 && !opt_bool("accept_incomplete_solve", false))  // src/pprts.F90:4271-4273, 4298-4302
    die("***** SOLVER did NOT converge :( -- KSP reason " + std::to_string(res.reason));
}

extern "C" void pprts_f2c_get_result(int Nz, int Nx, int Ny, float *edn, float *eup, float *abso, float *edir) {
  F2cState &st = g_st;
  if (!st.h) die("pprts_f2c_get_result: call pprts_f2c_init first");
  const bool root = g_comm.rank == 0;
  if (root && (Nz != st.Nz || Nx != st.Nx || Ny != st.Ny)) die("pprts_f2c_get_result: shape differs from init");
  Nz = st.Nz;
  Nx = st.Nx;
  Ny = st.Ny;
  const int gx = st.gNx, gy = st.gNy, L = Nz + 1, xm = st.xm, ym = st.ym;
  std::vector<double> dn((size_t)L * xm * ym), up(dn.size()), di(dn.size()), ab((size_t)Nz * xm * ym);
  chk(tsx_pprts_get_result(st.h, dn.data(), up.data(), ab.data(), di.data(), TSX_HOST), "tsx_pprts_get_result");
  // pprts_get_result_toZero (src/pprts.F90:6265-6359): the blocks of all ranks on rank 0
  std::vector<double> Gl[3], Gc;
  const std::vector<double> *loc[3] = {&dn, &up, &di};
  for (int f = 0; f < 3; ++f) {
    Gl[f].assign((size_t)L * gx * gy, 0.0);
    for (int j = 0; j < ym; ++j)
      for (int i = 0; i < xm; ++i)
        for (int k = 0; k < L; ++k)
          Gl[f][(size_t)k + (size_t)L * ((size_t)(st.xs + i) + (size_t)gx * (st.ys + j))] = (*loc[f])[(size_t)k + (size_t)L * ((size_t)i + (size_t)xm * j)];
    allsum(Gl[f].data(), Gl[f].size());
  }
  Gc.assign((size_t)Nz * gx * gy, 0.0);
  for (int j = 0; j < ym; ++j)
    for (int i = 0; i < xm; ++i)
      for (int k = 0; k < Nz; ++k)
        Gc[(size_t)k + (size_t)Nz * ((size_t)(st.xs + i) + (size_t)gx * (st.ys + j))] = ab[(size_t)k + (size_t)Nz * ((size_t)i + (size_t)xm * j)];
  allsum(Gc.data(), Gc.size());
  if (!root) return;  // "only zeroth node gets the results back", f2c_pprts.F90:349
  for (int j = 0; j < Ny; ++j)  // res = redn(:, 1:Nx, 1:Ny), f2c_pprts.F90:378-381
    for (int i = 0; i < Nx; ++i) {
      for (int k = 0; k < L; ++k) {
        const size_t o = (size_t)k + (size_t)L * ((size_t)i + (size_t)Nx * j), q = (size_t)k + (size_t)L * ((size_t)i + (size_t)gx * j);
        edn[o] = (float)Gl[0][q];
        eup[o] = (float)Gl[1][q];
        edir[o] = (float)Gl[2][q];
      }
      for (int k = 0; k < Nz; ++k)
        abso[(size_t)k + (size_t)Nz * ((size_t)i + (size_t)Nx * j)] = (float)Gc[(size_t)k + (size_t)Nz * ((size_t)i + (size_t)gx * j)];
    }
}

extern "C" void pprts_f2c_destroy(int lfinalizepetsc) {
  (void)lfinalizepetsc;
  if (g_st.h) tsx_destroy(g_st.h);
  g_st = F2cState();
}

// ---- coefficient probe (c_wrapper/f2c_pprts.h:54-83): the handle is a one-rank solver that only holds the tables
extern "C" void pprts_f2c_opp_init(const int comm, const int solver_id, void **opp, int *ierr) {
  (void)comm;
  *ierr = 0;
  *opp = nullptr;
  if (solver_id != TSX_SOLVER_3_10 && solver_id != TSX_SOLVER_8_16) {  // the reference knows 3_10 only (f2c_pprts.F90:606-613)
    *ierr = 1;
    die("pprts_f2c_init_OPP not implemented for solver_id " + std::to_string(solver_id));
  }
  tsx_grid grid;
  memset(&grid, 0, sizeof(grid));
  grid.solver_id = solver_id;
  grid.Nz = 1;
  grid.xm = grid.ym = grid.glob_xm = grid.glob_ym = 3;
  grid.nranks = 1;
  grid.device = -1;
  tsx_solver *h = nullptr;
  chk(tsx_create(&grid, &h), "tsx_create");
  load_luts(h, solver_id);
  *opp = h;
}

extern "C" void pprts_f2c_opp_get_coeff(void *opp, const float tauz, const float w0, const float g, const float aspect_zx,
                                        const float phi, const float theta, const int imode, const int lswitch_east,
                                        const int lswitch_north, const int Ncoeff, float *coeff, int *ierr) {
  *ierr = 0;
  if (imode < 1 || imode > 3) {
    *ierr = 1;
    die("imode option " + std::to_string(imode) + " not recognized");
  }
  chk(tsx_opp_get_coeff((tsx_solver *)opp, tauz, w0, g, aspect_zx, phi, theta, imode, lswitch_east, lswitch_north, Ncoeff, coeff),
      "pprts_f2c_opp_get_coeff");
}

extern "C" void pprts_f2c_opp_destroy(void *opp, int *ierr) {
  *ierr = 0;
  if (opp) tsx_destroy((tsx_solver *)opp);
}

extern "C" void pprts_f2c_opp_get_info(void *opp, int *Ndir, int *Ndiff, float *diff_tauz_range, float *diff_w0_range,
                                       float *diff_g_range, float *diff_aspect_zx_range, float *dir_tauz_range,
                                       float *dir_w0_range, float *dir_g_range, float *dir_aspect_zx_range,
                                       float *dir_phi_range, float *dir_theta_range, int *ierr) {
  *ierr = 0;
  float r[20];
  int32_t nd = 0, nf = 0;
  chk(tsx_opp_get_info((tsx_solver *)opp, &nd, &nf, r), "pprts_f2c_opp_get_info");
  *Ndir = nd;
  *Ndiff = nf;
  float *dst[10] = {diff_tauz_range, diff_w0_range, diff_g_range, diff_aspect_zx_range, dir_tauz_range,
                    dir_w0_range,    dir_g_range,   dir_aspect_zx_range, dir_phi_range, dir_theta_range};
  for (int q = 0; q < 10; ++q) {
    dst[q][0] = r[2 * q];
    dst[q][1] = r[2 * q + 1];
  }
}
