// tsx_f2c.cpp -- c_wrapper/f2c_pprts.h served by libtsx (see include/tsx_f2c.h).  Host code only: argument
// conversion (float32 -> ireals, minimal_dimension tiling, dz from hhl) around the device pipeline tsx_pprts_*; delta
// scaling, 1-D layer detection and the Eddington coefficients run on the device (tsx_pprts_set_optical_properties).
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

struct F2cState {
  tsx_solver *h = nullptr;
  int solver_id = 0, Nz = 0, Nx = 0, Ny = 0, gNx = 0, gNy = 0;  // gN*: after minimal_dimension tiling (src/pprts.F90:205)
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

}  // namespace

extern "C" int tsx_lut_load_direct_mmap4(tsx_solver *s, const char *tdir_path, const char *sdir_path);

extern "C" void pprts_f2c_init(int fcomm, int *solver_id, int *Nz, int *Nx, int *Ny, double *dx, double *dy, float *hhl,
                               float *phi0, float *theta0, int *collapseindex) {
  (void)fcomm;
  if (g_st.h) {  // already initialised: only the solver type is checked (f2c_pprts.F90:148-182)
    if (*solver_id != g_st.solver_id)
      die("seems you changed the solver type id in between calls... you must destroy the solver first");
    return;
  }
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
  for (int k = 0; k < st.Nz; ++k) st.dz1d[k] = (double)hhl[k] - (double)hhl[k + 1];  // :231-234
  tsx_grid grid;
  memset(&grid, 0, sizeof(grid));
  grid.solver_id = st.solver_id;
  grid.Nz = st.Nz;
  grid.xm = grid.glob_xm = st.gNx;
  grid.ym = grid.glob_ym = st.gNy;
  grid.nranks = 1;
  grid.device = -1;
  chk(tsx_create(&grid, &st.h), "tsx_create");
  chk(tsx_pprts_set_angles(st.h, st.phi0, st.theta0), "tsx_pprts_set_angles");
  // look-up tables: $LUT_BASENAME + the reference's file names (src/optprop_LUT.F90:364-374, 453, 505, 1348)
  const char *base = getenv("LUT_BASENAME");
  if (!base) die("LUT_BASENAME is not set: cannot find the look-up tables");
  const std::string b(base);
  const bool s310 = st.solver_id == TSX_SOLVER_3_10;
  const std::string diff = b + (s310 ? "_diffuse_10" : "_diffuse_16") + ".tau31.w020.aspect_zx23.g6.ds1000.nc.Sdiff.mmap4";
  if (!file_exists(diff)) die("LUT Sdiff data not loaded -- mmap4 file missing: " + diff);
  chk(tsx_lut_load_diffuse_mmap4(st.h, diff.c_str()), "diffuse LUT");
  const char *dirdims = getenv("TSX_LUT_DIRECT_DIMS");  // e.g. "tau11.w05.aspect_zx6.g3.phi3.theta5" for thinned test tables
  const std::string dd = dirdims ? dirdims : "tau31.w020.aspect_zx23.g6.phi19.theta19";
  const std::string dirb = b + (s310 ? "_direct_3_10." : "_direct_8_16.") + dd + ".ds1000.nc.";
  if (file_exists(dirb + "Tdir.mmap4") && file_exists(dirb + "Sdir.mmap4"))
    chk(tsx_lut_load_direct_mmap4(st.h, (dirb + "Tdir.mmap4").c_str(), (dirb + "Sdir.mmap4").c_str()), "direct LUT");
  // (a missing direct table only matters for solar solves and is reported there)
}

extern "C" void pprts_f2c_set_global_optical_properties(int Nz, int Nx, int Ny, float *albedo, float *kabs, float *ksca,
                                                        float *g, float *planck) {
  F2cState &st = g_st;
  if (!st.h) die("pprts_f2c_set_global_optical_properties: call pprts_f2c_init first");
  if (Nz != st.Nz || Nx != st.Nx || Ny != st.Ny) die("pprts_f2c_set_global_optical_properties: shape differs from init");
  const int gx = st.gNx, gy = st.gNy;
  const size_t nc = (size_t)Nz * gx * gy, nl = (size_t)(Nz + 1) * gx * gy;
  std::vector<double> ka(nc), ks(nc), gg(nc), dz(nc), alb((size_t)gx * gy), pl;
  st.have_planck = false;
  if (planck)
    for (size_t q = 0; q < (size_t)(Nz + 1) * Nx * Ny; ++q) st.have_planck |= planck[q] > 0.0f;  // any(oplanck > 0), :303
  if (st.have_planck) pl.resize(nl);
  for (int j = 0; j < gy; ++j)
    for (int i = 0; i < gx; ++i) {
      const int si = i % Nx, sj = j % Ny;  // tiling of degenerate dimensions, src/pprts.F90:2453-2483
      alb[(size_t)i + (size_t)gx * j] = (double)*albedo;
      for (int k = 0; k < Nz; ++k) {
        const size_t o = (size_t)k + (size_t)Nz * ((size_t)i + (size_t)gx * j);
        const size_t q = (size_t)k + (size_t)Nz * ((size_t)si + (size_t)Nx * sj);
        ka[o] = kabs[q];
        ks[o] = ksca[q];
        gg[o] = g[q];
        dz[o] = st.dz1d[k];
      }
      if (st.have_planck)
        for (int k = 0; k <= Nz; ++k)
          pl[(size_t)k + (size_t)(Nz + 1) * ((size_t)i + (size_t)gx * j)] =
              planck[(size_t)k + (size_t)(Nz + 1) * ((size_t)si + (size_t)Nx * sj)];
    }
  // delta scaling (-pprts_delta_scale default true), 1-D layers, Eddington coefficients, lookups: on the device
  chk(tsx_pprts_set_optical_properties(st.h, alb.data(), ka.data(), ks.data(), gg.data(), dz.data(),
                                       st.have_planck ? pl.data() : nullptr, st.dx, st.dy, 1, TSX_HOST),
      "tsx_pprts_set_optical_properties");
}

extern "C" void pprts_f2c_solve(int fcomm, float edirTOA) {
  (void)fcomm;
  if (!g_st.h) die("pprts_f2c_solve: call pprts_f2c_init first");
  const int lsolar = edirTOA > 0;  // lthermal = .not. lsolar, f2c_pprts.F90:340-341
  tsx_ksp_result res;
  chk(tsx_pprts_solve(g_st.h, (double)edirTOA, lsolar, nullptr, &res), "tsx_pprts_solve");
  if (res.reason <= 0)  // src/pprts.F90:4298-4302
    die("***** SOLVER did NOT converge :( -- KSP reason " + std::to_string(res.reason));
}

extern "C" void pprts_f2c_get_result(int Nz, int Nx, int Ny, float *edn, float *eup, float *abso, float *edir) {
  F2cState &st = g_st;
  if (!st.h) die("pprts_f2c_get_result: call pprts_f2c_init first");
  if (Nz != st.Nz || Nx != st.Nx || Ny != st.Ny) die("pprts_f2c_get_result: shape differs from init");
  const int gx = st.gNx, gy = st.gNy, L = Nz + 1;
  std::vector<double> dn((size_t)L * gx * gy), up(dn.size()), di(dn.size()), ab((size_t)Nz * gx * gy);
  chk(tsx_pprts_get_result(st.h, dn.data(), up.data(), ab.data(), di.data(), TSX_HOST), "tsx_pprts_get_result");
  for (int j = 0; j < Ny; ++j)  // res = redn(:, 1:Nx, 1:Ny), f2c_pprts.F90:378-381
    for (int i = 0; i < Nx; ++i) {
      for (int k = 0; k < L; ++k) {
        const size_t o = (size_t)k + (size_t)L * ((size_t)i + (size_t)Nx * j), q = (size_t)k + (size_t)L * ((size_t)i + (size_t)gx * j);
        edn[o] = (float)dn[q];
        eup[o] = (float)up[q];
        edir[o] = (float)di[q];
      }
      for (int k = 0; k < Nz; ++k)
        abso[(size_t)k + (size_t)Nz * ((size_t)i + (size_t)Nx * j)] = (float)ab[(size_t)k + (size_t)Nz * ((size_t)i + (size_t)gx * j)];
    }
}

extern "C" void pprts_f2c_destroy(int lfinalizepetsc) {
  (void)lfinalizepetsc;
  if (g_st.h) tsx_destroy(g_st.h);
  g_st = F2cState();
}
