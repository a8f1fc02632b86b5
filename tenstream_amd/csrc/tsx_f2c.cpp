// tsx_f2c.cpp -- c_wrapper/f2c_pprts.h served by libtsx (see include/tsx_f2c.h).  Host code only: the per-call
// preparations the reference does in set_optical_properties (delta scaling src/pprts.F90:1903-1917, 1-D layer
// detection :669-677 + :708-719, Eddington coefficients :1962-1992 / src/eddington.F90:173-241), then the device
// pipeline tsx_pprts_*.
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
  double dx = 0, dy = 0, phi0 = 0, theta0 = 0, mu0 = 0;
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

// eddington_coeff_ec, src/eddington.F90:173-241
void eddington_ec(double dtau, double w0, double g, double mu0, double &t, double &r, double &rdir, double &sdir, double &tdir) {
  const double f = 0.75 * g;
  const double g1 = 2.0 - w0 * (1.25 + f), g2 = w0 * (0.75 - f), g3 = 0.5 - mu0 * f;
  const double slant = fmax(dtau / fmax(sqrt(DBL_MIN), mu0), 0.0);
  if (slant > 1e-6) {
    const double g4 = 1.0 - g3, alpha1 = g1 * g4 + g2 * g3, alpha2 = g1 * g3 + g2 * g4;
    const double A = sqrt(fmax((g1 - g2) * (g1 + g2), 1e-12));
    double k_mu0 = A * mu0;
    const double k_g3 = A * g3, k_g4 = A * g4, e0 = exp(-slant), e = exp(-A * dtau), e2 = e * e, k2e = 2 * A * e;
    tdir = e0;
    if (fabs(k_mu0 - 1.0) <= 10 * DBL_EPSILON) k_mu0 = 1 - 10 * DBL_EPSILON;
    double beta = 1 / (A + g1 + (A - g1) * e2);
    r = g2 * (1 - e2) * beta;
    t = k2e * beta;
    beta = w0 * beta / (1 - k_mu0 * k_mu0);
    sdir = beta * (k2e * (g4 + alpha1 * mu0) - e0 * ((1 + k_mu0) * (alpha1 + k_g4) - (1 - k_mu0) * (alpha1 - k_g4) * e2));
    rdir = beta * ((1 - k_mu0) * (alpha2 + k_g3) - (1 + k_mu0) * (alpha2 - k_g3) * e2 - k2e * (g3 - alpha2 * mu0) * e0);
  } else {
    t = 1.0 - g1 * dtau;
    r = g2 * dtau;
    sdir = (1.0 - g3) * (w0 * dtau);
    rdir = g3 * (w0 * dtau);
    tdir = 1.0 - slant;
  }
}

// delta_scale with f = g**2, src/helper_functions.fypp:1622-1666
void delta_scale(double &kabs, double &ksca, double &g) {
  const double f = g * g;
  double dtau = kabs + ksca;
  if (dtau < DBL_EPSILON) return;
  double w0 = ksca / dtau;
  if (g >= 1.0 - DBL_EPSILON * 10) {
    dtau *= (1.0 - w0);
    w0 = 0;
    g = 0;
  } else {
    dtau *= (1.0 - w0 * f);
    g = (g - f) / (1.0 - f);
    w0 = w0 * (1.0 - f) / (1.0 - f * w0);
  }
  kabs = dtau * (1.0 - w0);
  ksca = dtau * w0;
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
  if (*solver_id != TSX_SOLVER_3_10) die("solver_id " + std::to_string(*solver_id) + ": only 3_10 (310) is served by this back-end");
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
  st.mu0 = st.theta0 >= 90.0 ? 0.0 : fmax(cos(st.theta0 * 3.14159265358979323846 / 180.0), 0.0);
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
  const std::string diff = b + "_diffuse_10.tau31.w020.aspect_zx23.g6.ds1000.nc.Sdiff.mmap4";
  if (!file_exists(diff)) die("LUT Sdiff data not loaded -- mmap4 file missing: " + diff);
  chk(tsx_lut_load_diffuse_mmap4(st.h, diff.c_str()), "diffuse LUT");
  const char *dirdims = getenv("TSX_LUT_DIRECT_DIMS");  // e.g. "tau11.w05.aspect_zx6.g3.phi3.theta5" for thinned test tables
  const std::string dd = dirdims ? dirdims : "tau31.w020.aspect_zx23.g6.phi19.theta19";
  const std::string dirb = b + "_direct_3_10." + dd + ".ds1000.nc.";
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
  std::vector<double> ka(nc), ks(nc), gg(nc), dz(nc), a11(nc), a12(nc), a13(nc), a23(nc), a33(nc), alb((size_t)gx * gy), pl;
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
        delta_scale(ka[o], ks[o], gg[o]);  // -pprts_delta_scale default true
      }
      if (st.have_planck)
        for (int k = 0; k <= Nz; ++k)
          pl[(size_t)k + (size_t)(Nz + 1) * ((size_t)i + (size_t)gx * j)] =
              planck[(size_t)k + (size_t)(Nz + 1) * ((size_t)si + (size_t)Nx * sj)];
    }
  // which layers are 1-D: src/pprts.F90:669-677, then the count is applied from the top (:708-719)
  std::vector<uint8_t> l1d(Nz, 0);
  const double twostr_ratio = 2.0;  // src/tenstream_options.F90
  auto exceeds = [&](int k) { return st.dz1d[k] / st.dx > twostr_ratio; };
  l1d[Nz - 1] = exceeds(Nz - 1);
  for (int k = Nz - 2; k >= 0; --k)
    if (exceeds(k)) {
      for (int q = 0; q <= k; ++q) l1d[q] = 1;
      break;
    }
  int n1d = 0;
  for (int k = 0; k < Nz; ++k) n1d += l1d[k];
  for (int k = 0; k < n1d; ++k) l1d[k] = 1;
  for (size_t o = 0; o < nc; ++o) {
    const int k = (int)(o % Nz);
    if (!l1d[k]) continue;
    const double ext = fmax(DBL_MIN, ka[o] + ks[o]);
    eddington_ec(dz[o] * ext, ks[o] / ext, gg[o], st.mu0, a11[o], a12[o], a13[o], a23[o], a33[o]);
  }
  chk(tsx_pprts_set_optprop(st.h, ka.data(), ks.data(), gg.data(), dz.data(), st.dx, st.dy, alb.data(), l1d.data(), a11.data(),
                            a12.data(), a13.data(), a23.data(), a33.data(), st.have_planck ? pl.data() : nullptr, TSX_HOST),
      "tsx_pprts_set_optprop");
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
