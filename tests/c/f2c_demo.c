/* A C caller written against TenStream's own C-ABI (c_wrapper/f2c_pprts.h:48-52), in the spirit of
 * c_wrapper/pprts.c: init -> set_global_optical_properties -> solve -> get_result -> destroy, first a solar then
 * a thermal g-point.  Links libtsx_f2c.so.  Writes the float32 result arrays to argv[1] for the test to compare.
 *   argv: out.bin Nx Ny Nz phi0 theta0 */
#include <stdio.h>
#include <stdlib.h>

#include "tsx_f2c.h"

int main(int argc, char **argv) {
  if (argc < 7) return 2;
  int Nx = atoi(argv[2]), Ny = atoi(argv[3]), Nz = atoi(argv[4]);
  float phi0 = (float)atof(argv[5]), theta0 = (float)atof(argv[6]);
  int solver_id = 310, collapseindex = 1;
  double dx = 100, dy = 100;
  float albedo = 0.1f, dz = 40.41f;
  float *hhl = malloc(sizeof(float) * (Nz + 1));
  for (int k = 0; k <= Nz; ++k) hhl[k] = dz * (Nz - k); /* heights, top first */
  size_t nc = (size_t)Nz * Nx * Ny, nl = (size_t)(Nz + 1) * Nx * Ny;
  float *kabs = malloc(4 * nc), *ksca = malloc(4 * nc), *g = malloc(4 * nc), *planck = calloc(nl, 4);
  float *edn = malloc(4 * nl), *eup = malloc(4 * nl), *edir = malloc(4 * nl), *abso = malloc(4 * nc);
  for (int j = 0; j < Ny; ++j)
    for (int i = 0; i < Nx; ++i)
      for (int k = 0; k < Nz; ++k) {
        size_t o = (size_t)k + (size_t)Nz * (i + (size_t)Nx * j);
        kabs[o] = 1e-4f;
        ksca[o] = 1e-4f;
        g[o] = 0.f;
        if (k >= Nz / 3 && k < Nz / 2 && (i + 2 * j) % 5 < 2) { /* a broken cloud deck */
          ksca[o] = 2e-2f;
          kabs[o] = 1e-5f;
          g[o] = 0.85f;
        }
      }
  pprts_f2c_init(0, &solver_id, &Nz, &Nx, &Ny, &dx, &dy, hhl, &phi0, &theta0, &collapseindex);
  FILE *f = fopen(argv[1], "wb");
  /* solar */
  pprts_f2c_set_global_optical_properties(Nz, Nx, Ny, &albedo, kabs, ksca, g, planck);
  pprts_f2c_solve(0, 1000.f);
  pprts_f2c_get_result(Nz, Nx, Ny, edn, eup, abso, edir);
  fwrite(edn, 4, nl, f); fwrite(eup, 4, nl, f); fwrite(abso, 4, nc, f); fwrite(edir, 4, nl, f);
  printf("solar:   edir(srf)=%g edn(srf)=%g eup(toa)=%g\n", edir[Nz], edn[Nz], eup[0]);
  /* thermal */
  for (size_t q = 0; q < nl; ++q) planck[q] = 3.0f + 2.0f * (float)(q % (Nz + 1)) / (float)Nz;
  pprts_f2c_set_global_optical_properties(Nz, Nx, Ny, &albedo, kabs, ksca, g, planck);
  pprts_f2c_solve(0, 0.f);
  pprts_f2c_get_result(Nz, Nx, Ny, edn, eup, abso, edir);
  fwrite(edn, 4, nl, f); fwrite(eup, 4, nl, f); fwrite(abso, 4, nc, f); fwrite(edir, 4, nl, f);
  printf("thermal: edn(srf)=%g eup(toa)=%g\n", edn[Nz], eup[0]);
  fclose(f);
  pprts_f2c_destroy(0);
  return 0;
}
