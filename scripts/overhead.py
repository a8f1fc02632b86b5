"""Where does wall time go around one tsx_diff_solve? (host overhead vs device solve time)"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tenstream_amd import DiffuseSolver, synthetic as S, lut as LUT
Nx = Ny = 256; Nz = 64
dev = torch.device("cuda", 0)
kabs, ksca, g = S.cloud_field(Nx, Ny, Nz); kabs, ksca, g = S.delta_scale(kabs, ksca, g)
alb = np.full((Ny, Nx), 0.1)
b = torch.tensor(S.solar_source("3_10", kabs, ksca, g, 50.0, 100.0, alb), device=dev)
s = DiffuseSolver("3_10", Nz, Nx, Ny)
s.set_lut_diffuse(LUT.synthetic_diffuse_table("3_10"), LUT.diffuse_axes("3_10"))
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
s.set_optprop(t(kabs), t(ksca), t(g), torch.full((Ny, Nx, Nz), 50.0, dtype=torch.float64, device=dev), 100.0,
              torch.zeros(Nz, dtype=torch.uint8, device=dev), t(np.zeros_like(kabs)), t(np.zeros_like(kabs)), t(alb))
x = torch.zeros_like(b)
for ce in (1, 2, 4, 5, 10, 16):
    for rep in range(3):
        x.zero_(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        info = s.solve(b, x, check_every=ce)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
    print(json.dumps(dict(check_every=ce, its=info.niter, wall_ms=wall, solve_ms=info.solve_ms, import_ms=info.import_ms, export_ms=info.export_ms)))
