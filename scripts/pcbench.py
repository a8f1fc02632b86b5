"""A/B of preconditioner variants inside one process/box: iterations, ms per iteration, solve time."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tenstream_amd import DiffuseSolver, synthetic as S, lut as LUT
Nx = Ny = int(os.environ.get("NX", 256)); Nz = 64
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
for pc, sw, mixed, h in ((3, 5, 1, 1), (3, 7, 1, 1), (3, 8, 1, 1), (3, 9, 1, 1), (3, 10, 1, 1), (3, 11, 1, 1), (3, 13, 1, 1)):
    best = 1e9
    for rep in range(3):
        x.zero_(); info = s.solve(b, x, pc=pc, pc_sweeps=sw, fp32_directions=mixed, pc_coeff_fp16=h)
        best = min(best, info.solve_ms)
    print(json.dumps(dict(pc=pc, sweeps=sw, fp32=mixed, half=h, its=info.niter, solve_ms=best, ms_per_it=best / max(info.niter, 1), reason=info.reason)))
