"""How much does a rank-local preconditioner cost in iterations?  One GPU, global domain, TSX_PC_TILE emulating ranks."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tenstream_amd import DiffuseSolver, synthetic as S, lut as LUT
Nx, Ny, Nz = int(os.environ.get("NX", 256)), int(os.environ.get("NY", 256)), 64
dev = torch.device("cuda", 0)
kabs, ksca, g = S.cloud_field(Nx, Ny, Nz); kabs, ksca, g = S.delta_scale(kabs, ksca, g)
alb = np.full((Ny, Nx), 0.1)
b = torch.tensor(S.solar_source("3_10", kabs, ksca, g, 50.0, 100.0, alb), device=dev)
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
for tile in os.environ.get("TILES", "0,0;128,128;128,64;64,64;32,32").split(";"):
    os.environ["TSX_PC_TILE"] = tile
    s = DiffuseSolver("3_10", Nz, Nx, Ny)
    s.set_lut_diffuse(LUT.synthetic_diffuse_table("3_10"), LUT.diffuse_axes("3_10"))
    s.set_optprop(t(kabs), t(ksca), t(g), torch.full((Ny, Nx, Nz), 50.0, dtype=torch.float64, device=dev), 100.0,
                  torch.zeros(Nz, dtype=torch.uint8, device=dev), t(np.zeros_like(kabs)), t(np.zeros_like(kabs)), t(alb))
    x = torch.zeros_like(b)
    info = s.solve(b, x, pc=int(os.environ.get("PC", 3)), pc_sweeps=int(os.environ.get("SWEEPS", 9)))
    print(json.dumps(dict(tile=tile, its=info.niter, solve_ms=info.solve_ms, reason=info.reason)))
    s.close()
