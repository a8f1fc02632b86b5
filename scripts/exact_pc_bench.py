"""The preconditioner in the reference's default arithmetic (tsx_pcx.hip: exact blocks, fp64 iterates) on the metric domain: pass count
against iterations and solve time, next to round 1's zebra rows (TSX_PC_EXACT_SCAN=0).  usage (GPU box): python scripts/exact_pc_bench.py"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tenstream_amd import DiffuseSolver, synthetic as S, lut as LUT
Nx = Ny = int(os.environ.get("NX", 256)); Nz = int(os.environ.get("NZ", 64))
dev = torch.device("cuda", 0)
kabs, ksca, g = S.cloud_field(Nx, Ny, Nz, heterogeneous=os.environ.get("FIELD", "clouds") == "heterogeneous"); kabs, ksca, g = S.delta_scale(kabs, ksca, g)
alb = np.full((Ny, Nx), 0.1)
b = torch.tensor(S.solar_source("3_10", kabs, ksca, g, 50.0, 100.0, alb), device=dev)
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
for sweeps in [int(v) for v in os.environ.get("SWEEPS", "0,5,9,13,17,21,27").split(",")] + ["zebra"]:
    if sweeps == "zebra":
        os.environ["TSX_PC_EXACT_SCAN"] = "0"; sweeps = 0
    s = DiffuseSolver("3_10", Nz, Nx, Ny)
    s.set_lut_diffuse(LUT.synthetic_diffuse_table("3_10"), LUT.diffuse_axes("3_10"))
    s.set_optprop(t(kabs), t(ksca), t(g), torch.full((Ny, Nx, Nz), 50.0, dtype=torch.float64, device=dev), 100.0,
                  torch.zeros(Nz, dtype=torch.uint8, device=dev), t(np.zeros_like(kabs)), t(np.zeros_like(kabs)), t(alb))
    x = torch.zeros_like(b)
    best = 1e9
    for rep in range(3):
        x.zero_(); info = s.solve(b, x, fp32_directions=0, pc_coeff_fp16=0, pc_sweeps=sweeps)
        best = min(best, info.solve_ms)
    pc, sw, scan, _ = s.pc_info()
    print(json.dumps(dict(exact_scan=os.environ.get("TSX_PC_EXACT_SCAN", "1"), pc=pc, passes=sw + 1, its=info.niter, reason=info.reason,
                          rel=info.rnorm / info.rnorm0, solve_ms=round(best, 3), Mcells_s=round(Nx * Ny * Nz / best / 1e3, 1))), flush=True)
    s.close()
