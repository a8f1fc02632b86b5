"""A/B of the scan red-black preconditioner configurations (TSX_PCS_CFG=lseg,nseg,cw; TSX_PC_SCAN=0 = one lane per column)
inside one process/box: iterations, solve time, ms per preconditioner application and per intermediate pass."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tenstream_amd import DiffuseSolver, synthetic as S, lut as LUT
SOLVER = os.environ.get("SOLVER", "3_10")
Nx = Ny = int(os.environ.get("NX", 256)); Nz = int(os.environ.get("NZ", 64))
dev = torch.device("cuda", 0)
kabs, ksca, g = S.cloud_field(Nx, Ny, Nz, heterogeneous=os.environ.get("FIELD", "clouds") == "heterogeneous"); kabs, ksca, g = S.delta_scale(kabs, ksca, g)
alb = np.full((Ny, Nx), 0.1)
b = torch.tensor(S.solar_source(SOLVER, kabs, ksca, g, 50.0, 100.0, alb), device=dev)
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
cfgs = [c for c in os.environ.get("CFGS", "old;4,16,64;8,8,64;4,16,32;8,8,32;4,16,16").split(";") if c]
for cfg in cfgs:
    if cfg.startswith("nodd:"):
        os.environ["TSX_DEDUP"] = "0"; cfg = cfg[5:]
    else:
        os.environ["TSX_DEDUP"] = "1"
    if cfg == "old":
        os.environ["TSX_PC_SCAN"] = "0"; os.environ.pop("TSX_PCS_CFG", None)
    else:
        os.environ["TSX_PC_SCAN"] = "1"; os.environ["TSX_PCS_CFG"] = cfg
    s = DiffuseSolver(SOLVER, Nz, Nx, Ny)
    s.set_lut_diffuse(LUT.synthetic_diffuse_table(SOLVER), LUT.diffuse_axes(SOLVER))
    s.set_optprop(t(kabs), t(ksca), t(g), torch.full((Ny, Nx, Nz), 50.0, dtype=torch.float64, device=dev), 100.0,
                  torch.zeros(Nz, dtype=torch.uint8, device=dev), t(np.zeros_like(kabs)), t(np.zeros_like(kabs)), t(alb))
    x = torch.zeros_like(b)
    best = 1e9
    for rep in range(4):
        x.zero_(); info = s.solve(b, x)
        best = min(best, info.solve_ms)
    out = dict(cfg=cfg, nx=Nx, nz=Nz, its=info.niter, reason=info.reason, rel=info.rnorm / info.rnorm0, hist=[float("%.3g" % (h / info.rnorm0)) for h in list(info.res_hist)[max(0, info.niter - 3):info.niter + 1]], solve_ms=best, Mcells_s=Nx * Ny * Nz / best / 1e3)
    out["dedup"] = s.dedup_info()
    out["spmv_ms"] = s.bench_kernel(0, 20)
    out["pc_apply_ms"] = s.bench_kernel(2, 20)
    out["pc_apply_GBps"] = s.algorithmic_bytes(2) / out["pc_apply_ms"] / 1e6
    if cfg != "old":
        out["pass_ms"] = s.bench_kernel(3, 50)
        out["pass_GBps"] = s.algorithmic_bytes(3) / out["pass_ms"] / 1e6
    print(json.dumps(out), flush=True)
    s.close()
