"""What hipGraph replay buys one BiCGStab iteration (round 6: measured instead of argued).  tsx_bench_kernel(1): the iteration's kernels
enqueued eagerly, `reps` times back to back; tsx_bench_kernel(5): the same iteration captured once from the solver's stream and
replayed `reps` times.  With the flow kernel (default: ~90 launches per iteration) and with a launch per pass (TSX_PC_FLOW=0: ~290).
usage (GPU box): python scripts/graph_ab.py"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tenstream_amd import DiffuseSolver, synthetic as S, lut as LUT
dev = torch.device("cuda", 0)
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
for Nx, Ny in ((256, 256), (128, 128), (64, 64)):
    Nz = 64
    kabs, ksca, g = S.cloud_field(Nx, Ny, Nz); kabs, ksca, g = S.delta_scale(kabs, ksca, g)
    alb = np.full((Ny, Nx), 0.1)
    b = torch.tensor(S.solar_source("3_10", kabs, ksca, g, 50.0, 100.0, alb), device=dev)
    for flow in ("1", "0"):
        os.environ["TSX_PC_FLOW"] = flow
        s = DiffuseSolver("3_10", Nz, Nx, Ny)
        s.set_lut_diffuse(LUT.synthetic_diffuse_table("3_10"), LUT.diffuse_axes("3_10"))
        s.set_optprop(t(kabs), t(ksca), t(g), torch.full((Ny, Nx, Nz), 50.0, dtype=torch.float64, device=dev), 100.0,
                      torch.zeros(Nz, dtype=torch.uint8, device=dev), t(np.zeros_like(kabs)), t(np.zeros_like(kabs)), t(alb))
        x = torch.zeros_like(b)
        info = s.solve(b, x)
        out = dict(nx=Nx, ny=Ny, flow_kernel=flow == "1", solve_ms=round(info.solve_ms, 3), its=info.niter)
        for rep in range(2):
            out[f"eager_ms_{rep}"] = round(s.bench_kernel(1, 20), 4)
            try:
                out[f"graph_ms_{rep}"] = round(s.bench_kernel(5, 20), 4)
            except Exception as e:   # noqa: BLE001
                out[f"graph_ms_{rep}"] = str(e)[:200]
        print(json.dumps(out), flush=True)
        s.close()
