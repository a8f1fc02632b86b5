"""Pass count of the red-black preconditioner x workload: iterations, residual after the last one and time per solve (zero guess,
reference default tolerances), one solver per workload, the pass counts in a loop.  With the stop test at the half step (round 4)
an 'iteration' may be a half.

usage (GPU box): python scripts/sweeps_grid.py [passes ...]      (default 20 22 24 26 28 30 32)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tenstream_amd import DiffuseSolver, lut  # noqa: E402
from tenstream_amd import synthetic as S  # noqa: E402

passes = [int(v) for v in sys.argv[1:]] or [20, 22, 24, 26, 28, 30, 32]
dev = torch.device("cuda", 0)
WORK = [("3_10", 256, 256, "clouds", 0.3, 20240611), ("3_10", 128, 128, "clouds", 0.3, 20240611), ("3_10", 512, 256, "clouds", 0.3, 20240611),
        ("3_10", 256, 256, "heterogeneous", 0.3, 20240611), ("3_10", 256, 256, "clouds", 0.1, 7), ("3_10", 256, 256, "clouds", 0.6, 99),
        ("3_10", 256, 256, "clouds", 1.0, 20240611), ("3_10", 256, 256, "clouds", 0.0, 20240611), ("3_10", 64, 64, "clouds", 0.3, 20240611),
        ("8_16", 256, 256, "clouds", 0.3, 20240611)]
if os.environ.get("WORK"):
    WORK = [WORK[int(q)] for q in os.environ["WORK"].split(",")]
nz = 64
for solver, nx, ny, field, cover, seed in WORK:
    kabs, ksca, g = S.cloud_field(nx, ny, nz, seed=seed, cover=cover, heterogeneous=field == "heterogeneous")
    kabs, ksca, g = S.delta_scale(kabs, ksca, g)
    b = torch.tensor(S.solar_source(solver, kabs, ksca, g, 50.0, 100.0, np.full((ny, nx), 0.1)), device=dev)
    s = DiffuseSolver(solver, nz, nx, ny)
    s.set_lut_diffuse(lut.synthetic_diffuse_table(solver), lut.diffuse_axes(solver))
    t = lambda v: torch.tensor(v, dtype=torch.float64, device=dev)
    z = torch.zeros((ny, nx, nz), dtype=torch.float64, device=dev)
    s.set_optprop(t(kabs), t(ksca), t(g), torch.full((ny, nx, nz), 50.0, dtype=torch.float64, device=dev), 100.0,
                  torch.zeros(nz, dtype=torch.uint8, device=dev), z, z, torch.full((ny, nx), 0.1, dtype=torch.float64, device=dev))
    x = torch.zeros_like(b)
    row = []
    for p in passes:
        for _ in range(2):
            info = s.solve(b, x, initial_guess_zero=1, pc_sweeps=p - 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            info = s.solve(b, x, initial_guess_zero=1, pc_sweeps=p - 1)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        hist = info.res_hist / info.res_hist[0]
        row.append((p, info.niter, ms, info.rnorm / info.rnorm0, hist[-2] if len(hist) > 1 else 1.0))
    best = min(r[2] for r in row)
    print(f"{solver} {nx}x{ny}x{nz} {field} cover {cover} seed {seed}")
    for p, it, ms, rel, prev in row:
        print(f"   {p:2d} passes: {it:2d} its  {ms:7.3f} ms {'*' if ms == best else ' '}  rel {rel:.2e} (before the last: {prev:.2e})", flush=True)
    s.close()
    del b, x
    torch.cuda.empty_cache()
