"""usage (GPU box): python scripts/hist.py [--nx N --ny N --nz N --field F --solver S] -- residual history of the default solve"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tenstream_amd import DiffuseSolver, lut
from tenstream_amd import synthetic as S

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=256)
ap.add_argument("--ny", type=int, default=256)
ap.add_argument("--nz", type=int, default=64)
ap.add_argument("--field", default="clouds")
ap.add_argument("--solver", default="3_10")
ap.add_argument("--rtol", type=float, default=1e-8)
ap.add_argument("--pc-sweeps", type=int, default=None)
ap.add_argument("--cover", type=float, default=0.3)
ap.add_argument("--seed", type=int, default=20240611)
a = ap.parse_args()
dev = torch.device("cuda", 0)
kabs, ksca, g = S.cloud_field(a.nx, a.ny, a.nz, seed=a.seed, cover=a.cover, heterogeneous=a.field == "heterogeneous")
kabs, ksca, g = S.delta_scale(kabs, ksca, g)
b = torch.tensor(S.solar_source(a.solver, kabs, ksca, g, 50.0, 100.0, np.full((a.ny, a.nx), 0.1)), device=dev)
s = DiffuseSolver(a.solver, a.nz, a.nx, a.ny)
s.set_lut_diffuse(lut.synthetic_diffuse_table(a.solver), lut.diffuse_axes(a.solver))
t = lambda v: torch.tensor(v, dtype=torch.float64, device=dev)
z = torch.zeros((a.ny, a.nx, a.nz), dtype=torch.float64, device=dev)
s.set_optprop(t(kabs), t(ksca), t(g), torch.full((a.ny, a.nx, a.nz), 50.0, dtype=torch.float64, device=dev), 100.0,
              torch.zeros(a.nz, dtype=torch.uint8, device=dev), z, z, torch.full((a.ny, a.nx), 0.1, dtype=torch.float64, device=dev))
x = torch.zeros_like(b)
i = s.solve(b, x, rtol=a.rtol, atol=1e-30, pc_sweeps=a.pc_sweeps)
print(a.nx, a.ny, a.nz, a.field, a.solver, "cover", a.cover, "seed", a.seed, "sweeps", a.pc_sweeps, "its", i.niter, "rel. residuals:", " ".join(f"{v / i.res_hist[0]:.2e}" for v in i.res_hist))
