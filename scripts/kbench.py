import sys, os, json, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tenstream_amd import DiffuseSolver
Nx = int(os.environ.get("NX", 256)); Ny = Nx; Nz = 64
solver = os.environ.get("SOLVER", "3_10"); D = 10 if solver == "3_10" else 16
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)
coeff = (torch.rand((Ny, Nx, Nz, D*D), device=dev, generator=g, dtype=torch.float32) * (0.9 / D)).contiguous()
l1d = torch.zeros(Nz, dtype=torch.uint8, device=dev)
a11 = torch.zeros((Ny, Nx, Nz), dtype=torch.float64, device=dev); a12 = torch.zeros_like(a11)
alb = torch.full((Ny, Nx), 0.1, dtype=torch.float64, device=dev)
s = DiffuseSolver(solver, Nz, Nx, Ny)
s.set_coeffs(coeff, l1d, a11, a12, alb)
b = torch.rand((Ny, Nx, Nz+1, D), device=dev, dtype=torch.float64, generator=g)
x = torch.zeros_like(b)
info = s.solve(b, x, rtol=1e-3, maxit=8)
ms0 = s.bench_kernel(0, 30); ms1 = s.bench_kernel(1, 10)
print(json.dumps(dict(cpt=os.environ.get("TSX_SPMV_CPT"), nx=Nx, solver=solver, spmv_ms=ms0, spmv_GBps=s.algorithmic_bytes(0)/ms0/1e6, iter_ms=ms1, iter_GBps=s.algorithmic_bytes(1)/ms1/1e6)))
