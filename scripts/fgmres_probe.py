"""How many applications of M^-1 would another outer Krylov method need on the real problem with the real (reduced-precision)
preconditioner?  The library's operator (s.apply) and preconditioner (s.pc_apply, the default 22 red-black passes on the packed
blocks) on device tensors; the outer iteration in torch (fp64).  Flexible BiCGStab here must reproduce the library's count.

usage (GPU box): python scripts/fgmres_probe.py [nx ny [field]]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tenstream_amd import DiffuseSolver, lut, synthetic  # noqa: E402

nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 256)
field = sys.argv[3] if len(sys.argv) > 3 else "clouds"
nz, solver = 64, os.environ.get("SOLVER", "3_10")
dev = torch.device("cuda", 0)
kabs, ksca, g = synthetic.cloud_field(nx, ny, nz, seed=20240611, heterogeneous=field == "heterogeneous")
kabs, ksca, g = synthetic.delta_scale(kabs, ksca, g)
b = torch.tensor(synthetic.solar_source(solver, kabs, ksca, g, 50.0, 100.0, np.full((ny, nx), 0.1)), device=dev)
s = DiffuseSolver(solver, nz, nx, ny)
s.set_lut_diffuse(lut.synthetic_diffuse_table(solver), lut.diffuse_axes(solver))
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
z = torch.zeros((ny, nx, nz), dtype=torch.float64, device=dev)
s.set_optprop(t(kabs), t(ksca), t(g), torch.full((ny, nx, nz), 50.0, dtype=torch.float64, device=dev), 100.0,
              torch.zeros(nz, dtype=torch.uint8, device=dev), z, z, torch.full((ny, nx), 0.1, dtype=torch.float64, device=dev))
x = torch.zeros_like(b)
info = s.solve(b, x, initial_guess_zero=1)
print(f"library: {info.niter} iterations, reason {info.reason}, rel {info.rnorm / info.rnorm0:.2e}, hist {[f'{v / info.rnorm0:.1e}' for v in info.res_hist]}")
SW = int(os.environ.get("PC_SWEEPS", 21))
A = lambda v: s.apply(v)
M = lambda v: s.pc_apply(v, pc=3, sweeps=SW, mixed=True)
dot = lambda a, c: float((a * c).sum())
nrm = lambda a: float(torch.linalg.vector_norm(a))
r0 = nrm(b)
rtol = 1e-5


def fbcgs():
    xx = torch.zeros_like(b); r = b.clone(); rh = r.clone(); p = r.clone(); rho = dot(rh, r); apps = 0; hist = []
    for it in range(50):
        ph = M(p); apps += 1; v = A(ph); alpha = rho / dot(rh, v)
        sres = r - alpha * v
        hist.append(nrm(sres) / r0)
        if hist[-1] <= rtol:
            return apps, hist
        sh = M(sres); apps += 1; tt = A(sh); omega = dot(tt, sres) / dot(tt, tt)
        xx += alpha * ph + omega * sh; r = sres - omega * tt
        hist.append(nrm(r) / r0)
        if hist[-1] <= rtol:
            return apps, hist
        rho_new = dot(rh, r); beta = (rho_new / rho) * (alpha / omega); rho = rho_new
        p = r + beta * (p - omega * v)
    return apps, hist


def fgmres(m=30):
    V = [b / r0]; Z = []; H = np.zeros((m + 1, m)); gvec = np.zeros(m + 1); gvec[0] = r0; hist = []
    for j in range(m):
        zj = M(V[j]); Z.append(zj); w = A(zj)
        for i in range(j + 1):   # modified Gram-Schmidt (the device version would fuse the dots: classical, twice if needed)
            H[i, j] = dot(V[i], w); w = w - H[i, j] * V[i]
        H[j + 1, j] = nrm(w); V.append(w / H[j + 1, j])
        y, *_ = np.linalg.lstsq(H[: j + 2, : j + 1], gvec[: j + 2], rcond=None)
        hist.append(float(np.linalg.norm(H[: j + 2, : j + 1] @ y - gvec[: j + 2])) / r0)
        if hist[-1] <= rtol:
            xx = sum(float(yk) * zk for yk, zk in zip(y, Z))
            return j + 1, hist, nrm(b - A(xx)) / r0
    return m, hist, None


def gcr(m=30):
    """GCR / flexible: minimal residual over span{Z}, orthogonalising A z against the earlier A z (same iterates as FGMRES in
    exact arithmetic; x and r updated every step -- no basis of the Krylov space besides C = A Z)"""
    xx = torch.zeros_like(b); r = b.clone(); Cs = []; Zs = []; hist = []
    for j in range(m):
        zj = M(r); c = A(zj)
        for ci, zi in zip(Cs, Zs):
            h = dot(ci, c); c = c - h * ci; zj = zj - h * zi
        cn = nrm(c); c = c / cn; zj = zj / cn
        a = dot(c, r); xx += a * zj; r = r - a * c
        Cs.append(c); Zs.append(zj)
        hist.append(nrm(r) / r0)
        if hist[-1] <= rtol:
            return j + 1, hist, nrm(b - A(xx)) / r0
    return m, hist, None


a, h = fbcgs()
print(f"flexible BiCGStab (torch): {a} applications; residuals after each half step: {[f'{v:.1e}' for v in h]}")
a, h, tr = fgmres()
print(f"FGMRES: {a} applications; residual estimates: {[f'{v:.1e}' for v in h]}; true residual of the result {tr:.2e}")
a, h, tr = gcr()
print(f"GCR on M^-1 r: {a} applications; residuals: {[f'{v:.1e}' for v in h]}; true {tr:.2e}")
