"""Flexible BiCGStab accepts a different M^-1 in every application: does a pass count that varies over the solve reach the stop
rule with fewer passes in total?  The library's operator and preconditioner on device tensors, the outer iteration in torch
(as scripts/fgmres_probe.py).  usage (GPU box): python scripts/sched_probe.py [nx ny [field]]"""
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
kabs, ksca, g = synthetic.cloud_field(nx, ny, nz, seed=int(os.environ.get("SEED", 20240611)), cover=float(os.environ.get("COVER", 0.3)),
                                      heterogeneous=field == "heterogeneous")
kabs, ksca, g = synthetic.delta_scale(kabs, ksca, g)
b = torch.tensor(synthetic.solar_source(solver, kabs, ksca, g, 50.0, 100.0, np.full((ny, nx), 0.1)), device=dev)
s = DiffuseSolver(solver, nz, nx, ny)
s.set_lut_diffuse(lut.synthetic_diffuse_table(solver), lut.diffuse_axes(solver))
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
z = torch.zeros((ny, nx, nz), dtype=torch.float64, device=dev)
s.set_optprop(t(kabs), t(ksca), t(g), torch.full((ny, nx, nz), 50.0, dtype=torch.float64, device=dev), 100.0,
              torch.zeros(nz, dtype=torch.uint8, device=dev), z, z, torch.full((ny, nx), 0.1, dtype=torch.float64, device=dev))
A = lambda v: s.apply(v)
dot = lambda a, c: float((a * c).sum())
nrm = lambda a: float(torch.linalg.vector_norm(a))
r0 = nrm(b)
rtol = 1e-5


def fbcgs(sched):
    """sched(k, rel) -> passes of application k (0-based) given the last known relative residual"""
    xx = torch.zeros_like(b); r = b.clone(); rh = r.clone(); p = r.clone(); rho = dot(rh, r); apps = 0; used = []; hist = []
    rel = 1.0
    for it in range(40):
        P = sched(apps, rel); used.append(P); ph = s.pc_apply(p, pc=3, sweeps=P - 1, mixed=True); apps += 1
        v = A(ph); alpha = rho / dot(rh, v); sres = r - alpha * v; rel = nrm(sres) / r0; hist.append(rel)
        if rel <= 0.9 * rtol:
            return used, hist
        P = sched(apps, rel); used.append(P); sh = s.pc_apply(sres, pc=3, sweeps=P - 1, mixed=True); apps += 1
        tt = A(sh); omega = dot(tt, sres) / dot(tt, tt)
        xx += alpha * ph + omega * sh; r = sres - omega * tt; rel = nrm(r) / r0; hist.append(rel)
        if rel <= rtol:
            return used, hist
        rho_new = dot(rh, r); beta = (rho_new / rho) * (alpha / omega); rho = rho_new
        p = r + beta * (p - omega * v)
    return used, hist


def const(P):
    return lambda k, rel: P


def taper(P, last, thresh):
    """P passes until the residual is within `thresh` of the rule, then `last`"""
    return lambda k, rel: last if rel <= thresh * rtol else P


def ramp(lst):
    return lambda k, rel: lst[min(k, len(lst) - 1)]


cases = [("28", const(28)), ("22", const(22)), ("32", const(32)),
         ("28 then 20 within 100x", taper(28, 20, 100.0)), ("28 then 16 within 30x", taper(28, 16, 30.0)), ("28 then 22 within 1000x", taper(28, 22, 1000.0)),
         ("32 then 20 within 100x", taper(32, 20, 100.0)), ("32 then 24 within 300x", taper(32, 24, 300.0)),
         ("ramp 20,24,28,32,32,..", ramp([20, 24, 28, 32, 32, 32, 32, 32])), ("ramp 32,32,28,28,24,24,20,20", ramp([32, 32, 28, 28, 24, 24, 20, 20, 20, 20])),
         ("ramp 24,24,28,28,32,32", ramp([24, 24, 28, 28, 32, 32, 32, 32]))]
print(f"{solver} {nx}x{ny}x{nz} {field}: passes per application -> total passes + 13 per application (operator + vector updates in pass units)")
for name, sc in cases:
    used, hist = fbcgs(sc)
    print(f"  {name:32s} apps {len(used):2d} passes {sum(used):3d} cost {sum(used) + 13 * len(used):4d}  {used}  last {hist[-1]:.1e}", flush=True)
