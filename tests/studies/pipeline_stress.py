"""Ad-hoc stress (GPU box): the g-point pipeline against the oracle on random small domains, sun positions and solvers, with the
round-3 kernels switched on and off (tiled direct sweep, cell-order LUT coordinates, shared blocks).  usage: python
tests/studies/pipeline_stress.py [n_cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_gpu_pipeline as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "2026")))
worst = 0.0
for case in range(n):
    solver = "3_10" if rng.random() < 0.7 else "8_16"
    Nx, Ny, Nz = int(rng.integers(3, 28)), int(rng.integers(3, 22)), int(rng.integers(2, 18))
    phi0, theta0 = float(rng.uniform(0, 360)), float(rng.uniform(0, 75))
    tall = int(rng.integers(0, min(3, Nz)))
    lsolar = rng.random() < 0.6
    env = {"TSX_EDIR_TILED": str(int(rng.integers(0, 2))), "TSX_CELL_SAMPLES": str(int(rng.integers(0, 2))),
           "TSX_DEDUP": str(int(rng.integers(0, 2))), "TSX_DEDUP_COORDS": str(int(rng.integers(0, 2))),
           "TSX_HALF_EXIT": str(int(rng.integers(0, 2)))}   # round 4: coordinates-first sharing, half-step stop test
    os.environ.update(env)
    P, I = T._setup(Nx, Ny, Nz, phi0, theta0, tall, solver=solver)
    if lsolar:
        P.set_optical_properties(0.15, I["kabs"], I["ksca"], I["g"], I["dz"])
        info = P.solve(1000.0, rtol=1e-10, atol=1e-30, maxit=3000)
        R = T._oracle_pipeline(P, I, 0.15, 1000.0, True)
    else:
        planck = np.linspace(2.0, 6.0, Nz + 1)[None, None, :] * np.ones((Ny, Nx, 1)) * (1 + 0.05 * rng.random((Ny, Nx, 1)))
        srfc = planck[:, :, -1] * (1.0 + 0.3 * rng.random((Ny, Nx))) if rng.random() < 0.5 else None   # atm%Bsrfc (round 4)
        P.set_optical_properties(0.05, I["kabs"], I["ksca"], I["g"], I["dz"], planck=planck, planck_srfc=srfc)
        info = P.solve(0.0, rtol=1e-10, atol=1e-30, maxit=3000)
        R = T._oracle_pipeline(P, I, 0.05, 0.0, False, planck=planck)
    edn, eup, abso, edir = P.get_result()
    errs = [np.abs(got - want).max() / max(np.abs(want).max(), 1e-30) for got, want in ((edn, R["edn"]), (eup, R["eup"]), (abso, R["abso"]))]
    if lsolar:
        errs.append(np.abs(edir - R["redir"]).max() / np.abs(R["redir"]).max())
    worst = max(worst, max(errs))
    print(f"{case:3d} {solver} {Nx:2d}x{Ny:2d}x{Nz:2d} phi {phi0:5.1f} theta {theta0:4.1f} 1d {tall} {'solar' if lsolar else 'thermal'} {env} "
          f"reason {info.reason} max rel err {max(errs):.2e}", flush=True)
    assert info.reason == 2 and max(errs) <= 2e-4, (case, errs)
print("worst", worst)
