"""Ad-hoc stress (GPU box): one rank with itself as its four neighbours over the peer transport (every stage of the passes' exchange,
both workgroup widths of the sending pass) against the plain periodic domain: same iteration count, same solution.
usage: python tests/studies/peer_stress.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tenstream_amd import DiffuseSolver, synthetic  # noqa: E402

os.environ.setdefault("TSX_PEER_TIMEOUT_S", "10")
cases = [("3_10", 12, 10, 8), ("3_10", 40, 26, 20), ("3_10", 64, 64, 32), ("3_10", 128, 64, 64), ("3_10", 128, 128, 16),
         ("3_10", 256, 128, 8), ("8_16", 16, 12, 10), ("8_16", 64, 32, 16), ("3_10", 30, 22, 70), ("3_10", 18, 14, 130)]
for solver, Nx, Ny, Nz in cases:
    P = synthetic.make_problem(solver, Nx=Nx, Ny=Ny, Nz=Nz, n1d=1 if Nz > 8 else 0)
    ref = None
    for mode in ("wrap", "0", "1", "2"):
        if mode != "wrap":
            os.environ["TSX_PEER_INPLACE"] = mode
        s = DiffuseSolver(solver, Nz, Nx, Ny, force_halo=mode != "wrap")
        if mode != "wrap":
            s.comm_peer_init(lambda blob: [blob])
        s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
        x = np.zeros(s.vec_shape)
        info = s.solve(P["b"], x, rtol=1e-10, atol=1e-30)
        s.close()
        if ref is None:
            ref = (x, info)
            continue
        err = np.abs(x - ref[0]).max() / np.abs(ref[0]).max()
        print(f"{solver} {Nx}x{Ny}x{Nz} inplace {mode}: its {info.niter} (periodic {ref[1].niter}) reason {info.reason} max rel diff {err:.2e}", flush=True)
        assert info.reason == 2 and info.niter == ref[1].niter and err < 1e-7, (solver, Nx, Ny, Nz, mode)
print("ok")
