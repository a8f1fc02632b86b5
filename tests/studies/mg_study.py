"""Offline (CPU, scipy) study: does a coarse-grid correction over 2 x 2 column aggregates pay as part of the preconditioner?

The device's M^-1 is P half-grid passes of red-black Gauss-Seidel over exact column-block solves; iterations x passes
stays near 110 whatever the split, i.e. the passes behave like a stationary smoother whose slow modes are horizontally smooth.
Here: the same smoother (exact blocks, fp64) wrapped into a two-grid / V-cycle with plain aggregation
    P = indicator of the aggregate (same stream, same level, owner column in the 2 x 2 block),  R = P^T,  A_c = R A P
and flexible BiCGStab around it.  Reported: preconditioner applications to rtol 1e-5 and the work in units of one
half-grid fine pass (a fine operator apply counted as `SPMV` passes, the measured ratio on the device is about 4.3).

    NX=64 NY=64 NZ=32 python tests/studies/mg_study.py
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O  # noqa: E402
from tenstream_amd import synthetic  # noqa: E402

Nx, Ny, Nz = int(os.environ.get("NX", 64)), int(os.environ.get("NY", 64)), int(os.environ.get("NZ", 32))
SPMV = float(os.environ.get("SPMV", 4.3))
D = 10


class Level:
    def __init__(self, A, d, k, oi, oj, nx, ny):
        self.A = A.tocsr()
        self.n = A.shape[0]
        self.d, self.k, self.oi, self.oj, self.nx, self.ny = d, k, oi, oj, nx, ny
        owner = oj * nx + oi
        Ac = self.A.tocoo()
        same = owner[Ac.row] == owner[Ac.col]
        M = sp.csc_matrix((Ac.data[same], (Ac.row[same], Ac.col[same])), shape=A.shape)
        self.Noff = sp.csr_matrix((Ac.data[~same], (Ac.row[~same], Ac.col[~same])), shape=A.shape)
        self.lu = spla.splu(M, permc_spec="NATURAL")
        self.rb = (oi + oj) % 2
        self.cells = nx * ny

    def passes(self, v, npass, x=None, first_colour=0):
        """npass half-grid passes; x = None: from a zero iterate"""
        x = np.zeros(self.n) if x is None else x.copy()
        for p in range(npass):
            c = (first_colour + p) % 2
            rhs = v - self.Noff @ x
            mk = self.rb == c
            x[mk] = self.lu.solve(rhs)[mk]
        return x

    def coarsen(self):
        """2 x 2 aggregation of owner columns; returns (P, coarse Level)"""
        cx, cy = self.nx // 2, self.ny // 2
        L = self.k.max() + 1
        agg = ((self.oj // 2) * cx + (self.oi // 2)) * (L * D) + self.k * D + self.d
        nc = cx * cy * L * D
        P = sp.csr_matrix((np.ones(self.n), (np.arange(self.n), agg)), shape=(self.n, nc))
        Ac = (P.T @ self.A @ P).tocsr()
        idx = np.arange(nc)
        d, k = idx % D, (idx // D) % L
        col = idx // (D * L)
        return P, Level(Ac, d, k, col % cx, col // cx, cx, cy)


def build():
    Pm = synthetic.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz)
    lay = O.layout("3_10", Nz, Nx, Ny)
    A = O.assemble_csr(lay, Pm["coeff"].astype(np.float64), Pm["l1d"], Pm["a11"], Pm["a12"], Pm["albedo"]).tocsr()
    n = A.shape[0]
    L = Nz + 1
    idx = np.arange(n)
    d, k = idx % D, (idx // D) % L
    i, j = (idx // (D * L)) % Nx, idx // (D * L * Nx)
    oi, oj = i.copy(), j.copy()
    qx, qy = d - 2, d - 6
    mx = (qx >= 0) & (qx < 4) & (qx % 2 == 1) & (k < Nz)
    my = (qy >= 0) & (qy < 4) & (qy % 2 == 1) & (k < Nz)
    oi[mx] = (i[mx] - 1) % Nx
    oj[my] = (j[my] - 1) % Ny
    return Level(A, d, k, oi, oj, Nx, Ny), Pm["b"].ravel()


def fbcgs(A, b, Minv, rtol=1e-5, maxit=100):
    n = len(b)
    x = np.zeros(n); r = b.copy(); rh = r.copy(); p = r.copy()
    rho = rh @ r; r0 = np.linalg.norm(r)
    hist = []
    for it in range(1, maxit + 1):
        ph = Minv(p); v = A @ ph; alpha = rho / (rh @ v)
        s = r - alpha * v
        if np.linalg.norm(s) / r0 <= rtol:   # half iteration
            hist.append(np.linalg.norm(s) / r0)
            return it - 0.5, hist
        sh = Minv(s); t = A @ sh
        omega = (t @ s) / (t @ t)
        x += alpha * ph + omega * sh; r = s - omega * t
        hist.append(np.linalg.norm(r) / r0)
        if hist[-1] <= rtol:
            return it, hist
        rho_new = rh @ r; beta = (rho_new / rho) * (alpha / omega); rho = rho_new
        p = r + beta * (p - omega * v)
    return maxit, hist


class Cycle:
    """V-cycle as a preconditioner: pre passes from zero, coarse correction of the defect, post passes continuing from the
    corrected iterate.  work: in fine half-grid passes (level l costs 4^-l per pass and per SPMV-equivalent)"""

    def __init__(self, levels, Ps, pre, post, coarse_passes, omega=1.0, exact_coarsest=False):
        self.levels, self.Ps, self.pre, self.post, self.cp, self.omega, self.exact = levels, Ps, pre, post, coarse_passes, omega, exact_coarsest
        self.lu_c = spla.splu(levels[-1].A.tocsc()) if exact_coarsest else None
        self.work = 0.0

    def apply(self, v, l=0):
        lev = self.levels[l]
        scale = 0.25 ** l
        if l == len(self.levels) - 1:
            if self.exact:
                return self.lu_c.solve(v)
            self.work += self.cp * scale
            return lev.passes(v, self.cp)
        x = lev.passes(v, self.pre)
        self.work += self.pre * scale
        r = v - lev.A @ x
        self.work += SPMV * scale
        ec = self.apply(self.Ps[l].T @ r, l + 1)
        x = x + self.omega * (self.Ps[l] @ ec)
        if self.post:
            # continuing the passes from x: the smoother solves for colour c with the other colour's latest values
            x = lev.passes(v, self.post, x=x, first_colour=self.pre % 2)
            self.work += self.post * scale
        return x

    def __call__(self, v):
        return self.apply(v)


def main():
    t0 = time.time()
    fine, b = build()
    print(f"problem {Nx}x{Ny}x{Nz}, n = {fine.n}, setup {time.time() - t0:.1f}s", flush=True)
    levels, Ps = [fine], []
    for _ in range(int(os.environ.get("LEVELS", 3)) - 1):
        if levels[-1].nx < 4 or levels[-1].nx % 2 or levels[-1].ny % 2:
            break
        P, c = levels[-1].coarsen()
        Ps.append(P)
        levels.append(c)
        print(f"  level {len(levels) - 1}: {c.nx}x{c.ny} columns, n = {c.n}, nnz/row {c.A.nnz / c.n:.1f}", flush=True)

    print("baseline: red-black passes alone")
    for npass in (6, 10, 14, 22):
        its, hist = fbcgs(fine.A, b, lambda v: fine.passes(v, npass))
        apps = int(2 * its)
        print(f"  {npass:2d} passes: {its:4.1f} its, work {apps * npass + apps * SPMV:7.1f} (passes {apps * npass}), last {hist[-1]:.1e}", flush=True)

    print("cycles (pre, post, coarse passes | levels | omega):")
    for nl in (2, 3):
        if nl > len(levels):
            continue
        for pre, post, cp, om, exact in ((2, 2, 0, 1.0, True), (2, 2, 8, 1.0, False), (4, 2, 8, 1.0, False), (2, 2, 8, 1.5, False),
                                         (2, 2, 8, 2.0, False), (4, 4, 12, 1.5, False), (2, 0, 8, 1.5, False), (6, 4, 16, 1.5, False)):
            cyc = Cycle(levels[:nl], Ps[: nl - 1], pre, post, cp, om, exact)
            its, hist = fbcgs(fine.A, b, cyc)
            apps = int(2 * its)
            tag = "exact coarsest" if exact else f"{cp} coarsest passes"
            print(f"  V({pre},{post}) {tag:18s} | {nl} levels | omega {om:3.1f}: {its:4.1f} its, work {cyc.work + apps * SPMV:7.1f} "
                  f"(cycle {cyc.work / max(apps, 1):5.1f}), last {hist[-1]:.1e}", flush=True)


if __name__ == "__main__":
    main()
