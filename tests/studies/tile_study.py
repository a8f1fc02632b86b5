"""Offline (CPU, scipy) study: would a pass over TILES of columns held on chip pay?

Today one launch is one half-grid pass of red-black Gauss-Seidel over exact column-block solves; every pass streams the
per-cell words (right-hand side, record index, neighbour iterates) from HBM and moves information by one column.  A
workgroup that keeps a TX x TY tile of columns (both colours) in LDS could run K full red-black sweeps on it between two
trips to HBM, with the iterates of columns outside the tile frozen at what they were when the launch began (Jacobi across
tiles).  Modelled here exactly (same blocks, fp64): flexible BiCGStab around
    M^-1 = OUTER launches x K inner sweeps over TX x TY tiles   [tiles shifted by half a tile on odd launches if SHIFT=1]
Reported: applications of M^-1 to rtol 1e-5 (with the half step), launches and inner half-passes in total.

    NX=48 NY=48 NZ=24 python tests/studies/tile_study.py
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O  # noqa: E402
from tenstream_amd import synthetic as S  # noqa: E402

Nx, Ny, Nz = int(os.environ.get("NX", 48)), int(os.environ.get("NY", 48)), int(os.environ.get("NZ", 24))
P = S.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz)
lay = O.layout("3_10", Nz, Nx, Ny)
A = O.assemble_csr(lay, P["coeff"].astype(np.float64), P["l1d"], P["a11"], P["a12"], P["albedo"]).tocsr()
n = A.shape[0]
D, L = 10, Nz + 1
idx = np.arange(n)
d, k = idx % D, (idx // D) % L
i, j = (idx // (D * L)) % Nx, idx // (D * L * Nx)
oi, oj = i.copy(), j.copy()
qx, qy = d - 2, d - 6
mx = (qx >= 0) & (qx < 4) & (qx % 2 == 1) & (k < Nz)
my = (qy >= 0) & (qy < 4) & (qy % 2 == 1) & (k < Nz)
oi[mx] = (i[mx] - 1) % Nx
oj[my] = (j[my] - 1) % Ny
owner = oj * Nx + oi
Ac = A.tocoo()
same = owner[Ac.row] == owner[Ac.col]
M = sp.csc_matrix((Ac.data[same], (Ac.row[same], Ac.col[same])), shape=A.shape)
lu = spla.splu(M, permc_spec="NATURAL")
b = P["b"].ravel()
rb = (oi + oj) % 2
off_r, off_c, off_v = Ac.row[~same], Ac.col[~same], Ac.data[~same]
Noff = sp.csr_matrix((off_v, (off_r, off_c)), shape=A.shape)


def split(tx, ty, sx, sy):
    """couplings between columns of one tile / of different tiles, tiles of tx x ty columns with origin (sx, sy)"""
    tile = (((oj - sy) % Ny) // ty) * ((Nx + tx - 1) // tx) + ((oi - sx) % Nx) // tx
    inside = tile[off_r] == tile[off_c]
    mk = lambda m: sp.csr_matrix((off_v[m], (off_r[m], off_c[m])), shape=A.shape)
    return mk(inside), mk(~inside)


def plain(npass):
    def apply(v):
        x = np.zeros(n)
        for p in range(npass):
            xg = lu.solve(v - Noff @ x)
            mk = rb == (p % 2)
            x[mk] = xg[mk]
        return x
    apply.launches, apply.inner = npass, npass
    return apply


def tiled(outer, K, tx, ty, shift):
    splits = [split(tx, ty, 0, 0)] + ([split(tx, ty, tx // 2, ty // 2)] if shift else [])

    def apply(v):
        x = np.zeros(n)
        for o in range(outer):
            Nin, Nout = splits[o % len(splits)]
            frozen = v - Nout @ x
            for p in range(2 * K):
                xg = lu.solve(frozen - Nin @ x)
                mk = rb == (p % 2)
                x[mk] = xg[mk]
        return x
    apply.launches, apply.inner = outer, outer * 2 * K
    return apply


def fbcgs(Minv, rtol=1e-5, maxit=60):
    x = np.zeros(n); r = b.copy(); rh = r.copy(); p = r.copy()
    rho = rh @ r; r0 = np.linalg.norm(r); apps = 0
    for it in range(1, maxit + 1):
        ph = Minv(p); apps += 1; v = A @ ph; alpha = rho / (rh @ v)
        s = r - alpha * v
        if np.linalg.norm(s) / r0 <= rtol:
            return apps, np.linalg.norm(s) / r0
        sh = Minv(s); apps += 1; t = A @ sh
        omega = (t @ s) / (t @ t)
        x += alpha * ph + omega * sh; r = s - omega * t
        if np.linalg.norm(r) / r0 <= rtol:
            return apps, np.linalg.norm(r) / r0
        rho_new = rh @ r; beta = (rho_new / rho) * (alpha / omega); rho = rho_new
        p = r + beta * (p - omega * v)
    return apps, np.linalg.norm(r) / r0


print(f"problem {Nx}x{Ny}x{Nz}, n = {n}", flush=True)
cases = [("plain 28 half passes", plain(28)), ("plain 14", plain(14))]
for spec in os.environ.get("CASES", "8x8:4:2:1,8x8:4:3:1,8x8:6:2:1,8x8:8:2:1,8x8:6:2:0,4x8:6:2:1,16x16:4:4:1,8x8:10:1:1").split(","):
    t, o, K, sh = spec.split(":")
    tx, ty = (int(c) for c in t.split("x"))
    cases.append((f"tiles {tx}x{ty}, {o} launches x {K} sweeps" + (", shifted" if sh == "1" else ""), tiled(int(o), int(K), tx, ty, sh == "1")))
for name, Mi in cases:
    apps, rel = fbcgs(Mi)
    print(f"  {name:44s}: {apps:2d} applications, {apps * Mi.launches:4d} launches, {apps * Mi.inner:4d} inner half passes, last {rel:.1e}", flush=True)
