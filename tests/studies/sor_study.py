"""Offline (CPU, scipy) study: over-relaxed red-black passes (SOR on the column blocks) as the preconditioner of flexible
BiCGStab -- iteration counts against the number of half-grid passes and omega."""
import sys, os
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from tenstream_amd import synthetic as S

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
Noff = sp.csr_matrix((Ac.data[~same], (Ac.row[~same], Ac.col[~same])), shape=A.shape)
lu = spla.splu(M, permc_spec="NATURAL")
b = P["b"].ravel()
rb = (oi + oj) % 2


side = d >= 2
SIDE_ONLY = os.environ.get("SIDE_ONLY", "1") == "1"   # the top streams are slaved to the side inflows: never relaxed


def sor_passes(npass, omegas):
    def apply(v):
        x = np.zeros(n)
        for p in range(npass):
            rhs = v - Noff @ x
            mk = rb == (p % 2)
            xg = lu.solve(rhs)
            w = omegas[p]
            if SIDE_ONLY:
                ms, mt = mk & side, mk & ~side
                x[ms] = x[ms] + w * (xg[ms] - x[ms])
                x[mt] = xg[mt]
            else:
                x[mk] = x[mk] + w * (xg[mk] - x[mk])
        return x
    return apply


def fbcgs(Minv, rtol=1e-5, maxit=200):
    x = np.zeros(n); r = b.copy(); rh = r.copy(); p = r.copy()
    rho = rh @ r; r0 = np.linalg.norm(r)
    for it in range(1, maxit + 1):
        ph = Minv(p); v = A @ ph; alpha = rho / (rh @ v)
        s = r - alpha * v
        if np.linalg.norm(s) / r0 <= rtol:
            return it - 0.5
        sh = Minv(s); t = A @ sh
        omega = (t @ s) / (t @ t)
        x += alpha * ph + omega * sh; r = s - omega * t
        if np.linalg.norm(r) / r0 <= rtol:
            return it
        rho_new = rh @ r; beta = (rho_new / rho) * (alpha / omega); rho = rho_new
        p = r + beta * (p - omega * v)
    return maxit


print("problem", Nx, Ny, Nz, "n", n, flush=True)
TAIL = int(os.environ.get("TAIL", "0"))  # that many last passes plain (omega = 1)
for npass in [int(v) for v in os.environ.get("PASSES", "6,8,10").split(",")]:
    row = []
    for w in [float(v) for v in os.environ.get("OMEGAS", "1.0,1.1,1.2,1.3").split(",")]:
        om = [1.0, 1.0] + [w] * (npass - 2 - TAIL) + [1.0] * TAIL   # each colour's first pass starts from zero: plain
        row.append((w, fbcgs(sor_passes(npass, om))))
    print(f"{npass:2d} passes:", "  ".join(f"w={w:.1f}: {it}" for w, it in row), flush=True)
