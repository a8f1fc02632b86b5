"""Offline (CPU, scipy) study of column-block smoothers as preconditioners for flexible BiCGStab: iteration counts on a
small synthetic problem.  Variants differ only in the colouring / ordering of the column-block Gauss-Seidel."""
import sys, os
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from tenstream_amd import synthetic

Nx, Ny, Nz = int(os.environ.get("NX", 48)), int(os.environ.get("NY", 48)), int(os.environ.get("NZ", 24))
P = synthetic.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz)
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
owner = oj * Nx + oi  # the column whose cell the unknown leaves
Ac = A.tocoo()
same = owner[Ac.row] == owner[Ac.col]
M = sp.csc_matrix((Ac.data[same], (Ac.row[same], Ac.col[same])), shape=A.shape)
Noff = sp.csr_matrix((Ac.data[~same], (Ac.row[~same], Ac.col[~same])), shape=A.shape)
lu = spla.splu(M, permc_spec="NATURAL")
b = P["b"].ravel()


def gs_passes(colour_of, order, lag=None):
    """block GS over colour classes in the given order of (colour, ...) passes; all off-block couplings use the latest
    values (true GS between colours, Jacobi within a colour)"""
    def apply(v):
        x = np.zeros(n)
        for c in order:
            rhs = v - Noff @ x
            mk = colour_of == c
            x[mk] = lu.solve(rhs)[mk]
        return x
    return apply


def fbcgs(Minv, rtol=1e-5, maxit=200):
    x = np.zeros(n); r = b.copy(); rh = r.copy(); p = r.copy()
    rho = rh @ r; r0 = np.linalg.norm(r)
    for it in range(1, maxit + 1):
        ph = Minv(p); v = A @ ph; alpha = rho / (rh @ v)
        s = r - alpha * v; sh = Minv(s); t = A @ sh
        omega = (t @ s) / (t @ t)
        x += alpha * ph + omega * sh; r = s - omega * t
        if np.linalg.norm(r) / r0 <= rtol:
            return it
        rho_new = rh @ r; beta = (rho_new / rho) * (alpha / omega); rho = rho_new
        p = r + beta * (p - omega * v)
    return maxit


zeb = oj % 2
rb = (oi + oj) % 2
c4 = (oi % 2) + 2 * (oj % 2)
print("problem", Nx, Ny, Nz, "n", n)
VARIANTS = os.environ.get("VARIANTS", "")
for name, col, order in (
    ("jacobi (1 pass)", np.zeros(n, int), [0]),
    ("zebra-y 2 passes", zeb, [0, 1]),
    ("zebra-y 4 passes", zeb, [0, 1, 0, 1]),
    ("zebra-y 6 passes", zeb, [0, 1, 0, 1, 0, 1]),
    ("red-black 2 passes", rb, [0, 1]),
    ("red-black 4 passes", rb, [0, 1, 0, 1]),
    ("red-black 6 passes", rb, [0, 1, 0, 1, 0, 1]),
    ("4-colour 4 passes", c4, [0, 1, 2, 3]),
    ("4-colour 8 passes", c4, [0, 1, 2, 3, 0, 1, 2, 3]),
):
    if VARIANTS and not any(v in name for v in VARIANTS.split(",")):
        continue
    print(f"{name:22s} its {fbcgs(gs_passes(col, order))}", flush=True)


# ---- which outer iteration makes best use of one preconditioner application? ---------------------------------
def richardson(Minv, rtol=1e-5, maxit=200):
    x = np.zeros(n); r = b.copy(); r0 = np.linalg.norm(r)
    for it in range(1, maxit + 1):
        x += Minv(r); r = b - A @ x
        if np.linalg.norm(r) / r0 <= rtol:
            return it
    return maxit


def fgmres(Minv, rtol=1e-5, maxit=200, m=30):
    x = np.zeros(n); r = b.copy(); r0 = np.linalg.norm(r); apps = 0
    while apps < maxit:
        beta = np.linalg.norm(r); V = [r / beta]; Z = []; H = np.zeros((m + 1, m)); g = np.zeros(m + 1); g[0] = beta
        for jj in range(m):
            z = Minv(V[jj]); apps += 1; Z.append(z); w = A @ z
            for ii in range(jj + 1):
                H[ii, jj] = V[ii] @ w; w -= H[ii, jj] * V[ii]
            H[jj + 1, jj] = np.linalg.norm(w); V.append(w / H[jj + 1, jj])
            y, res, *_ = np.linalg.lstsq(H[: jj + 2, : jj + 1], g[: jj + 2], rcond=None)
            rn = np.linalg.norm(H[: jj + 2, : jj + 1] @ y - g[: jj + 2])
            if rn / r0 <= rtol or jj == m - 1:
                x += sum(yk * zk for yk, zk in zip(y, Z)); r = b - A @ x
                break
        if np.linalg.norm(r) / r0 <= rtol:
            return apps
    return apps


if os.environ.get("OUTER"):
    print("outer iterations with red-black passes as M^-1 (preconditioner applications to rtol 1e-5):")
    for P_ in (10, 22):
        Minv = gs_passes(rb, [q % 2 for q in range(P_)])
        print(f"  {P_:2d} passes: BiCGStab {2 * fbcgs(Minv):3d} apps | FGMRES {fgmres(Minv):3d} apps | Richardson {richardson(Minv):3d} apps", flush=True)
