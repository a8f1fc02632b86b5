"""Offline (CPU, scipy) study beside mg_study.py: a coarse correction over LARGE aggregates of columns (down to one coarse unknown per
level and stream: the horizontally constant modes) inside M^-1 -- before the passes (they continue from P e_c), behind them on the
defect, or added.  Result in profiles/NEGATIVE_RESULTS.md (round 5).   python tests/studies/deflate_study.py"""
import os, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NX","48"); os.environ.setdefault("NY","48"); os.environ.setdefault("NZ","24")
import mg_study as M
fine, b = M.build()
n = fine.n; D = 10; L = fine.k.max() + 1
print("n", n, flush=True)
def coarse_space(bx, by):
    # aggregates of bx x by owner columns (bx = nx: the whole row)
    cx, cy = fine.nx // bx, fine.ny // by
    agg = ((fine.oj // by) * cx + (fine.oi // bx)) * (L * D) + fine.k * D + fine.d
    nc = cx * cy * L * D
    P = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, nc))
    Ac = (P.T @ fine.A @ P).tocsc()
    return P, spla.splu(Ac), nc
for npass in (22, 28):
    its, hist = M.fbcgs(fine.A, b, lambda v: fine.passes(v, npass))
    print(f"passes {npass}: {its} its last {hist[-1]:.1e}", flush=True)
for (bx, by) in ((fine.nx, fine.ny), (fine.nx // 2, fine.ny // 2), (fine.nx // 4, fine.ny // 4), (6, 6), (4, 4)):
    P, lu, nc = coarse_space(bx, by)
    for npass in (14, 22, 28):
        def add(v): return fine.passes(v, npass) + P @ lu.solve(P.T @ v)
        def first(v):
            x0 = P @ lu.solve(P.T @ v)
            return fine.passes(v, npass, x=x0)
        def last(v):
            x = fine.passes(v, npass)
            r = v - fine.A @ x
            return x + P @ lu.solve(P.T @ r)
        for name, f in (("additive", add), ("coarse first, passes continue", first), ("passes, then coarse on the defect", last)):
            its, hist = M.fbcgs(fine.A, b, f)
            print(f"aggregates {bx}x{by} (coarse n {nc}) {npass} passes {name:34s}: {its} its last {hist[-1]:.1e}", flush=True)
