"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on identical inputs.

Bar: the operator is fp64 arithmetic on identical inputs -> <= 1e-13 relative to max|y| (sum order
differs from the reference's scatter form, so not bit-exact); converged solutions within the solver
tolerance (rtol 1e-10 runs: <= 1e-8 relative).
"""
import numpy as np
import pytest

from oracle import oracle as O
from tenstream_amd import DiffuseSolver, synthetic

pytestmark = pytest.mark.gpu

CASES = [
    ("3_10", 12, 10, 8, 0),
    ("3_10", 7, 5, 3, 2),      # ragged, with 1-D layers
    ("3_10", 3, 3, 1, 0),      # minimal_dimension = 3 (src/pprts.F90:205), single layer
    ("3_10", 64, 4, 20, 5),
    ("3_10", 67, 33, 9, 0),    # not a multiple of the wave size
    ("8_16", 9, 6, 5, 1),
    ("8_16", 32, 8, 4, 0),
]


def _ref_apply(P, lay, x):
    c64 = P["coeff"].astype(np.float64)
    if P["solver"] == "3_10":
        return O.diff_apply(lay, c64, P["l1d"], P["a11"], P["a12"], P["albedo"], x)
    # 8_16: assembled semantics (albedo/streams on every pair) is the parity target, SURVEY a5
    A = O.assemble_csr(lay, c64, P["l1d"], P["a11"], P["a12"], P["albedo"])
    return (A @ x.ravel()).reshape(x.shape)


@pytest.mark.parametrize("solver,Nx,Ny,Nz,n1d", CASES)
@pytest.mark.parametrize("force_halo", [False, True])
def test_apply_matches_oracle(gpu, solver, Nx, Ny, Nz, n1d, force_halo):
    P = synthetic.make_problem(solver, Nx=Nx, Ny=Ny, Nz=Nz, n1d=n1d, seed=Nx * 100 + Nz)
    s = DiffuseSolver(solver, Nz, Nx, Ny, force_halo=force_halo)
    s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    lay = O.layout(solver, Nz, Nx, Ny)
    rng = np.random.default_rng(1)
    x = rng.standard_normal(s.vec_shape)
    y = s.apply(x)
    y_ref = _ref_apply(P, lay, x)
    assert np.abs(y - y_ref).max() <= 1e-13 * np.abs(y_ref).max()
    s.close()


@pytest.mark.parametrize("solver,Nx,Ny,Nz,n1d", [("3_10", 20, 12, 9, 2), ("3_10", 3, 3, 4, 0), ("8_16", 6, 5, 4, 1)])
def test_apply_split_launches_equal_single_launch(gpu, monkeypatch, solver, Nx, Ny, Nz, n1d):
    """exchange/compute overlap: interior + frame launches (default) must give the bits of the one-launch SpMV"""
    P = synthetic.make_problem(solver, Nx=Nx, Ny=Ny, Nz=Nz, n1d=n1d, seed=5)
    x = np.random.default_rng(2).standard_normal((Ny, Nx, Nz + 1, 10 if solver == "3_10" else 16))
    ys = []
    for ov in ("1", "0"):
        monkeypatch.setenv("TSX_OVERLAP", ov)
        s = DiffuseSolver(solver, Nz, Nx, Ny, force_halo=True)
        s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
        assert x.shape == tuple(s.vec_shape)
        ys.append(s.apply(x))
        info_x = np.zeros_like(x)
        info = s.solve(P["b"], info_x, pc=1, pc_sweeps=1)
        ys.append(info_x)
        ys.append(info.res_hist)
        s.close()
    assert np.array_equal(ys[0], ys[3])
    np.testing.assert_allclose(ys[1], ys[4], rtol=1e-6, atol=1e-9 * np.abs(ys[4]).max())
    assert abs(len(ys[2]) - len(ys[5])) <= 1


def test_apply_fp64_coefficients_kept_when_lossy(gpu):
    """Blocks that are not fp32-representable must be stored as fp64 (results identical to the reference)."""
    P = synthetic.make_problem("3_10", Nx=8, Ny=6, Nz=5)
    rng = np.random.default_rng(3)
    c = P["coeff"].astype(np.float64) * (1.0 - 1e-9 * rng.random(P["coeff"].shape))
    s = DiffuseSolver("3_10", 5, 8, 6)
    s.set_coeffs(c, P["l1d"], P["a11"], P["a12"], P["albedo"])
    lay = O.layout("3_10", 5, 8, 6)
    x = rng.standard_normal(s.vec_shape)
    y = s.apply(x)
    y_ref = O.diff_apply(lay, c, P["l1d"], P["a11"], P["a12"], P["albedo"], x)
    assert np.abs(y - y_ref).max() <= 1e-13 * np.abs(y_ref).max()


@pytest.mark.parametrize("solver,Nx,Ny,Nz,n1d", [("3_10", 12, 10, 8, 2), ("3_10", 33, 17, 20, 0), ("8_16", 8, 6, 6, 0)])
@pytest.mark.parametrize("force_halo", [False, True])
@pytest.mark.parametrize("pc", [0, 2, 3])  # bare operator / zebra rows / red-black (default; falls back to zebra on odd grids)
def test_solve_matches_oracle(gpu, solver, Nx, Ny, Nz, n1d, force_halo, pc):
    P = synthetic.make_problem(solver, Nx=Nx, Ny=Ny, Nz=Nz, n1d=n1d)
    s = DiffuseSolver(solver, Nz, Nx, Ny, force_halo=force_halo)
    s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    lay = O.layout(solver, Nz, Nx, Ny)
    c64 = P["coeff"].astype(np.float64)
    x = np.zeros(s.vec_shape)
    info = s.solve(P["b"], x, rtol=1e-10, atol=1e-30, maxit=2000, pc=pc)
    assert info.reason == 2, info
    if solver == "3_10":
        x_ref, ri = O.solve_matfree(lay, c64, P["l1d"], P["a11"], P["a12"], P["albedo"], P["b"], rtol=1e-12,
                                    atol=1e-30, maxit=4000)
        assert ri["reason"] == 2
    else:
        import scipy.sparse.linalg as spla

        A = O.assemble_csr(lay, c64, P["l1d"], P["a11"], P["a12"], P["albedo"])
        x_ref = spla.spsolve(A.tocsc(), P["b"].ravel()).reshape(x.shape)
    assert np.abs(x - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
    # residual history: first entry is ||b - A x0|| = ||b||
    assert abs(info.res_hist[0] - np.linalg.norm(P["b"])) <= 1e-12 * np.linalg.norm(P["b"])


def test_initial_guess_zero_ignores_what_x_holds(gpu):
    """opts.initial_guess_zero = 1 (KSPSetInitialGuessNonzero(FALSE)): x is not read -- garbage on entry, the same iterates and
    solution as a solve from an explicit zero guess."""
    P = synthetic.make_problem("3_10", Nx=12, Ny=10, Nz=8, n1d=1)
    s = DiffuseSolver("3_10", 8, 12, 10)
    s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    x0 = np.zeros(s.vec_shape)
    i0 = s.solve(P["b"], x0, rtol=1e-10, atol=1e-30)
    x1 = np.full(s.vec_shape, np.nan)
    i1 = s.solve(P["b"], x1, rtol=1e-10, atol=1e-30, initial_guess_zero=1)
    assert i1.reason == 2 and i1.niter == i0.niter and np.array_equal(i0.res_hist, i1.res_hist) and np.array_equal(x0, x1)
    s.close()


def test_solve_default_tolerances_and_warm_start(gpu):
    """Reference stop rule (rtol 1e-5 / atol formula, src/pprts_base.F90:1126-1131) and nonzero initial guess."""
    P = synthetic.make_problem("3_10", Nx=16, Ny=16, Nz=16)
    s = DiffuseSolver("3_10", 16, 16, 16)
    s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    rt, at, mx = s.default_tolerances()
    assert rt == 1e-5 and mx == 1000 and at == pytest.approx(1e-4 * 16 * 16 * 17)
    x = np.zeros(s.vec_shape)
    info = s.solve(P["b"], x)
    assert info.reason in (2, 3)
    assert info.rnorm / info.rnorm0 <= 1e-5 or info.rnorm <= at
    # warm start (nonzero initial guess, src/pprts.F90:4343): the first residual of the second solve is the
    # last residual of the first one, and the stop rule is relative to *that* (MyKSPConverged n == 0)
    info2 = s.solve(P["b"], x, rtol=1e-3)
    assert info2.res_hist[0] == pytest.approx(info.rnorm, rel=1e-6)
    assert info2.niter < info.niter and info2.reason in (2, 3)


@pytest.mark.parametrize("solver,Nx,Ny,Nz,n1d,mixed", [("3_10", 12, 10, 8, 0, 1), ("3_10", 7, 5, 6, 2, 1), ("3_10", 16, 8, 20, 3, 0),
                                                        ("8_16", 8, 6, 5, 1, 1)])
def test_explicit_solver_shares_the_fixed_point_and_the_stop_rule_of_explicit_ediff(gpu, solver, Nx, Ny, Nz, n1d, mixed):
    """-<prefix>explicit (src/pprts.F90:2799): explicit_ediff (src/pprts_explicit.F90:461-713) sweeps until the 2-norm of the
    iterate's change is < atol or < rtol x the first iteration's.  The device's sweeps are the red-black column-block passes
    (a different sweep order than the reference's cell-by-cell SOR, so the iterates differ), the fixed point A x = b and the
    stop rule are the reference's: converged tightly it equals the oracle's explicit solver and the Krylov solve; the history
    holds the change norms and the rule is the strict '<' on them."""
    P = synthetic.make_problem(solver, Nx=Nx, Ny=Ny, Nz=Nz, n1d=n1d)
    lay = O.layout(solver, Nz, Nx, Ny)
    s = DiffuseSolver(solver, Nz, Nx, Ny)
    s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    x = np.zeros(s.vec_shape)
    info = s.solve(P["b"], x, explicit_solver=1, rtol=1e-11, atol=1e-30, maxit=2000, pc_sweeps=3, fp32_directions=mixed)
    assert info.reason == 2 and info.niter > 3
    h = info.res_hist
    assert info.rnorm0 == h[0] and len(h) == min(info.niter, 100)
    assert h[-1] / h[0] < 1e-11 if info.niter <= 100 else True
    if info.niter <= 100:   # strict '<' on the change norms: the iteration before the last did not satisfy it
        assert h[-2] / h[0] >= 1e-11
    assert np.all(np.diff(h[: min(len(h), 30)]) < 0)   # a contraction: the change shrinks monotonically
    xk = np.zeros(s.vec_shape)
    ik = s.solve(P["b"], xk, rtol=1e-12, atol=1e-30, maxit=3000)
    assert ik.reason == 2
    assert np.abs(x - xk).max() <= 1e-8 * np.abs(xk).max()
    if solver == "3_10":   # the oracle's restatement of the reference's explicit solver (3_10 sweep)
        c64 = P["coeff"].astype(np.float64)
        xo, io = O.solve_sor(lay, c64, P["l1d"], P["a11"], P["a12"], P["albedo"], P["b"], rtol=1e-11, atol=1e-30, maxit=10000)
        assert io["converged"]
        assert np.abs(x - xo).max() <= 1e-8 * np.abs(xo).max()
    # atol is tested first and reported as reason 3; maxit exhausted is -3 (CHKERR 'did not converge' in the reference)
    x2 = np.zeros(s.vec_shape)
    i2 = s.solve(P["b"], x2, explicit_solver=1, rtol=1e-30, atol=1e-3 * h[0], maxit=2000, pc_sweeps=3, fp32_directions=mixed)
    assert i2.reason == 3 and i2.res_hist[-1] < 1e-3 * h[0] <= i2.res_hist[-2]
    x3 = np.zeros(s.vec_shape)
    i3 = s.solve(P["b"], x3, explicit_solver=1, rtol=1e-30, atol=1e-300, maxit=3, pc_sweeps=3, fp32_directions=mixed)
    assert i3.reason == -3 and i3.niter == 3
    s.close()


@pytest.mark.parametrize("solver", ["3_10", "8_16"])
def test_repeated_coefficient_updates_and_solves_leak_no_device_memory(gpu, solver):
    """A spectral loop hands over new blocks and solves hundreds of times per call (src/pprts.F90:2281-2300 per g-point):
    50 rounds of host-path set_coeffs + solve on 64 x 64 x 32 leave the free device memory where it was after the first
    rounds (every per-call temporary, event and packed-block buffer is reused or released)."""
    import torch

    import ctypes

    def pool():   # round 6: device memory comes from the library's pool (tsx_pool.hip); [2] = bytes handed out now, [0] = driver allocations
        st = (ctypes.c_int64 * 8)()
        assert gpu.tsx_pool_stats(-1, st) == 0
        return [int(v) for v in st]

    live_before = pool()[2]
    P = synthetic.make_problem(solver, Nx=64, Ny=64, Nz=32, n1d=2)
    s = DiffuseSolver(solver, 32, 64, 64)
    x = np.zeros(s.vec_shape)

    def round_(q):
        c = P["coeff"] if q % 2 == 0 else P["coeff"] * np.float32(0.999)   # a new coefficient set every round
        s.set_coeffs(c, P["l1d"], P["a11"], P["a12"], P["albedo"])
        x[...] = 0
        info = s.solve(P["b"], x)
        assert info.reason in (2, 3)

    for q in range(3):   # first rounds allocate the persistent buffers (packed blocks, shared-block table, scratch)
        round_(q)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    slabs0, _, live0 = pool()[:3]
    for q in range(50):
        round_(q)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 <= (2 << 20), f"device memory shrank by {(free0 - free1) / 2**20:.1f} MiB over 50 rounds"
    slabs1, _, live1 = pool()[:3]
    assert slabs1 == slabs0, "a driver allocation on the per-coefficient-set path"   # (alloc_coeff_* allocate once per solver: src/pprts.F90:3396-3490)
    assert live1 - live0 <= (2 << 20), f"the pool handed out {(live1 - live0) / 2**20:.1f} MiB more over 50 rounds"
    s.close()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0
    assert pool()[2] <= live_before   # destroy returns everything the solver held to the pool


@pytest.mark.parametrize("solver,Nx,Ny,Nz,n1d", [("3_10", 8, 6, 6, 1), ("3_10", 6, 8, 5, 0), ("3_10", 12, 4, 7, 0),
                                                  ("8_16", 6, 4, 5, 1), ("8_16", 4, 6, 4, 0)])
@pytest.mark.parametrize("sweeps", [1, 2, 3, 5])
def test_red_black_preconditioner_is_checkerboard_gauss_seidel(gpu, solver, Nx, Ny, Nz, n1d, sweeps):
    """TSX_PC_REDBLACK = Gauss-Seidel over the column blocks in checkerboard order: colour (i + j) & 1 == 0, then 1, ...;
    each pass an exact column-block solve with *all* couplings to the other colour (x and y) on the right-hand side.
    Only the reduced-precision path exists (fp16 block, fp8 couplings, fp32 iterate): compared at 6 % of max."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    P = synthetic.make_problem(solver, Nx=Nx, Ny=Ny, Nz=Nz, n1d=n1d)
    lay = O.layout(solver, Nz, Nx, Ny)
    M, A = _column_block_matrix(P, lay)
    D, L = lay.D, Nz + 1
    idx = np.arange(A.shape[0])
    d, k = idx % D, (idx // D) % L
    i, j = (idx // (D * L)) % Nx, idx // (D * L * Nx)
    oi, oj = i.copy(), j.copy()
    qx, qy = d - lay.ntop, d - lay.ntop - lay.nside
    mx = (qx >= 0) & (qx < lay.nside) & (qx % 2 == 1) & (k < Nz)
    my = (qy >= 0) & (qy < lay.nside) & (qy % 2 == 1) & (k < Nz)
    oi[mx] = (i[mx] - 1) % Nx
    oj[my] = (j[my] - 1) % Ny
    colour = (oi + oj) % 2
    Noff = (A - M.tocsr()).tocsr()
    lu = spla.splu(M.tocsc(), permc_spec="NATURAL")
    v = np.random.default_rng(4).standard_normal(P["b"].shape)
    x = np.zeros(v.size)
    for p_ in range(sweeps + 1):
        rhs = v.ravel() - (Noff @ x if p_ > 0 else 0.0)
        mk = colour == (p_ % 2)
        x[mk] = lu.solve(rhs)[mk]
    s = DiffuseSolver(solver, Nz, Nx, Ny)
    s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    z = s.pc_apply(v, pc=3, sweeps=sweeps, mixed=True)
    assert np.abs(z.ravel() - x).max() <= 6e-2 * np.abs(x).max()
    # and it preconditions the solve to the same fixed point, in no more iterations than zebra rows
    xs, xz = np.zeros(s.vec_shape), np.zeros(s.vec_shape)
    ir = s.solve(P["b"], xs, rtol=1e-10, atol=1e-30, pc=3, pc_sweeps=sweeps)
    iz = s.solve(P["b"], xz, rtol=1e-10, atol=1e-30, pc=2, pc_sweeps=sweeps)
    assert ir.reason == 2 and ir.niter <= iz.niter + 1
    assert np.abs(xs - xz).max() <= 1e-8 * np.abs(xz).max()
    s.close()


@pytest.mark.parametrize("Nx,Ny,Nz,n1d", [(8, 6, 6, 1), (6, 8, 5, 0), (12, 4, 7, 2), (34, 6, 70, 3), (6, 4, 130, 0), (2, 2, 3, 0)])
@pytest.mark.parametrize("sweeps", [1, 2, 5])
def test_exact_scan_preconditioner_is_checkerboard_gauss_seidel_to_rounding(gpu, Nx, Ny, Nz, n1d, sweeps):
    """Round 6 (tsx_pcx.hip): with fp64 directions (fp32_directions = 0 -- the reference's default `ireals`) TSX_PC_REDBLACK is the
    same checkerboard Gauss-Seidel over exact column-block solves, evaluated as a segmented scan over the levels on the operator's
    own blocks with fp64 intermediates: equal to the sparse-direct model to rounding, for 4, 8 and 16 levels per thread."""
    import scipy.sparse.linalg as spla

    P = synthetic.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz, n1d=n1d)
    lay = O.layout("3_10", Nz, Nx, Ny)
    M, A = _column_block_matrix(P, lay)
    D, L = lay.D, Nz + 1
    idx = np.arange(A.shape[0])
    d, k = idx % D, (idx // D) % L
    i, j = (idx // (D * L)) % Nx, idx // (D * L * Nx)
    oi, oj = i.copy(), j.copy()
    qx, qy = d - lay.ntop, d - lay.ntop - lay.nside
    mx = (qx >= 0) & (qx < lay.nside) & (qx % 2 == 1) & (k < Nz)
    my = (qy >= 0) & (qy < lay.nside) & (qy % 2 == 1) & (k < Nz)
    oi[mx] = (i[mx] - 1) % Nx
    oj[my] = (j[my] - 1) % Ny
    colour = (oi + oj) % 2
    Noff = (A - M.tocsr()).tocsr()
    lu = spla.splu(M.tocsc(), permc_spec="NATURAL")
    v = np.random.default_rng(4).standard_normal(P["b"].shape)
    x = np.zeros(v.size)
    for p_ in range(sweeps + 1):
        rhs = v.ravel() - (Noff @ x if p_ > 0 else 0.0)
        mk = colour == (p_ % 2)
        x[mk] = lu.solve(rhs)[mk]
    s = DiffuseSolver("3_10", Nz, Nx, Ny)
    s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    z = s.pc_apply(v, pc=3, sweeps=sweeps, mixed=False)
    assert np.abs(z.ravel() - x).max() <= 1e-11 * np.abs(x).max()
    # nothing reduced anywhere: the solve with fp64 directions on the exact blocks reaches the sparse-direct solution
    xs = np.zeros(s.vec_shape)
    info = s.solve(P["b"], xs, rtol=1e-12, atol=1e-30, pc=3, pc_sweeps=sweeps, fp32_directions=0, pc_coeff_fp16=0)
    assert info.reason == 2
    x_ref = spla.spsolve(A.tocsc(), P["b"].ravel()).reshape(P["b"].shape)
    assert np.abs(xs - x_ref).max() <= 1e-9 * np.abs(x_ref).max()
    assert s.pc_info()[0] == 3   # red-black ran (not the zebra fallback)
    s.close()


def test_exact_scan_preconditioner_reads_shared_blocks_like_dense_ones(gpu, monkeypatch):
    """On the LUT path the blocks live in the shared storage only (tsx_dedup_from_coords); the exact scan passes read the
    entry-major entries through the per-cell index: same z, bit for bit, as with every cell's block stored (TSX_DEDUP=0)."""
    from tenstream_amd import lut

    Nx, Ny, Nz = 16, 12, 10
    kabs, ksca, g = synthetic.cloud_field(Nx, Ny, Nz, seed=3)
    kabs, ksca, g = synthetic.delta_scale(kabs, ksca, g)
    dz = np.full((Ny, Nx, Nz), 50.0)
    l1d = np.zeros(Nz, dtype=np.uint8)
    l1d[0] = 1
    a11, a12 = 0.6 + 0.0 * kabs, 0.1 + 0.0 * kabs
    alb = np.full((Ny, Nx), 0.2)
    v = np.random.default_rng(1).standard_normal((Ny, Nx, Nz + 1, 10))
    out = {}
    for dd in ("1", "0"):
        monkeypatch.setenv("TSX_DEDUP", dd)
        s = DiffuseSolver("3_10", Nz, Nx, Ny)
        s.set_lut_diffuse(lut.synthetic_diffuse_table("3_10"), lut.diffuse_axes("3_10"))
        s.set_optprop(kabs, ksca, g, dz, 100.0, l1d, a11, a12, alb)
        out[dd] = s.pc_apply(v, pc=3, sweeps=3, mixed=False)
        if dd == "1":
            assert s.dedup_info()[0]
        s.close()
    assert np.array_equal(out["1"], out["0"])


@pytest.mark.parametrize("solver,Nx,Ny,Nz,n1d", [
    ("3_10", 2, 2, 3, 0), ("3_10", 2, 3, 2, 0), ("3_10", 4, 2, 1, 0), ("3_10", 64, 2, 5, 1), ("3_10", 2, 64, 4, 0),
    ("3_10", 130, 4, 3, 0), ("3_10", 6, 6, 70, 3),   # Nz = 70: the sweep temporaries no longer fit the LDS block -> global
    ("3_10", 66, 6, 65, 0), ("8_16", 2, 2, 2, 0), ("8_16", 4, 6, 66, 2), ("8_16", 10, 3, 3, 0)])
def test_default_solver_on_awkward_shapes(gpu, solver, Nx, Ny, Nz, n1d):
    """The default configuration (red-black where it applies, zebra rows otherwise; LDS or global sweep temporaries) on
    minimal, thin, tall and odd grids against a sparse direct solve of the oracle's CSR."""
    import scipy.sparse.linalg as spla

    P = synthetic.make_problem(solver, Nx=Nx, Ny=Ny, Nz=Nz, n1d=n1d)
    lay = O.layout(solver, Nz, Nx, Ny)
    A = O.assemble_csr(lay, P["coeff"].astype(np.float64), P["l1d"], P["a11"], P["a12"], P["albedo"])
    x_ref = spla.spsolve(A.tocsc(), P["b"].ravel()).reshape(P["b"].shape)
    s = DiffuseSolver(solver, Nz, Nx, Ny)
    s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    x = np.zeros(s.vec_shape)
    info = s.solve(P["b"], x, rtol=1e-10, atol=1e-30)
    assert info.reason == 2, info
    assert np.abs(x - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
    # what the automatic choices resolved to (tsx_pc_info): red-black with 28 passes on the scan kernels where
    # the grid has an even number of columns per row (and of rows, the rank wrapping onto itself) and Nz <= 256; else zebra
    # rows, 10 passes
    pc, sweeps, scan, _ = s.pc_info()
    redblack = Nx % 2 == 0 and Ny % 2 == 0
    auto = 27
    assert pc == (3 if redblack else 2) and scan == redblack and sweeps == (auto if redblack else 9), (pc, sweeps, scan)
    s.close()


def test_stop_rule_reason_codes_like_MyKSPConverged(gpu, monkeypatch):
    """Diverged reasons of MyKSPConverged (src/pprts.F90:4437-4486): -3 iteration limit, -9 NaN; the oracle's restatement
    of KSPFBCGS gives the same reason and iteration count for the bare operator.  (TSX_NO_RETRY: look at the first
    attempt only; the second-solver path has its own test below.)"""
    monkeypatch.setenv("TSX_NO_RETRY", "1")
    P = synthetic.make_problem("3_10", Nx=12, Ny=10, Nz=8)
    s = DiffuseSolver("3_10", 8, 12, 10)
    s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    lay = O.layout("3_10", 8, 12, 10)
    x = np.zeros(s.vec_shape)
    info = s.solve(P["b"], x, rtol=1e-14, atol=1e-300, maxit=5, pc=0, fp32_directions=0)
    _, oi = O.solve_matfree(lay, P["coeff"].astype(np.float64), P["l1d"], P["a11"], P["a12"], P["albedo"], P["b"],
                            rtol=1e-14, atol=1e-300, maxit=5)
    assert info.reason == -3 == oi["reason"] and info.niter == oi["niter"] == 5
    np.testing.assert_allclose(info.res_hist, oi["res_hist"][: len(info.res_hist)], rtol=1e-9)  # same iterates in fp64
    b = P["b"].copy()
    b[0, 0, 0, 0] = np.nan
    x = np.zeros(s.vec_shape)
    info = s.solve(b, x, pc=2)
    assert info.reason == -9
    s.close()


def test_failed_solve_is_retried_from_zero_with_the_conservative_solver(gpu, monkeypatch):
    """src/pprts.F90:4277-4302: a solve that ends with a non-positive reason is repeated once from a zero initial guess
    with a second solver; only a second failure is reported.  A poisoned initial guess (NaN) makes the first attempt end
    with -9; the retry (exact fp64 blocks and directions, zebra-ordered column solves) converges to the oracle's solution.
    A right-hand side that is itself NaN fails twice and reports -9."""
    import scipy.sparse.linalg as spla

    P = synthetic.make_problem("3_10", Nx=12, Ny=10, Nz=8, n1d=1)
    lay = O.layout("3_10", 8, 12, 10)
    A = O.assemble_csr(lay, P["coeff"].astype(np.float64), P["l1d"], P["a11"], P["a12"], P["albedo"])
    x_ref = spla.spsolve(A.tocsc(), P["b"].ravel()).reshape(P["b"].shape)
    s = DiffuseSolver("3_10", 8, 12, 10)
    s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    x = np.zeros(s.vec_shape)
    x[3, 4, 2, 1] = np.nan
    info = s.solve(P["b"], x, rtol=1e-10, atol=1e-30)
    assert info.reason == 2 and np.isfinite(x).all()
    assert np.abs(x - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
    monkeypatch.setenv("TSX_NO_RETRY", "1")
    x = np.zeros(s.vec_shape)
    x[3, 4, 2, 1] = np.nan
    assert s.solve(P["b"], x, rtol=1e-10, atol=1e-30).reason == -9
    monkeypatch.delenv("TSX_NO_RETRY")
    b = P["b"].copy()
    b[0, 0, 0, 0] = np.nan
    x = np.zeros(s.vec_shape)
    assert s.solve(b, x).reason == -9
    s.close()


def test_accept_incomplete_solve_keeps_the_partial_iterate(gpu):
    """-accept_incomplete_solve (src/pprts.F90:4271-4273): the reference returns BEFORE the retry, so a solve limited by
    -ksp_max_it leaves its (warm-started) partial iterate and the negative reason.  Without the option the same call is
    retried from zero (reason of the second attempt, iteration counts added)."""
    import scipy.sparse.linalg as spla

    P = synthetic.make_problem("3_10", Nx=12, Ny=10, Nz=8, n1d=1)
    lay = O.layout("3_10", 8, 12, 10)
    A = O.assemble_csr(lay, P["coeff"].astype(np.float64), P["l1d"], P["a11"], P["a12"], P["albedo"])
    x_ref = spla.spsolve(A.tocsc(), P["b"].ravel()).reshape(P["b"].shape)
    s = DiffuseSolver("3_10", 8, 12, 10)
    s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    x = np.zeros(s.vec_shape)
    errs = []
    kw = dict(rtol=1e-13, atol=1e-30, maxit=2, pc=1, pc_sweeps=1)   # column block-Jacobi: converges, but not in 2 iterations
    for call in range(6):   # a fixed budget of 2 iterations per call, warm-started: the error keeps falling across calls
        info = s.solve(P["b"], x, accept_incomplete_solve=1, **kw)
        assert info.reason == -3 and info.niter == 2, info
        errs.append(np.abs(x - x_ref).max() / np.abs(x_ref).max())
    assert errs[-1] < 0.1 * errs[0] and errs[-1] > 1e-12, errs
    # without the option: the first attempt fails after its budget, the retry starts from ZERO with the conservative solver (round 6:
    # the red-black passes on the exact blocks, tsx_pcx.hip -- on this small domain two of its iterations converge, one does not) and
    # the iteration counts are added.  With a budget of one: -3 after 1 + 1 iterations, and the warm start is gone -- whatever the
    # guess was, the retry leaves the same iterate, bit for bit
    kw1 = dict(kw, maxit=1)
    x2 = x.copy()
    info = s.solve(P["b"], x2, **kw1)
    assert info.reason == -3 and info.niter == 2, info
    x3 = np.full(s.vec_shape, 0.5)
    info3 = s.solve(P["b"], x3, **kw1)
    assert info3.reason == -3 and info3.niter == 2, info3
    assert np.array_equal(x2, x3)
    # with a budget of two the retry converges: reason 2, 2 + 2 iterations, the reference's solution
    x4 = x.copy()
    info4 = s.solve(P["b"], x4, **kw)
    assert info4.reason == 2 and info4.niter == 4, info4
    assert np.abs(x4 - x_ref).max() <= 1e-9 * np.abs(x_ref).max()
    s.close()


def test_zero_guess_flag_does_not_outlive_the_guess(gpu):
    """tsx_pprts_zero_guess lets the next Krylov solve skip A x0 -- only while vx really is zero.  Three sequences that wrote
    vx afterwards (a stored solution selected, an explicit solve, the |b| < atol shortcut) used to leave the flag set: the
    next warm-started solve then took r = b with x0 != 0 and converged to x0 + A^-1 b."""
    from test_gpu_pipeline import _setup

    P, I = _setup(8, 6, 8, 200.0, 40.0, 0)
    P.set_optical_properties(0.15, I["kabs"], I["ksca"], I["g"], I["dz"])
    tight = dict(rtol=1e-11, atol=1e-30, maxit=500)
    P.solve(1000.0, uid=1, **tight)
    x1 = P.get_field("ediff").copy()
    # (1) select a fresh uid (zeroes the guess, sets the flag), then go back to uid 1 without solving in between
    import ctypes as C
    from tenstream_amd import _lib
    _lib.check(P.lib.tsx_pprts_select_solution(P.h, 7))
    info = P.solve(1000.0, uid=1, **tight)
    assert info.reason > 0
    assert np.abs(P.get_field("ediff") - x1).max() <= 1e-8 * np.abs(x1).max()
    # (2) zero guess + explicit solve, then a warm-started Krylov solve
    P.solve(1000.0, zero_guess=True, explicit_solver=1, rtol=1e-4, atol=1e-30, maxit=200, pc_sweeps=5)
    info = P.solve(1000.0, **tight)
    assert info.reason > 0
    assert np.abs(P.get_field("ediff") - x1).max() <= 1e-8 * np.abs(x1).max()
    # (3) zero guess + a solve that takes the |b|_1 < atol shortcut (ediff = b), then a real solve warm-started from it
    info = P.solve(1000.0, zero_guess=True, rtol=1e-5, atol=1e30, skip_complete_initial_run=1)
    assert info.reason == 3 and info.niter == 0
    info = P.solve(1000.0, **tight)
    assert info.reason > 0
    assert np.abs(P.get_field("ediff") - x1).max() <= 1e-8 * np.abs(x1).max()
    P.close()


def _column_block_matrix(P, lay):
    """M = entries of the assembled matrix whose row and column unknowns leave the same cell column
    (dst-owned numbering): the matrix the column preconditioner inverts exactly."""
    import scipy.sparse as sp

    D, L, Nx, Ny, Nz = lay.D, lay.Nz + 1, lay.xm, lay.ym, lay.Nz
    A = O.assemble_csr(lay, P["coeff"].astype(np.float64), P["l1d"], P["a11"], P["a12"], P["albedo"]).tocoo()
    idx = np.arange(A.shape[0])
    d, k = idx % D, (idx // D) % L
    i, j = (idx // (D * L)) % Nx, idx // (D * L * Nx)
    ntop, nside = lay.ntop, lay.nside
    oi, oj = i.copy(), j.copy()
    qx, qy = d - ntop, d - ntop - nside
    mx = (qx >= 0) & (qx < nside) & (qx % 2 == 1) & (k < Nz)
    my = (qy >= 0) & (qy < nside) & (qy % 2 == 1) & (k < Nz)
    oi[mx] = (i[mx] - 1) % Nx
    oj[my] = (j[my] - 1) % Ny
    owner = oj * Nx + oi
    same = owner[A.row] == owner[A.col]
    return sp.csc_matrix((A.data[same], (A.row[same], A.col[same])), shape=A.shape), A.tocsr()


@pytest.mark.parametrize("solver,Nx,Ny,Nz,n1d", [("3_10", 9, 7, 6, 2), ("3_10", 70, 5, 12, 0), ("8_16", 6, 5, 4, 1)])
def test_column_preconditioner_is_exact_block_inverse(gpu, solver, Nx, Ny, Nz, n1d):
    import scipy.sparse.linalg as spla

    P = synthetic.make_problem(solver, Nx=Nx, Ny=Ny, Nz=Nz, n1d=n1d)
    lay = O.layout(solver, Nz, Nx, Ny)
    M, A = _column_block_matrix(P, lay)
    s = DiffuseSolver(solver, Nz, Nx, Ny)
    s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    rng = np.random.default_rng(5)
    v = rng.standard_normal(s.vec_shape)
    z = s.pc_apply(v, pc=1, sweeps=1)
    z_ref = spla.spsolve(M, v.ravel()).reshape(v.shape)
    assert np.abs(z - z_ref).max() <= 1e-12 * np.abs(z_ref).max()
    # two sweeps: z2 = z1 + M^-1 (v - A z1)
    z2 = s.pc_apply(v, pc=1, sweeps=2)
    z2_ref = z_ref.ravel() + spla.spsolve(M, v.ravel() - A @ z_ref.ravel())
    assert np.abs(z2.ravel() - z2_ref).max() <= 1e-12 * np.abs(z2_ref).max()


@pytest.mark.parametrize("solver,Nx,Ny,Nz,n1d", [("3_10", 16, 12, 10, 2), ("8_16", 8, 6, 6, 0)])
@pytest.mark.parametrize("sweeps", [1, 2])
def test_preconditioned_solve_same_fixed_point(gpu, solver, Nx, Ny, Nz, n1d, sweeps):
    import scipy.sparse.linalg as spla

    P = synthetic.make_problem(solver, Nx=Nx, Ny=Ny, Nz=Nz, n1d=n1d)
    lay = O.layout(solver, Nz, Nx, Ny)
    A = O.assemble_csr(lay, P["coeff"].astype(np.float64), P["l1d"], P["a11"], P["a12"], P["albedo"])
    x_ref = spla.spsolve(A.tocsc(), P["b"].ravel()).reshape(P["b"].shape)
    s = DiffuseSolver(solver, Nz, Nx, Ny)
    s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    x0 = np.zeros(s.vec_shape)
    plain = s.solve(P["b"], x0, rtol=1e-10, atol=1e-30, pc=0)
    x = np.zeros(s.vec_shape)
    info = s.solve(P["b"], x, rtol=1e-10, atol=1e-30, pc=1, pc_sweeps=sweeps)
    assert info.reason == 2 and info.niter < plain.niter
    assert np.abs(x - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
    # true residual agrees with the recurrence residual the stop rule used
    r = P["b"].ravel() - A @ x.ravel()
    assert np.linalg.norm(r) <= 1.5 * info.rnorm + 1e-12 * np.linalg.norm(P["b"])


def test_rccl_transport_with_one_rank_communicator(gpu):
    """Drives the RCCL code path (ncclSend/ncclRecv face exchange in one group, ncclAllReduce of the dot
    products) on a single GPU: a 1-rank communicator whose four neighbours are the rank itself."""
    P = synthetic.make_problem("3_10", Nx=20, Ny=12, Nz=9, n1d=1)
    s = DiffuseSolver("3_10", 9, 20, 12, force_halo=True)
    s.comm_init(s.comm_unique_id())
    s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    lay = O.layout("3_10", 9, 20, 12)
    c64 = P["coeff"].astype(np.float64)
    x = np.random.default_rng(2).standard_normal(s.vec_shape)
    y = s.apply(x)
    y_ref = O.diff_apply(lay, c64, P["l1d"], P["a11"], P["a12"], P["albedo"], x)
    assert np.abs(y - y_ref).max() <= 1e-13 * np.abs(y_ref).max()
    xs = np.zeros(s.vec_shape)
    info = s.solve(P["b"], xs, rtol=1e-10, atol=1e-30, pc=1, pc_sweeps=1)
    x_ref, _ = O.solve_matfree(lay, c64, P["l1d"], P["a11"], P["a12"], P["albedo"], P["b"], rtol=1e-12, atol=1e-30)
    assert info.reason == 2 and np.abs(xs - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
    # the default preconditioner too: its boundary columns travel over the same communicator after every pass
    xs2 = np.zeros(s.vec_shape)
    info2 = s.solve(P["b"], xs2, rtol=1e-10, atol=1e-30)
    assert info2.reason == 2 and np.abs(xs2 - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
    s.close()   # destroy the communicator now, not at interpreter exit (RCCL's own teardown runs there)


def test_fortran_shim_driver(gpu):
    """The ISO_C_BINDING shim (tenstream_amd/fortran/m_pprts_hip.F90) called from a Fortran program."""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tenstream_amd", "lib", "test_shim")
    if not os.path.exists(exe):
        pytest.skip("amdflang not available when the tree was built")
    r = None
    for attempt in range(2):   # (a fresh box's first RCCL communicator has been seen to take minutes once: one more try)
        try:
            r = subprocess.run([exe], capture_output=True, text=True, timeout=180)
            break
        except subprocess.TimeoutExpired as e:
            out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
            assert attempt == 0, "test_shim hung twice; it got as far as:\n" + out
    assert r.returncode == 0, r.stdout + r.stderr
    # every group of bindings ran: the diffuse seam (real64, real32), the direct seam + setup_b, the communicator entries
    # (Fortran callbacks, RCCL id / init), the LUT and whole-g-point entries
    for line in ("shim ok", "shim real32 ok", "shim direct seam ok", "shim thermal source ok", "shim comm bindings ok",
                 "shim pipeline ok (solar)", "shim pipeline ok (thermal)", "shim all ok"):
        assert line in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("solver,Nx,Ny,Nz", [("3_10", 21, 9, 12), ("8_16", 8, 5, 6)])
def test_device_lut_lookup_bit_exact(gpu, solver, Nx, Ny, Nz, tmp_path):
    """K4 on the device (clamp -> bisection -> N-linear interpolation with snapping) is bit-identical to the oracle's
    restatement of get_coeff/LUT_get_diff2diff, both through tsx_lut_set_diffuse and through a `.mmap4` file."""
    from tenstream_amd import lut

    axes = lut.diffuse_axes(solver)
    table = lut.synthetic_diffuse_table(solver)
    kabs, ksca, g = synthetic.cloud_field(Nx, Ny, Nz, seed=3)
    rng = np.random.default_rng(4)
    ksca *= rng.uniform(0.2, 30.0, ksca.shape)  # spread over many table cells, some beyond the tau range
    kabs[0, 0, :] = 0.0
    ksca[0, 0, :] = 0.0                          # tau == 0 -> clamped to the first node
    kabs, ksca, g = synthetic.delta_scale(kabs, ksca, g)
    dz = rng.uniform(20.0, 400.0, kabs.shape)
    l1d = np.zeros(Nz, dtype=np.uint8)
    l1d[:2] = 1
    a11 = np.full(kabs.shape, 0.5)
    a12 = np.full(kabs.shape, 0.2)
    alb = np.full((Ny, Nx), 0.1)
    L = O.make_lut(axes, table)
    ref = O.alloc_coeff_diff2diff(L, kabs, ksca, g, dz, 100.0, l1d)

    s = DiffuseSolver(solver, Nz, Nx, Ny)
    s.set_lut_diffuse(table, axes)
    s.set_optprop(kabs, ksca, g, dz, 100.0, l1d, a11, a12, alb)
    got = s.get_coeffs()
    assert np.array_equal(got[:, :, 2:], ref[:, :, 2:])  # bit-exact (1-D layers hold no blocks)

    if solver == "3_10":
        p = tmp_path / lut.diffuse_lut_filename("LUT", solver)
        lut.write_mmap4(p, table)
        s2 = DiffuseSolver(solver, Nz, Nx, Ny)
        s2.load_lut_diffuse_mmap4(str(p))
        s2.set_optprop(kabs, ksca, g, dz, 100.0, l1d, a11, a12, alb)
        assert np.array_equal(s2.get_coeffs()[:, :, 2:], ref[:, :, 2:])
        # and the operator built from looked-up blocks equals the oracle operator on the oracle's blocks
        lay = O.layout(solver, Nz, Nx, Ny)
        x = rng.standard_normal(s2.vec_shape)
        y_ref = O.diff_apply(lay, ref, l1d, a11, a12, alb, x)
        assert np.abs(s2.apply(x) - y_ref).max() <= 1e-13 * np.abs(y_ref).max()


def test_breakdown_restart(gpu):
    """A right-hand side supported only on the TOA identity rows makes rho = (rhat, r) vanish after one step
    (rhat = r0 sees nothing of the interior): a true BiCGStab breakdown.  The reference retries with GMRES
    (src/pprts.F90:4277-4296); here the solve restarts from the current iterate with a fresh shadow residual."""
    P = synthetic.make_problem("3_10", Nx=6, Ny=5, Nz=8)
    s = DiffuseSolver("3_10", 8, 6, 5)
    s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    b = np.zeros(s.vec_shape)
    b[:, :, 0, 1] = 1.0
    lay = O.layout("3_10", 8, 6, 5)
    import scipy.sparse.linalg as spla

    A = O.assemble_csr(lay, P["coeff"].astype(np.float64), P["l1d"], P["a11"], P["a12"], P["albedo"])
    x_ref = spla.spsolve(A.tocsc(), b.ravel()).reshape(b.shape)
    # the oracle's textbook BiCGStab (no restart, like PETSc's) reports the breakdown
    _, oi = O.solve_matfree(lay, P["coeff"].astype(np.float64), P["l1d"], P["a11"], P["a12"], P["albedo"], b, rtol=1e-10,
                            atol=1e-30)
    assert oi["reason"] == -5
    for pc in (0, 1):
        x = np.zeros(s.vec_shape)
        info = s.solve(b, x, rtol=1e-10, atol=1e-30, pc=pc, pc_sweeps=1)
        assert info.reason == 2, (pc, info)
        assert np.abs(x - x_ref).max() <= 1e-8 * np.abs(x_ref).max()


@pytest.mark.parametrize("solver,Nx,Ny,Nz,n1d", [("3_10", 9, 8, 6, 1), ("3_10", 6, 7, 5, 0), ("8_16", 5, 6, 4, 0),
                                                  ("8_16", 6, 6, 7, 2)])
@pytest.mark.parametrize("sweeps", [1, 2, 3, 4])
def test_zebra_preconditioner_is_line_gauss_seidel(gpu, solver, Nx, Ny, Nz, n1d, sweeps):
    """TSX_PC_ZEBRA = Gauss-Seidel in y over the column blocks: even rows, odd rows (+ even again), each pass an exact
    column-block solve with the other colour's +-y streams on the right-hand side."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    P = synthetic.make_problem(solver, Nx=Nx, Ny=Ny, Nz=Nz, n1d=n1d)
    lay = O.layout(solver, Nz, Nx, Ny)
    M, A = _column_block_matrix(P, lay)
    D, L = lay.D, Nz + 1
    idx = np.arange(A.shape[0])
    d, k = idx % D, (idx // D) % L
    j = idx // (D * L * Nx)
    qy = d - lay.ntop - lay.nside
    my = (qy >= 0) & (qy < lay.nside) & (qy % 2 == 1) & (k < Nz)
    oj = j.copy()
    oj[my] = (j[my] - 1) % Ny
    Ac = A.tocoo()
    ydiff = oj[Ac.row] != oj[Ac.col]
    if Ny % 2:  # odd row count: the kernel drops the coupling across the periodic seam (rows 0 and Ny-1 share a colour)
        seam = ((oj[Ac.row] == 0) & (oj[Ac.col] == Ny - 1)) | ((oj[Ac.row] == Ny - 1) & (oj[Ac.col] == 0))
        ydiff &= ~seam
    Nyc = sp.csr_matrix((Ac.data[ydiff], (Ac.row[ydiff], Ac.col[ydiff])), shape=A.shape)
    # x coupling = off-column entries within the same row of columns (3_10 only: lagged Jacobi in x from pass 3 on)
    qx = d - lay.ntop
    mx = (qx >= 0) & (qx < lay.nside) & (qx % 2 == 1) & (k < Nz)
    i_ = (idx // (D * L)) % Nx
    oi = i_.copy()
    oi[mx] = (i_[mx] - 1) % Nx
    xdiff = (oj[Ac.row] == oj[Ac.col]) & (oi[Ac.row] != oi[Ac.col])
    Nxc = sp.csr_matrix((Ac.data[xdiff], (Ac.row[xdiff], Ac.col[xdiff])), shape=A.shape)
    even = oj % 2 == 0
    lu = spla.splu(M.tocsc(), permc_spec="NATURAL")
    rng = np.random.default_rng(9)
    v = rng.standard_normal(P["b"].shape)
    def model(with_x):
        x = np.zeros(v.size)
        for p_ in range(sweeps + 1):
            mk = even if p_ % 2 == 0 else ~even
            rhs = v.ravel().copy()
            if p_ > 0:
                rhs -= Nyc @ x
            if p_ > 1 and with_x:
                rhs -= Nxc @ x
            x[mk] = lu.solve(rhs)[mk]
        return x

    x = model(solver == "3_10")  # exact fp64 path: the generic 8_16 kernel couples in y only
    s = DiffuseSolver(solver, Nz, Nx, Ny)
    s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    z = s.pc_apply(v, pc=2, sweeps=sweeps)
    assert np.abs(z.ravel() - x).max() <= 1e-12 * np.abs(x).max()
    # the path the solver takes by default: packed reduced-precision blocks, fp32 temporaries and output, x coupling for
    # both solvers -- the same preconditioner up to that rounding: fp16 (2^-11) for the column blocks, fp8 e4m3 (2^-4
    # per coefficient) for the couplings to neighbouring columns, which only enter the right-hand side
    xm = model(True)
    zm = s.pc_apply(v, pc=2, sweeps=sweeps, mixed=True)
    assert np.abs(zm.ravel() - xm).max() <= 6e-2 * np.abs(xm).max()
    # and the zebra-preconditioned solve reaches the same solution in fewer iterations than block-Jacobi
    xs, xj = np.zeros(s.vec_shape), np.zeros(s.vec_shape)
    iz = s.solve(P["b"], xs, rtol=1e-10, atol=1e-30, pc=2, pc_sweeps=sweeps)
    ij = s.solve(P["b"], xj, rtol=1e-10, atol=1e-30, pc=1, pc_sweeps=1)
    assert iz.reason == 2 and iz.niter <= ij.niter
    assert np.abs(xs - xj).max() <= 1e-8 * np.abs(xj).max()


@pytest.mark.parametrize("Nx,Ny,Nz,n1d,cfg", [
    (12, 6, 6, 1, "4,16,16"), (12, 6, 11, 0, "4,16,32"), (34, 6, 20, 3, "8,8,64"), (10, 4, 37, 0, "8,8,16"),
    (8, 6, 70, 2, "8,16,32"), (6, 4, 130, 0, "16,16,16"), (130, 2, 9, 0, "4,16,64"), (6, 4, 64, 0, "4,16,16"),
    (6, 4, 64, 0, "8,8,32"), (20, 10, 33, 4, "")])
def test_scan_preconditioner_equals_the_column_sweep(gpu, monkeypatch, Nx, Ny, Nz, n1d, cfg):
    """tsx_k_pcs_rb (segmented scan over the levels: LSEG x NSEG levels, CW columns per workgroup; ragged last segment,
    idle segments, columns that do not fill a workgroup) computes the same red-black M^-1 as the one-lane-per-column
    sweep tsx_k_pc_column_rb and as the sparse model.  Both run on reduced-precision blocks (roundings differ: the scan
    stores the matrix-only recurrences and the side -> top couplings in fp16, the sweep the four top coefficients in fp16 and
    every coupling in fp8 e4m3, 2^-4 per coefficient): 6 % of max between them and 6 % of max
    against the exact model, like test_red_black_preconditioner_is_checkerboard_gauss_seidel."""
    import scipy.sparse.linalg as spla

    P = synthetic.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz, n1d=n1d)
    lay = O.layout("3_10", Nz, Nx, Ny)
    v = np.random.default_rng(11).standard_normal(P["b"].shape)
    out = {}
    for scan in ("0", "1"):
        monkeypatch.setenv("TSX_PC_SCAN", scan)
        if cfg:
            monkeypatch.setenv("TSX_PCS_CFG", cfg)
        s = DiffuseSolver("3_10", Nz, Nx, Ny)
        s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
        out[scan] = [s.pc_apply(v, pc=3, sweeps=sw, mixed=True) for sw in (1, 2, 4)]
        x = np.zeros(s.vec_shape)
        info = s.solve(P["b"], x, rtol=1e-10, atol=1e-30, pc_sweeps=9)   # same pass count (the automatic one differs)
        out["its" + scan], out["x" + scan] = info.niter, x
        assert info.reason == 2
        s.close()
    for a, b in zip(out["0"], out["1"]):   # the sweep rounds the side -> top couplings to fp8 (2^-4), the scan to fp16
        assert np.abs(a - b).max() <= 6e-2 * np.abs(a).max()
    # tight solves: a few dozen iterations.  The scan never needs more than the sweep; it may need fewer: it keeps the side ->
    # top couplings in fp16 (tsx_kernels_pcs.hpp C16), the sweep kernels in fp8
    assert out["its1"] <= out["its0"] + max(1, round(0.1 * out["its0"])) and out["its1"] >= 0.6 * out["its0"], (out["its0"], out["its1"])
    assert np.abs(out["x0"] - out["x1"]).max() <= 1e-8 * np.abs(out["x0"]).max()
    # against the model: red-black Gauss-Seidel on the exact column blocks, 5 passes
    M, A = _column_block_matrix(P, lay)
    D, L = lay.D, Nz + 1
    idx = np.arange(A.shape[0])
    d, k = idx % D, (idx // D) % L
    i, j = (idx // (D * L)) % Nx, idx // (D * L * Nx)
    oi, oj = i.copy(), j.copy()
    qx, qy = d - 2, d - 6
    mx = (qx >= 0) & (qx < 4) & (qx % 2 == 1) & (k < Nz)
    my = (qy >= 0) & (qy < 4) & (qy % 2 == 1) & (k < Nz)
    oi[mx] = (i[mx] - 1) % Nx
    oj[my] = (j[my] - 1) % Ny
    colour = (oi + oj) % 2
    Noff = (A - M.tocsr()).tocsr()
    lu = spla.splu(M.tocsc(), permc_spec="NATURAL")
    x = np.zeros(v.size)
    for p_ in range(5):
        rhs = v.ravel() - (Noff @ x if p_ > 0 else 0.0)
        mk = colour == (p_ % 2)
        x[mk] = lu.solve(rhs)[mk]
    assert np.abs(out["1"][2].ravel() - x).max() <= 6e-2 * np.abs(x).max()


@pytest.mark.parametrize("solver,Nx,Ny,Nz", [("3_10", 256, 128, 6), ("8_16", 16, 12, 9)])
def test_shared_recurrence_records_are_lossless(gpu, monkeypatch, solver, Nx, Ny, Nz):
    """tsx_records_share: the preconditioner's column recurrence records repeat wherever the column is the same from the cell
    down to the surface (every clear column, every cloudy column below its lowest cloud); they are stored once behind a second
    per-cell index (3_10: on passes of >= 16 K columns; 8_16: always).  Same numbers in the same order: the solve -- residual
    history included -- is bit-identical with TSX_PC_RECSHARE=0 and 1."""
    P = synthetic.make_problem(solver, Nx=Nx, Ny=Ny, Nz=Nz, n1d=0)
    coeff = P["coeff"].copy()
    keep = np.zeros((Ny, Nx), dtype=bool)
    keep[1, 2] = keep[Ny // 2, Nx // 3] = keep[Ny - 1, Nx - 1] = True
    coeff[~keep] = coeff[0, 0, Nz - 1]                       # one block everywhere ...
    top = slice(0, Nz // 2)                                  # ... but a "cloud" in the upper half of three columns
    fac = 1.0 - 0.01 * np.random.default_rng(5).random((int(keep.sum()), Nz // 2, 1))
    cl = coeff[keep]
    cl[:, top] = (cl[:, top] * fac).astype(np.float32)
    cl[:, Nz // 2:] = coeff[0, 0, Nz - 1]                    # below the cloud those columns are like the clear ones
    coeff[keep] = cl
    res = {}
    for share in ("0", "1"):
        monkeypatch.setenv("TSX_PC_RECSHARE", share)
        s = DiffuseSolver(solver, Nz, Nx, Ny)
        s.set_coeffs(coeff, P["l1d"], P["a11"], P["a12"], P["albedo"])
        xs = np.zeros(s.vec_shape)
        info = s.solve(P["b"], xs, rtol=1e-9, atol=1e-30)
        # round 6: a second coefficient set with the same structure (every block scaled): the grouping of the records is taken over
        # after one validation kernel (tsx_k_rec_validate) instead of being rebuilt; a third with another structure rebuilds
        s.set_coeffs((coeff * np.float32(0.97)).astype(np.float32), P["l1d"], P["a11"], P["a12"], P["albedo"])
        x2 = np.zeros(s.vec_shape)
        info2 = s.solve(P["b"], x2, rtol=1e-9, atol=1e-30)
        s.dedup_info()
        reused2 = bool(s.dedup_mode & 8)
        c3 = coeff.copy()
        c3[keep] = (c3[keep] * np.float32(0.9)).astype(np.float32)   # now the three columns differ down to the surface
        s.set_coeffs(c3, P["l1d"], P["a11"], P["a12"], P["albedo"])
        x3 = np.zeros(s.vec_shape)
        info3 = s.solve(P["b"], x3, rtol=1e-9, atol=1e-30)
        s.dedup_info()
        reused3 = bool(s.dedup_mode & 8)
        res[share] = (s.pc_info(), xs, info, x2, info2, x3, info3, reused2, reused3)
        s.close()
    assert res["0"][0][2] and not res["0"][0][3] and res["1"][0][3], (res["0"][0], res["1"][0])
    assert res["0"][2].reason == 2 and res["0"][2].niter == res["1"][2].niter
    assert np.array_equal(res["0"][2].res_hist, res["1"][2].res_hist) and np.array_equal(res["0"][1], res["1"][1])
    assert res["1"][7] and not res["1"][8] and not res["0"][7], (res["1"][7:], res["0"][7:])
    for q in (3, 5):   # the taken-over grouping and the rebuilt one: bit-identical to no sharing at all
        assert np.array_equal(res["0"][q], res["1"][q]) and np.array_equal(res["0"][q + 1].res_hist, res["1"][q + 1].res_hist)


@pytest.mark.parametrize("solver,Nx,Ny,Nz,n1d", [("3_10", 16, 12, 9, 2), ("3_10", 10, 6, 5, 0), ("8_16", 8, 6, 5, 1)])
def test_shared_block_storage_is_lossless(gpu, monkeypatch, solver, Nx, Ny, Nz, n1d):
    """tsx_dedup.hip: cells with bit-identical transport blocks share one stored copy behind a per-cell index.  Same
    numbers, same order of operations: the operator apply and the whole preconditioned solve (residual history included)
    are bit-identical with TSX_DEDUP=0 and 1, on a field where most blocks repeat (a clear-sky background under a few
    cloudy columns, 1-D layers on top) -- and against the oracle.  A field of all-different blocks is left dense."""
    P = synthetic.make_problem(solver, Nx=Nx, Ny=Ny, Nz=Nz, n1d=n1d)
    D = P["D"]
    coeff = P["coeff"].copy()
    # homogeneous background: every cell gets the block of cell (0, 0, Nz-1), a few columns keep their own
    keep = np.zeros((Ny, Nx), dtype=bool)
    keep[1, 2] = keep[3, 3] = keep[Ny - 1, Nx - 1] = keep[2, 0] = True
    coeff[~keep] = coeff[0, 0, Nz - 1]
    fac = 1.0 - 0.01 * np.random.default_rng(9).random((Ny, Nx, Nz, 1))   # the kept columns: a different block in every cell
    coeff[keep] = (coeff[keep] * fac[keep]).astype(np.float32)
    lay = O.layout(solver, Nz, Nx, Ny)
    x = np.random.default_rng(3).standard_normal(P["b"].shape)
    res = {}
    monkeypatch.setenv("TSX_SPMV_CPT", "2")   # same cells per thread on both sides: same summation order of the fused dots
    for dd in ("0", "1"):
        monkeypatch.setenv("TSX_DEDUP", dd)
        s = DiffuseSolver(solver, Nz, Nx, Ny)
        s.set_coeffs(coeff, P["l1d"], P["a11"], P["a12"], P["albedo"])
        on, nent = s.dedup_info()
        y = s.apply(x)
        xs = np.zeros(s.vec_shape)
        info = s.solve(P["b"], xs, rtol=1e-10, atol=1e-30, pc_sweeps=9)   # (the automatic pass count depends on the storage)
        res[dd] = (on, nent, y, xs, info)
        s.close()
    assert not res["0"][0] and res["1"][0]
    # distinct blocks: the background, the kept columns' 3-D cells, and one entry standing for all 1-D cells
    assert res["1"][1] == 1 + int(keep.sum()) * (Nz - n1d) + (1 if n1d else 0) - (1 if keep[0, 0] else 0)
    assert np.array_equal(res["0"][2], res["1"][2])
    assert np.array_equal(res["0"][3], res["1"][3]) and res["0"][4].niter == res["1"][4].niter
    assert np.array_equal(res["0"][4].res_hist, res["1"][4].res_hist)
    y_ref = _ref_apply(dict(P, coeff=coeff), lay, x)
    assert np.abs(res["1"][2] - y_ref).max() <= 1e-13 * np.abs(y_ref).max()
    # the default launch geometry with shared blocks (one cell per thread): the same solution to rounding
    monkeypatch.delenv("TSX_SPMV_CPT")
    s = DiffuseSolver(solver, Nz, Nx, Ny)
    s.set_coeffs(coeff, P["l1d"], P["a11"], P["a12"], P["albedo"])
    xs = np.zeros(s.vec_shape)
    assert s.solve(P["b"], xs, rtol=1e-10, atol=1e-30).reason == 2
    assert np.abs(xs - res["0"][3]).max() <= 1e-9 * np.abs(xs).max()
    s.close()
    # every block different: the index would only cost, the dense planes stay
    monkeypatch.setenv("TSX_DEDUP", "1")
    s = DiffuseSolver(solver, Nz, Nx, Ny)
    s.set_coeffs(P["coeff"] * (1 + 1e-3 * np.random.default_rng(5).random(P["coeff"].shape)).astype(np.float32), P["l1d"], P["a11"], P["a12"], P["albedo"])
    on, nent = s.dedup_info()
    # nothing bit-identical: the operator keeps the dense planes; the blocks agree to 0.1 % though, so the PRECONDITIONER groups
    # them (tsx_dedup.hip "near-identical blocks": its per-block records are approximate by design) ...
    # (3_10 only: the 8_16 pass gathers 20 records per level, scattered over the groups they cost more than streamed per cell)
    near = solver == "3_10"
    assert not on and s.dedup_mode == (2 if near else 0) and (nent < 0.5 * Nx * Ny * Nz) == near
    xs = np.zeros(s.vec_shape)
    i_near = s.solve(P["b"], xs, rtol=1e-10, atol=1e-30)
    s.close()
    # ... unless that is switched off: then every cell counts as a block of its own
    monkeypatch.setenv("TSX_DEDUP_NEAR", "0")
    s = DiffuseSolver(solver, Nz, Nx, Ny)
    s.set_coeffs(P["coeff"] * (1 + 1e-3 * np.random.default_rng(5).random(P["coeff"].shape)).astype(np.float32), P["l1d"], P["a11"], P["a12"], P["albedo"])
    on, nent = s.dedup_info()
    assert not on and s.dedup_mode == 0 and nent > 0.5 * Nx * Ny * Nz
    xs2 = np.zeros(s.vec_shape)
    i_own = s.solve(P["b"], xs2, rtol=1e-10, atol=1e-30)
    s.close()
    # the same system either way (the operator is exact): same solution, iteration counts within one
    assert i_near.reason == 2 and i_own.reason == 2 and abs(i_near.niter - i_own.niter) <= 1
    assert np.abs(xs - xs2).max() <= 1e-8 * np.abs(xs2).max()


@pytest.mark.parametrize("Nx,Ny,Nz", [(64, 32, 20), (128, 128, 12)])
def test_entry_major_records_fetched_by_lane_groups_equal_the_lane_by_lane_fetch(gpu, monkeypatch, Nx, Ny, Nz):
    """Near-identical grouping stores the preconditioner's per-block records entry-major (an entry's eight records in one
    128-byte line); the passes fetch phase 3's six records of such entries by groups of eight lanes and hand them over through
    LDS (tsx_k_pcs_rb COOP; 16-column workgroups on the first domain, 32-column ones on the second).  Slot-major storage of the
    same records (TSX_PC_ENTRY_MAJOR=0) is fetched lane by lane: M^-1 v must be bit-identical, for the default number of passes
    and for a short sequence."""
    P = synthetic.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz)
    coeff = (P["coeff"] * (1 + 1e-3 * np.random.default_rng(11).random(P["coeff"].shape))).astype(np.float32)
    v = np.random.default_rng(12).standard_normal(P["b"].shape)
    out = {}
    for env in ("1", "0"):
        monkeypatch.setenv("TSX_PC_ENTRY_MAJOR", env)
        s = DiffuseSolver("3_10", Nz, Nx, Ny)
        s.set_coeffs(coeff, P["l1d"], P["a11"], P["a12"], P["albedo"])
        on, nent = s.dedup_info()
        assert not on and s.dedup_mode == 2 and nent < 0.5 * Nx * Ny * Nz   # grouped for the preconditioner only
        out[env] = [s.pc_apply(v, pc=3, sweeps=sw, mixed=True) for sw in (27, 3)]
        x = np.zeros(s.vec_shape)
        info = s.solve(P["b"], x, rtol=1e-8, atol=1e-30)
        assert info.reason == 2
        out[env].append(x)
        s.close()
    for a, b in zip(out["1"], out["0"]):
        assert np.isfinite(a).all() and np.array_equal(a, b)


def test_half_step_exit_meets_the_stop_rule_on_the_true_residual(gpu, monkeypatch):
    """BiCGStab's first half step already has an iterate, x + alpha p-hat, with residual s = r - alpha v.  When s meets
    MyKSPConverged's rule (src/pprts.F90:4437-4486) the solve stops there (TSX_STAGE_HALF) instead of finishing an iteration on
    a converged system.  Same criterion, decided on the true residual where the recurrence is fp32: checked here with an
    independent operator apply, against the same solves with the test switched off (TSX_HALF_EXIT=0)."""
    taken = 0
    for solver, shape, n1d in (("3_10", (16, 12, 10), 0), ("3_10", (12, 10, 8), 2), ("8_16", (8, 6, 8), 1), ("3_10", (24, 16, 12), 0)):
        Nx, Ny, Nz = shape
        P = synthetic.make_problem(solver, Nx=Nx, Ny=Ny, Nz=Nz, n1d=n1d)
        s = DiffuseSolver(solver, Nz, Nx, Ny)
        s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
        bn = np.linalg.norm(P["b"])
        for kw in (dict(rtol=1e-5, atol=1e-30), dict(rtol=1e-7, atol=1e-30), dict(rtol=1e-10, atol=1e-30), dict(rtol=1e-4, atol=1e-30, pc=1, pc_sweeps=1),
                   dict(rtol=1e-6, atol=1e-30, pc=0), dict(rtol=1e-6, atol=1e-30, fp32_directions=0, pc_coeff_fp16=0)):
            out = {}
            for env in ("1", "0"):
                monkeypatch.setenv("TSX_HALF_EXIT", env)
                x = np.zeros(s.vec_shape)
                info = s.solve(P["b"], x, **kw)
                rn = np.linalg.norm(P["b"] - s.apply(x))
                assert info.reason == 2, (solver, kw, env, info)
                assert rn <= kw["rtol"] * bn * (1 + 1e-6) + 1e-14 * bn, (solver, kw, env, rn / bn)
                assert abs(info.rnorm - rn) <= 1e-6 * bn and abs(info.res_hist[-1] - info.rnorm) <= 1e-12 * bn
                out[env] = (info.niter, x)
            assert out["1"][0] <= out["0"][0], (solver, kw, out["1"][0], out["0"][0])
            taken += not np.array_equal(out["1"][1], out["0"][1])
        s.close()
    assert taken >= 3, taken   # the exit is taken in a fair share of the solves (24 here)


def test_sharing_keyed_on_lut_coordinates_is_lossless(gpu, monkeypatch):
    """Round 4, tsx_dedup_from_coords: on the LUT path cells are grouped by their four clamped float32 LUT coordinates BEFORE
    anything is interpolated; only the distinct tuples are interpolated, straight into the shared storage, and no dense per-cell
    planes are written.  Against the same solver with TSX_DEDUP_COORDS=0 (every cell interpolated, blocks compared afterwards):
    the blocks read back (expanded from the entries on demand) are bit-identical and equal the oracle's lookup, operator apply
    and residual history are bit-identical; coordinates that differ may still give identical blocks, so the coordinate-keyed
    storage has at least as many entries.  Every consumer of dense planes still works afterwards (exact fp64 preconditioner,
    zebra rows)."""
    from tenstream_amd import lut

    Nx, Ny, Nz = 24, 16, 12
    kabs, ksca, g = synthetic.cloud_field(Nx, Ny, Nz, seed=4, cover=0.2)
    kabs, ksca, g = synthetic.delta_scale(kabs, ksca, g)
    dz = np.full((Ny, Nx, Nz), 50.0)
    dz[:, :, :2] = 300.0
    l1d = np.zeros(Nz, dtype=np.uint8)
    l1d[:2] = 1
    P = synthetic.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz, n1d=2, seed=4)
    axes, table = lut.diffuse_axes("3_10"), lut.synthetic_diffuse_table("3_10")
    x = np.random.default_rng(2).standard_normal(P["b"].shape)
    res = {}
    monkeypatch.setenv("TSX_SPMV_CPT", "1")
    for dc in ("0", "1"):
        monkeypatch.setenv("TSX_DEDUP_COORDS", dc)
        s = DiffuseSolver("3_10", Nz, Nx, Ny)
        s.set_lut_diffuse(table, axes)
        s.set_optprop(kabs, ksca, g, dz, 100.0, l1d, P["a11"], P["a12"], P["albedo"])
        on, nent = s.dedup_info()
        y = s.apply(x)
        xs = np.zeros(s.vec_shape)
        info = s.solve(P["b"], xs, rtol=1e-10, atol=1e-30)
        coeff = s.get_coeffs()     # dc = 1: expanded from the shared entries
        # consumers of the dense planes after the fact: exact fp64 blocks in zebra order, and the bare operator again
        x2 = np.zeros(s.vec_shape)
        i2 = s.solve(P["b"], x2, rtol=1e-10, atol=1e-30, fp32_directions=0, pc_coeff_fp16=0)
        assert i2.reason == 2 and np.abs(x2 - xs).max() <= 1e-8 * np.abs(xs).max()
        assert np.array_equal(s.apply(x), y)
        res[dc] = (on, nent, y, xs, info, coeff)
        s.close()
    assert res["0"][0] and res["1"][0] and res["1"][1] >= res["0"][1] and res["1"][1] < 0.5 * Nx * Ny * Nz
    sl = slice(2, None)
    assert np.array_equal(res["0"][5][:, :, sl], res["1"][5][:, :, sl])
    Ld = O.make_lut(axes, table)
    want = O.alloc_coeff_diff2diff(Ld, kabs, ksca, g, dz, 100.0, l1d)
    assert np.array_equal(res["1"][5][:, :, sl], want[:, :, sl])
    assert np.array_equal(res["0"][2], res["1"][2])
    assert res["0"][4].niter == res["1"][4].niter and np.array_equal(res["0"][4].res_hist, res["1"][4].res_hist)
    assert np.array_equal(res["0"][3], res["1"][3])


def test_sharing_structure_taken_over_from_the_previous_set_is_lossless(gpu, monkeypatch):
    """Round 5, tsx_k_dd_validate_coords: a spectral loop hands one solver a coefficient set per g-point; which cells share a LUT tuple
    rarely changes between them, so the grouping of the previous set is kept when ONE kernel finds every cell's clamped tuple still
    bit-equal to its entry's representative's (tsx_dedup_info bit 2), and only the entries are interpolated anew.  A sequence of sets on
    one solver -- the same scene scaled per g-point, another scene, another set of 1-D layers, the first again -- against a solver that
    rebuilds every time (TSX_DEDUP_REUSE=0): blocks read back, operator apply and residual histories bit-identical for every set; the
    grouping is taken over for the scaled sets and rebuilt when the scene or the 1-D layers change."""
    from tenstream_amd import lut

    Nx, Ny, Nz = 24, 16, 12
    dz = np.full((Ny, Nx, Nz), 50.0)
    dz[:, :, :2] = 300.0
    P = synthetic.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz, n1d=2, seed=4)
    axes, table = lut.diffuse_axes("3_10"), lut.synthetic_diffuse_table("3_10")
    x = np.random.default_rng(2).standard_normal(P["b"].shape)

    def scene(seed, scale, n1d):
        kabs, ksca, g = synthetic.cloud_field(Nx, Ny, Nz, seed=seed, cover=0.2)
        kabs, ksca, g = synthetic.delta_scale(kabs, ksca, g)
        l1d = np.zeros(Nz, dtype=np.uint8)
        l1d[:n1d] = 1
        return kabs * scale, ksca * scale, g, l1d

    sets = [(4, 1.0, 2), (4, 1.0, 2), (4, 0.5, 2), (4, 0.03, 2), (5, 1.0, 2), (5, 0.7, 2), (5, 0.7, 3), (4, 1.0, 2)]
    res = {}
    monkeypatch.setenv("TSX_SPMV_CPT", "1")
    for reuse in ("0", "1"):
        monkeypatch.setenv("TSX_DEDUP_REUSE", reuse)
        s = DiffuseSolver("3_10", Nz, Nx, Ny)
        s.set_lut_diffuse(table, axes)
        out = []
        for seed, scale, n1d in sets:
            kabs, ksca, g, l1d = scene(seed, scale, n1d)
            s.set_optprop(kabs, ksca, g, dz, 100.0, l1d, P["a11"], P["a12"], P["albedo"])
            on, nent = s.dedup_info()
            y = s.apply(x)
            xs = np.zeros(s.vec_shape)
            info = s.solve(P["b"], xs, rtol=1e-10, atol=1e-30)
            assert info.reason == 2
            out.append((s.dedup_mode, nent, y, xs, info.res_hist.copy(), s.get_coeffs()[:, :, n1d:]))
        res[reuse] = out
        s.close()
    taken = [bool(o[0] & 4) for o in res["1"]]
    assert not any(o[0] & 4 for o in res["0"])
    assert taken == [False, True, True, True, False, True, False, False], taken
    for a, b in zip(res["0"], res["1"]):
        assert (a[0] & 1) and (b[0] & 1) and b[1] >= a[1] and b[1] < 0.5 * Nx * Ny * Nz
        assert np.array_equal(a[5], b[5]) and np.array_equal(a[2], b[2])
        assert np.array_equal(a[4], b[4]) and np.array_equal(a[3], b[3])


@pytest.mark.parametrize("Nx,Ny,Nz,field", [(64, 32, 20, "shared"), (128, 128, 12, "shared"), (128, 128, 12, "near"),
                                             (128, 64, 64, "shared"), (64, 64, 24, "own"), (192, 128, 16, "own")])
def test_flow_kernel_is_bit_identical_to_launch_per_pass(gpu, monkeypatch, Nx, Ny, Nz, field):
    """Round 5: the intermediate passes of an application of the scan preconditioner run as ONE launch (tsx_k_pcs_flow: work
    items (pass, tile) behind a ticket counter, an item waiting for its four neighbour tiles' progress words; records handed
    between workgroups through sc1 stores / loads).  Same arithmetic per cell as a launch per pass (TSX_PC_FLOW=0): M^-1 v and
    whole solves must be bit-identical -- any record read before its producer's store became visible, or overwritten while a
    neighbour still read it, shows here.  Fields: blocks shared bit-identically (per-block records behind an index), grouped
    near-identical blocks (entry-major records, fetched by lane groups through LDS), every cell its own records."""
    P = synthetic.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz)
    coeff = P["coeff"]
    if field == "near":
        coeff = (coeff * (1 + 1e-3 * np.random.default_rng(11).random(coeff.shape))).astype(np.float32)
    if field == "own":
        monkeypatch.setenv("TSX_DEDUP", "0")
    v = np.random.default_rng(12).standard_normal(P["b"].shape)
    out = {}
    for env in ("1", "0"):
        monkeypatch.setenv("TSX_PC_FLOW", env)
        s = DiffuseSolver("3_10", Nz, Nx, Ny)
        s.set_coeffs(coeff, P["l1d"], P["a11"], P["a12"], P["albedo"])
        res = [s.pc_apply(v, pc=3, sweeps=sw, mixed=True) for sw in (27, 27, 9, 5)]
        for rtol in (1e-5, 1e-9):
            x = np.zeros(s.vec_shape)
            info = s.solve(P["b"], x, rtol=rtol, atol=1e-30)
            assert info.reason == 2
            res += [x, np.asarray(info.res_hist)]
        out[env] = res
        s.close()
    for a, b in zip(out["1"], out["0"]):
        assert np.isfinite(a).all() and np.array_equal(a, b)


@pytest.mark.parametrize("Nx,Ny,Nz", [(64, 32, 20), (128, 64, 16)])
def test_flow_kernel_with_self_neighbour_faces_is_bit_identical(gpu, monkeypatch, Nx, Ny, Nz):
    """One rank whose four neighbours are itself over the peer transport (force_halo: every face goes through the mailbox, the
    set-up of scripts/shard_study.py): the flow kernel with its faces inside the launch (tags per row / column, no
    acknowledgements, tsx_k_pcs_flow FPEER) against a launch per pass with the passes exchanging their records themselves
    (TSX_FLOW_PEER=0): same arithmetic, bit-identical M^-1 v and solves; and the periodic domain without any halo: the same solution."""
    P = synthetic.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz)
    v = np.random.default_rng(3).standard_normal(P["b"].shape)
    out = {}
    for mode in ("flow", "flow_fences", "flow_lean", "launches", "periodic"):
        monkeypatch.setenv("TSX_FLOW_PEER", "0" if mode == "launches" else "1")
        # round 6: the faces inside the LEAN body too (four waves per SIMD, the face columns send after the scan): what shards whose
        # passes are not resident at once run; forced here on a small domain
        if mode == "flow_lean":
            monkeypatch.setenv("TSX_FLOW_FAT", "0")
        else:
            monkeypatch.delenv("TSX_FLOW_FAT", raising=False)
        s = DiffuseSolver("3_10", Nz, Nx, Ny, force_halo=mode != "periodic")
        if mode != "periodic":
            s.comm_peer_init(lambda blob: [blob])
        if mode == "flow_fences":   # full system-scope fences around the flags and tags (what a failed self test switches on)
            s.comm_peer_set_fences(1)
        s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
        res = [s.pc_apply(v, pc=3, sweeps=sw, mixed=True) for sw in (27, 27, 9)]
        x = np.zeros(s.vec_shape)
        info = s.solve(P["b"], x, rtol=1e-8, atol=1e-30)
        assert info.reason == 2
        fl = s.flow_info()
        assert fl["in_use"] == (mode != "launches"), (mode, fl)
        if mode in ("flow", "flow_lean"):
            assert fl["fat"] == (mode == "flow"), (mode, fl)
        out[mode] = res + [x, np.asarray(info.res_hist)]
        s.close()
    for a, b, c, d in zip(out["flow"], out["launches"], out["flow_fences"], out["flow_lean"]):
        assert np.isfinite(a).all() and np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(a, d)
    # the periodic domain: the same M^-1 up to the precision of the records at the faces (bf16 through the mailbox also where the
    # last pass reads fp32 records in place), the same solution
    assert np.abs(out["flow"][0] - out["periodic"][0]).max() <= 2e-2 * np.abs(out["periodic"][0]).max()
    assert np.abs(out["flow"][3] - out["periodic"][3]).max() <= 1e-6 * np.abs(out["periodic"][3]).max()


def test_flow_kernel_restarts_its_epoch_before_it_wraps(gpu):
    """The flow kernel's progress words and granule tags are numbers that only grow (epoch + pass index); before the host's bound on
    the epoch reaches 2^30 the state, the words and the tags restart from zero in stream order (flow_ensure).  With the bound
    lowered to 60 (TSX_FLOW_EPOCH_LIMIT, read once per process: a subprocess) the restart happens every second application:
    repeated applications and solves must stay bit-identical to a launch per pass -- on a domain that uses the progress words
    and on one that uses granules."""
    import os
    import subprocess
    import sys

    code = r"""
import os, sys, numpy as np
sys.path.insert(0, sys.argv[1])
from tenstream_amd import DiffuseSolver, synthetic
for Nx, Ny, Nz in ((128, 64, 12), (64, 32, 16)):
    P = synthetic.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz)
    v = np.random.default_rng(2).standard_normal(P["b"].shape)
    out = {}
    for env in ("1", "0"):
        os.environ["TSX_PC_FLOW"] = env
        s = DiffuseSolver("3_10", Nz, Nx, Ny)
        s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
        res = [s.pc_apply(v, pc=3, sweeps=27, mixed=True) for _ in range(7)]
        for _ in range(3):
            x = np.zeros(s.vec_shape)
            assert s.solve(P["b"], x, rtol=1e-9, atol=1e-30).reason == 2
            res.append(x)
        out[env] = res
        s.close()
    assert all(np.array_equal(a, b) for a, b in zip(out["1"], out["0"])), (Nx, Ny)
    assert all(np.array_equal(out["1"][0], a) for a in out["1"][1:7])
print("ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code, root], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSX_FLOW_EPOCH_LIMIT="60"))
    assert p.returncode == 0 and "ok" in p.stdout, p.stderr[-2000:]
