"""CPU tests: the oracle against the reference's own known-answer vectors (tests/golden/*.json, written by
tests/golden/make_golden.py from the numbers in the reference's pFUnit suites)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def test_eddington_coeff_ec_matches_reference_vectors():
    d = load("eddington_ec.json")
    assert len(d["cases"]) == 13
    for c in d["cases"]:
        out = O.eddington_coeff_ec(*c["inp"])
        np.testing.assert_allclose(out, c["targ"], rtol=0, atol=d["tol"], err_msg=f"line {c['line']}")
        # the reference's side conditions (tests/eddington/test_delta_eddington.F90:25-28)
        assert sum(out[:2]) < 1 + d["tol"] and sum(out[2:]) < 1 + d["tol"]
        assert min(out) >= 0 and max(out) <= 1


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_search_sorted_bisection_matches_reference_vectors(dtype):
    for c in load("search.json")["cases"]:
        for val, want in c["pairs"]:
            assert O.search_sorted_bisection(c["arr"], val, dtype) == want


def test_interp_vec_nd_matches_reference_vectors():
    d = load("interp.json")
    for c in d["cases"]:
        db = np.array(c["db"], dtype=np.float32)[:, None]
        for pti, want in c["queries"]:
            got = O.interp_vec_nd(pti, db, c["shape"])[0]
            assert abs(got - want) <= d["tol"], (c["line"], pti, got, want)


def test_interp_vec_6d_matches_reference_vectors():
    s = load("interp.json")["six_d"]
    Nv, w = s["Nv"], np.array(s["weights"], dtype=np.float32)
    idx = np.indices((Nv,) * 6).reshape(6, -1).T[:, ::-1] + 1  # first dim fastest, 1-based
    x1 = (idx.astype(np.float32) * w).sum(axis=1)
    db = np.stack([x1, 2 * x1], axis=1).astype(np.float32)
    rng = np.random.default_rng(0)
    for _ in range(200):  # the reference loops all (Nv-1)^6 corners; a random subset pins the same property
        p = rng.integers(1, Nv, size=6).astype(np.float32)
        got = O.interp_vec_nd(p, db, [Nv] * 6)
        assert got[0] == (p * w).sum() and got[1] == 2 * (p * w).sum()
        got = O.interp_vec_nd(p + 0.5, db, [Nv] * 6)
        assert got[0] == ((p + 0.5) * w).sum() and got[1] == (2 * (p + 0.5) * w).sum()


def test_interp_lattice_snapping():
    """src/interpolation.F90:546-556: fractional parts < 1e-3 or > 1 - 1e-3 snap to the node (nint)."""
    db = np.array([0.0, 10.0, 20.0], dtype=np.float32)[:, None]
    assert O.interp_vec_nd([1.0005], db, [3])[0] == 0.0
    assert O.interp_vec_nd([1.9995], db, [3])[0] == 10.0
    assert O.interp_vec_nd([1.5], db, [3])[0] == 5.0
    assert abs(O.interp_vec_nd([1.002], db, [3])[0] - 0.02) < 1e-5


def test_boxmc_known_answer_block_structure_and_surrogate():
    """The reference's known-answer 3_10 block (tests/test_boxmc_3_10) is energy conserving up to absorption and
    the synthetic surrogate reproduces its sparsity pattern and values within the Monte-Carlo tolerance band x5."""
    from tenstream_amd import synthetic as S

    b = load("boxmc_3_10_block.json")
    T = np.array(b["S_by_src"])  # [src, dst]
    tau = (b["kabs"] + b["ksca"]) * b["dz"]
    assert np.all(T.sum(axis=1) <= 1.0) and np.all(T.sum(axis=1) > np.exp(-3 * tau))
    c = S.diff2diff_surrogate("3_10", np.float32(tau), np.float32(0.0), b["dz"] / b["dx"], np.float32(0.0))
    Tsur = c.reshape(10, 10).T  # c[dst*D+src] -> [src, dst]
    assert np.array_equal(Tsur > 0, T > 0)
    assert np.abs(Tsur - T).max() < 0.03


def test_matrix_free_equals_assembled_3_10():
    """(i) of SURVEY 8(c): op_mat_mult_ediff (pprts_shell.F90) == set_diff_coeff CSR (pprts.F90) on random x."""
    from tenstream_amd import synthetic as S

    P = S.make_problem("3_10", Nx=7, Ny=5, Nz=6, n1d=2)
    lay = O.layout("3_10", 6, 7, 5)
    c64 = P["coeff"].astype(np.float64)
    x = np.random.default_rng(0).standard_normal((5, 7, 7, 10))
    y = O.diff_apply(lay, c64, P["l1d"], P["a11"], P["a12"], P["albedo"], x)
    A = O.assemble_csr(lay, c64, P["l1d"], P["a11"], P["a12"], P["albedo"])
    assert np.abs(A @ x.ravel() - y.ravel()).max() < 1e-13 * np.abs(y).max()
    # A = I - T with T >= 0 and column sums of T <= 1 (energy conservation incl. albedo <= 1)
    Tm = -(A - __import__("scipy.sparse").sparse.identity(A.shape[0]))
    assert Tm.min() >= 0 and np.asarray(Tm.sum(axis=0)).max() <= 1 + 1e-6


def test_8_16_assembled_vs_shell_differ_only_on_surface_rows():
    """SURVEY a5 caveat: for multi-stream tops the shell form puts albedo on the inv_dof pair only."""
    from tenstream_amd import synthetic as S

    P = S.make_problem("8_16", Nx=4, Ny=3, Nz=3)
    lay = O.layout("8_16", 3, 4, 3)
    c64 = P["coeff"].astype(np.float64)
    x = np.random.default_rng(0).standard_normal((3, 4, 4, 16))
    y = O.diff_apply(lay, c64, P["l1d"], P["a11"], P["a12"], P["albedo"], x)
    A = O.assemble_csr(lay, c64, P["l1d"], P["a11"], P["a12"], P["albedo"])
    d = np.abs((A @ x.ravel()).reshape(y.shape) - y)
    assert d[:, :, :3, :].max() < 1e-13 and d[:, :, 3, :].max() > 1e-3


def test_krylov_ilu_sor_share_the_fixed_point():
    """(ii) of SURVEY 8(c): BiCGStab (matrix-free), BiCGStab+ILU(0) (assembled) and the reference's explicit SOR
    converge to the same solution."""
    from tenstream_amd import synthetic as S

    P = S.make_problem("3_10", Nx=6, Ny=5, Nz=8, n1d=1)
    lay = O.layout("3_10", 8, 6, 5)
    args = (lay, P["coeff"].astype(np.float64), P["l1d"], P["a11"], P["a12"], P["albedo"], P["b"])
    x1, i1 = O.solve_matfree(*args, rtol=1e-12, atol=1e-30)
    x2, i2 = O.solve_ilu(*args, rtol=1e-12, atol=1e-30)
    x3, i3 = O.solve_sor(*args, rtol=1e-13, atol=1e-9)
    assert i1["reason"] == 2 and i2["reason"] == 2 and i3["converged"]
    assert i2["niter"] < i1["niter"]
    scale = np.abs(x1).max()
    assert np.abs(x1 - x2).max() < 1e-9 * scale and np.abs(x1 - x3).max() < 1e-7 * scale


def test_threaded_block_jacobi_baseline_reduces_to_the_one_rank_default():
    """bench.py's CPU baseline (FBCGS + PCBJACOBI/ILU(0), one thread per subdomain): 1x1 is the serial ILU path to the
    bit; 2x3 subdomains converge to the same solution with at least as many iterations."""
    from tenstream_amd import synthetic as S

    P = S.make_problem("3_10", Nx=8, Ny=9, Nz=6, n1d=1)
    lay = O.layout("3_10", 6, 8, 9)
    args = (lay, P["coeff"].astype(np.float64), P["l1d"], P["a11"], P["a12"], P["albedo"], P["b"])
    x1, i1 = O.solve_ilu(*args, rtol=1e-11, atol=1e-30)
    x2, i2 = O.solve_bjacobi_ilu_mt(*args, 1, 1, rtol=1e-11, atol=1e-30)
    x3, i3 = O.solve_bjacobi_ilu_mt(*args, 2, 3, rtol=1e-11, atol=1e-30)
    assert i1["reason"] == i2["reason"] == i3["reason"] == 2
    assert i1["niter"] == i2["niter"] and np.allclose(x1, x2, rtol=1e-12, atol=0)
    assert i3["niter"] >= i1["niter"]
    assert np.abs(x3 - x1).max() < 1e-8 * np.abs(x1).max()


def test_default_tolerances():
    """determine_ksp_tolerances (src/pprts_base.F90:1126-1131)."""
    assert O.default_tolerances(4, 4, 21) == (1e-5, pytest.approx(1e-4 * 4 * 4 * 21), 1000)
    assert O.default_tolerances(1, 1, 1, 1e-9)[1] == 1e-8


def test_delta_scale_matches_host_mirror():
    from tenstream_amd import synthetic as S

    rng = np.random.default_rng(1)
    for _ in range(50):
        ka, ks, g = rng.uniform(0, 1e-2), rng.uniform(0, 1e-1), rng.uniform(0, 0.99)
        a = O.delta_scale(ka, ks, g)
        b = S.delta_scale(ka, ks, g)
        np.testing.assert_allclose(a, [float(v) for v in b], rtol=1e-14)
    assert O.delta_scale(0.0, 0.0, 0.5) == (0.0, 0.0, 0.5)  # dtau < eps: untouched
    ka, ks, g = O.delta_scale(1e-3, 1e-2, 1.0)  # g == 1: all scattering is forward
    assert ks == 0.0 and g == 0.0 and ka == pytest.approx(1e-3)


def test_B_eff_limits():
    """src/schwarzschild.F90:36-67: thin limit is the mean, thick limit tends to the near value."""
    assert O.B_eff(2.0, 4.0, 1e-6) == pytest.approx(3.0, rel=1e-12)
    assert O.B_eff(2.0, 4.0, 1e3) == pytest.approx(4.0, rel=1e-2)


def test_B_eff_like_the_reference_test():
    """tests/test_schwarzschild/test_schwarzschild.F90:13-55 restated: Planck levels 5.67e-8 (280+k)^4 / pi, k = 1..11;
    tau = 1000 -> the near value (abs 0.1); tau = 0 -> the mean (abs 0.1); for tau = 10^(-10 .. 4) and both orderings the
    result lies between the two level values."""
    Blev = [5.67e-8 * float(280 + k) ** 4 / np.pi for k in range(1, 12)]
    between = lambda v, a, b: min(a, b) <= v <= max(a, b)
    for k in range(10):
        B = O.B_eff(Blev[k], Blev[k + 1], 1000.0)
        assert abs(B - Blev[k + 1]) <= 0.1 and between(B, Blev[k], Blev[k + 1])
        B = O.B_eff(Blev[k], Blev[k + 1], 0.0)
        assert abs(B - 0.5 * (Blev[k] + Blev[k + 1])) <= 0.1 and between(B, Blev[k], Blev[k + 1])
    for itau in range(-100, 41):
        tau = 10.0 ** (itau / 10.0)
        for k in range(10):
            assert between(O.B_eff(Blev[k], Blev[k + 1], tau), Blev[k], Blev[k + 1]), (tau, k)
            assert between(O.B_eff(Blev[k + 1], Blev[k], tau), Blev[k], Blev[k + 1]), (tau, k)


def test_flux_scaling_round_trip_like_the_reference_test():
    """tests/test_pprts_solution_vecscale (:55-100) restated for 3_10: a solution of ones in W/m2, scaled to W and back
    (gen_scale_*_flx_vec_arr, src/pprts.F90:3901-3987), keeps its norm; first layer 10x as thick like there."""
    nv, nx, ny, dx = 3, 3, 3, 100.0
    lay, dlay = O.layout("3_10", nv, nx, ny), O.dir_layout_3_10()
    dz = np.full((ny, nx, nv), dx)
    dz[:, :, 0] = 10 * dx
    ediff = np.ones((ny, nx, nv + 1, 10))
    edir = np.ones((ny, nx, nv + 1, 3))
    w_diff = O.scale_diff(lay, dz, dx, dx, False, ediff)
    w_dir = O.scale_dir(lay, dlay, dz, dx, dx, False, edir)
    assert np.allclose(w_diff[:, :, :, :2], dx * dx) and np.allclose(w_dir[:, :, :, 0], dx * dx)
    assert np.allclose(w_diff[:, :, 0, 2:], dx * 10 * dx) and np.allclose(w_dir[:, :, 1, 1:], dx * dx)  # side faces: dx * dz(k)
    back_diff = O.scale_diff(lay, dz, dx, dx, True, w_diff)
    back_dir = O.scale_dir(lay, dlay, dz, dx, dx, True, w_dir)
    assert np.linalg.norm(back_diff) == pytest.approx(np.linalg.norm(ediff), rel=1e-14)
    assert np.linalg.norm(back_dir) == pytest.approx(np.linalg.norm(edir), rel=1e-14)


def test_oracle_surface_emission_follows_planck_srfc():
    """set_thermal_source's surface term (src/pprts.F90:4958-4985): with atm%Bsrfc the upward streams at the ground get
    Bsrfc * Az * clamp(1 - albedo, 0, 1) * pi / streams, without it planck(ze) * Az * (1 - albedo) * pi / streams; nothing else
    in b changes.  planck_srfc = planck(ze) with albedo in [0, 1] reproduces the other branch exactly."""
    from tenstream_amd import synthetic

    for solver, ntop in (("3_10", 2), ("8_16", 8)):
        P = synthetic.make_problem(solver, Nx=5, Ny=4, Nz=6, n1d=1)
        lay = O.layout(solver, 6, 5, 4)
        rng = np.random.default_rng(3)
        planck = 3.0 + rng.random((4, 5, 7))
        kabs, dz = 1e-4 * (1 + rng.random((4, 5, 6))), np.full((4, 5, 6), 50.0)
        albedo = np.array(P["albedo"], dtype=np.float64).reshape(4, 5)
        albedo[1, 2], albedo[3, 0] = 1.3, -0.2   # both clamps
        args = (lay, P["coeff"].astype(np.float64), P["l1d"], P["a11"], P["a12"], albedo, planck, kabs, dz, 100.0, 70.0)
        b0 = O.setup_b_thermal(*args)
        srfc = 5.0 + rng.random((4, 5))
        b1 = O.setup_b_thermal(*args, planck_srfc=srfc)
        d = b1 - b0
        up = list(range(0, ntop, 2))
        want = (srfc * np.clip(1 - albedo, 0, 1) - planck[:, :, -1] * (1 - albedo)) * 100.0 * 70.0 * np.pi / (ntop // 2)
        assert np.abs(d[:, :, -1, up] - want[:, :, None]).max() <= 1e-12 * np.abs(want).max()
        d[:, :, -1, up] = 0
        assert np.abs(d).max() == 0.0
        albedo2 = np.clip(albedo, 0, 1)
        a2 = args[:5] + (albedo2,) + args[6:]
        assert np.array_equal(O.setup_b_thermal(*a2), O.setup_b_thermal(*a2, planck_srfc=planck[:, :, -1]))
