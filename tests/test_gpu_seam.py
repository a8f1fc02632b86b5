"""The seam-level entries beside tsx_diff_*: the direct seam of `pprts()` (src/pprts.F90:2698-2755: set_dir_coeff +
explicit_edir fed with solver%dir2dir in the reference's layout), setup_b on its own (src/pprts.F90:4641-4987), and every
vector of the seam in real32 (ireals of a single-precision TenStream build).  Checker: the oracle's restatements
(explicit_edir, setup_b_solar / _thermal) on the same coefficient blocks; no LUT and no optical properties reach the device
on this path -- only what the reference's branch has in hand."""
import numpy as np
import pytest

from oracle import oracle as O
from tenstream_amd import DiffuseSolver, lut, synthetic
from tenstream_amd._lib import TsxError

pytestmark = pytest.mark.gpu
DX, DY = 100.0, 80.0


def _case(solver, Nx, Ny, Nz, phi0, theta0, tall_top):
    """coefficient blocks exactly as alloc_coeff_dir2dir / dir2diff / diff2diff deliver them (oracle lookups of the synthetic
    tables), 1-D layers on top with their Eddington coefficients"""
    from tenstream_amd.pprts import eddington_coeff_ec

    S, D = (3, 10) if solver == "3_10" else (8, 16)
    kabs, ksca, g = synthetic.cloud_field(Nx, Ny, Nz, seed=9)
    kabs *= 20.0
    kabs, ksca, g = synthetic.delta_scale(kabs, ksca, g)
    dz = np.full((Ny, Nx, Nz), 50.0)
    dz[:, :, :tall_top] = 400.0
    l1d = np.zeros(Nz, dtype=np.uint8)
    l1d[:tall_top] = 1
    sun = O.suninfo(phi0, theta0)
    dax = lut.direct_axes()
    Tdir, Sdir = lut.synthetic_direct_tables(dax, solver)
    LT, LS = O.make_lut(dax, Tdir), O.make_lut(dax, Sdir)
    Ld = O.make_lut(lut.diffuse_axes(solver), lut.synthetic_diffuse_table(solver))
    t = O.alloc_coeff_dir(LT, True, kabs, ksca, g, dz, DX, sun, l1d, S=S, D=D)
    sd = O.alloc_coeff_dir(LS, False, kabs, ksca, g, dz, DX, sun, l1d, S=S, D=D)
    c = O.alloc_coeff_diff2diff(Ld, kabs, ksca, g, dz, DX, l1d)
    ext = np.maximum(np.finfo(np.float64).tiny, kabs + ksca)
    mu0 = max(np.cos(np.deg2rad(theta0)), 0.0)
    a11, a12, a13, a23, a33 = (np.ascontiguousarray(a) for a in eddington_coeff_ec(dz * ext, ksca / ext, g, mu0))
    albedo = 0.05 + 0.1 * np.random.default_rng(2).random((Ny, Nx))
    return dict(S=S, D=D, kabs=kabs, dz=dz, l1d=l1d, sun=sun, dir2dir=np.ascontiguousarray(t), dir2diff=np.ascontiguousarray(sd),
                diff2diff=np.ascontiguousarray(c), a11=a11, a12=a12, a13=a13, a23=a23, a33=a33, albedo=albedo,
                lay=O.layout(solver, Nz, Nx, Ny), dlay=O.dir_layout(solver))


@pytest.mark.parametrize("solver,phi0,theta0,tall_top", [("3_10", 200.0, 50.0, 0), ("3_10", 30.0, 20.0, 2), ("8_16", 300.0, 35.0, 1)])
def test_direct_seam_equals_explicit_edir_and_setup_b(gpu, solver, phi0, theta0, tall_top):
    Nx, Ny, Nz = 9, 7, 8
    Q = _case(solver, Nx, Ny, Nz, phi0, theta0, tall_top)
    s = DiffuseSolver(solver, Nz, Nx, Ny)
    s.set_angles(phi0, theta0)
    s.dir_set_coeffs(Q["dir2dir"], Q["dir2diff"], Q["l1d"], DX, DY, a33=Q["a33"], a13=Q["a13"], a23=Q["a23"])
    edir = np.zeros((Ny, Nx, Nz + 1, Q["S"]))
    niter, res, conv = s.dir_solve(900.0, edir, rtol=1e-14, atol=1e-12, maxit=500)
    assert conv and niter >= 2
    want, _ = O.explicit_edir(Q["lay"], Q["dlay"], Q["sun"], Q["dir2dir"], Q["l1d"], Q["a33"], 900.0, DX, DY, rtol=1e-14, atol=1e-12,
                              maxit=500)
    assert np.abs(edir - want).max() <= 1e-9 * np.abs(want).max()
    # the reference's default stop rule from a zero iterate, then a warm start from that iterate: the beam handed in is v0
    e2 = np.zeros_like(edir)
    n2, _, c2 = s.dir_solve(900.0, e2)
    assert c2 and np.abs(e2 - want).max() <= 1e-4 * np.abs(want).max()
    n3, _, c3 = s.dir_solve(900.0, e2)
    assert c3 and n3 <= n2 and n3 <= 2
    # an iteration limit that does not suffice is reported, not raised
    e4 = np.zeros_like(edir)
    n4, _, c4 = s.dir_solve(900.0, e4, rtol=1e-14, atol=1e-30, maxit=1)
    assert n4 == 1 and not c4
    # setup_b from an identical beam (the oracle's), with the albedo handed over; and from the device's own beam
    b_want = O.setup_b_solar(Q["lay"], Q["dlay"], Q["sun"], Q["dir2diff"], Q["l1d"], Q["a13"], Q["a23"], Q["albedo"], want)
    b = s.setup_b_solar(np.zeros(s.vec_shape), edir=want, albedo=Q["albedo"])
    assert np.abs(b - b_want).max() <= 1e-13 * np.abs(b_want).max()
    s.dir_solve(900.0, edir, rtol=1e-14, atol=1e-12, maxit=500)
    b2 = s.setup_b_solar(np.zeros(s.vec_shape))
    assert np.abs(b2 - b_want).max() <= 1e-9 * np.abs(b_want).max()
    # real32 ireals: the beam and b cross the seam as float arrays
    e32 = np.zeros(edir.shape, dtype=np.float32)
    _, _, c32 = s.dir_solve(900.0, e32, rtol=1e-14, atol=1e-12, maxit=500)
    assert c32 and np.array_equal(e32, edir.astype(np.float32))
    b32 = s.setup_b_solar(np.zeros(s.vec_shape, dtype=np.float32))
    assert np.array_equal(b32, b2.astype(np.float32))
    s.close()


def test_direct_seam_rejects_blocks_that_real32_cannot_hold(gpu):
    Nx, Ny, Nz = 6, 4, 5
    Q = _case("3_10", Nx, Ny, Nz, 180.0, 40.0, 0)
    s = DiffuseSolver("3_10", Nz, Nx, Ny)
    with pytest.raises(TsxError):   # the sweep order needs the sun first
        s.dir_set_coeffs(Q["dir2dir"], None, Q["l1d"], DX, DY)
    s.set_angles(180.0, 40.0)
    bad = Q["dir2dir"].copy()
    bad[1, 2, 3, 0] = 0.1   # not a real32 value: did not come from the real32 tables (src/pprts.F90:3121-3143)
    with pytest.raises(TsxError):
        s.dir_set_coeffs(bad, None, Q["l1d"], DX, DY)
    s.dir_set_coeffs(Q["dir2dir"].astype(np.float32), None, Q["l1d"], DX, DY)   # coeff_kind 4
    e = np.zeros((Ny, Nx, Nz + 1, 3))
    assert s.dir_solve(500.0, e)[2]
    with pytest.raises(TsxError):   # no dir2diff was handed over
        s.setup_b_solar(np.zeros(s.vec_shape))
    s.set_angles(10.0, 40.0)        # set_angles invalidates the coefficients like the reference does (src/pprts.F90:1100-1116)
    with pytest.raises(TsxError):
        s.dir_solve(500.0, e)
    s.close()


def test_direct_seam_ignores_what_the_blocks_hold_in_1d_layers(gpu):
    """The reference never assigns solver%dir2dir / dir2diff in 1-D layers (alloc_coeff_dir2dir fills `if (.not. atm%l1d(...))`
    only, src/pprts.F90:3131): whatever that memory holds -- NaN, values real32 cannot hold -- must neither be rejected as lossy
    nor reach the beam (round-4 ADVICE).  Same beam and source term as with the oracle's clean blocks; and a call that IS rejected
    leaves no coefficients of an earlier call behind."""
    Nx, Ny, Nz, phi0, theta0 = 8, 6, 8, 30.0, 20.0
    Q = _case("3_10", Nx, Ny, Nz, phi0, theta0, 2)   # the two top layers are 1-D
    assert Q["l1d"].sum() == 2
    s = DiffuseSolver("3_10", Nz, Nx, Ny)
    s.set_angles(phi0, theta0)
    s.dir_set_coeffs(Q["dir2dir"], Q["dir2diff"], Q["l1d"], DX, DY, a33=Q["a33"], a13=Q["a13"], a23=Q["a23"])
    e0 = np.zeros((Ny, Nx, Nz + 1, Q["S"]))
    assert s.dir_solve(900.0, e0, rtol=1e-14, atol=1e-12, maxit=500)[2]
    b0 = s.setup_b_solar(np.zeros(s.vec_shape), albedo=Q["albedo"])
    t, sd = Q["dir2dir"].copy(), Q["dir2diff"].copy()
    k1 = np.flatnonzero(Q["l1d"])
    t[:, :, k1, :] = np.nan          # (j, i, k, coefficient): uninitialised memory in the 1-D layers
    sd[:, :, k1, :] = 0.1            # ... or values no real32 table delivered
    s.dir_set_coeffs(t, sd, Q["l1d"], DX, DY, a33=Q["a33"], a13=Q["a13"], a23=Q["a23"])
    e1 = np.zeros_like(e0)
    assert s.dir_solve(900.0, e1, rtol=1e-14, atol=1e-12, maxit=500)[2]
    b1 = s.setup_b_solar(np.zeros(s.vec_shape), albedo=Q["albedo"])
    assert np.isfinite(e1).all() and np.array_equal(e1, e0) and np.array_equal(b1, b0)
    # a lossy value in a 3-D layer is still rejected -- and the rejected call leaves nothing usable behind
    bad = Q["dir2dir"].copy()
    bad[1, 2, int(np.flatnonzero(Q["l1d"] == 0)[0]), 0] = 0.1
    with pytest.raises(TsxError):
        s.dir_set_coeffs(bad, Q["dir2diff"], Q["l1d"], DX, DY, a33=Q["a33"], a13=Q["a13"], a23=Q["a23"])
    with pytest.raises(TsxError):
        s.dir_solve(900.0, e1)
    with pytest.raises(TsxError):
        s.setup_b_solar(np.zeros(s.vec_shape))
    s.close()


@pytest.mark.parametrize("solver", ["3_10", "8_16"])
@pytest.mark.parametrize("srfc", [False, True])
def test_thermal_source_at_the_seam(gpu, solver, srfc):
    Nx, Ny, Nz = 8, 6, 7
    Q = _case(solver, Nx, Ny, Nz, 0.0, 0.0, 1)
    s = DiffuseSolver(solver, Nz, Nx, Ny)
    b = np.zeros(s.vec_shape)
    rng = np.random.default_rng(4)
    planck = np.linspace(2.0, 6.0, Nz + 1)[None, None, :] * (1 + 0.05 * rng.random((Ny, Nx, 1)))
    planck = np.ascontiguousarray(np.broadcast_to(planck, (Ny, Nx, Nz + 1)))
    skin = np.ascontiguousarray(planck[:, :, -1] * 1.2) if srfc else None
    with pytest.raises(TsxError):   # emissivities come from the diffuse blocks
        s.setup_b_thermal(b, planck, Q["kabs"], Q["dz"], DX, DY)
    s.set_coeffs(Q["diff2diff"], Q["l1d"], Q["a11"], Q["a12"], Q["albedo"])
    s.setup_b_thermal(b, planck, Q["kabs"], Q["dz"], DX, DY, planck_srfc=skin)
    want = O.setup_b_thermal(Q["lay"], Q["diff2diff"], Q["l1d"], Q["a11"], Q["a12"], Q["albedo"], planck, Q["kabs"], Q["dz"], DX, DY,
                             planck_srfc=skin)
    assert np.abs(b - want).max() <= 1e-13 * np.abs(want).max()
    # b feeds the diffuse seam directly
    x = np.zeros(s.vec_shape)
    info = s.solve(b, x, rtol=1e-10, atol=1e-30)
    x_ref, _ = O.solve_ilu(Q["lay"], Q["diff2diff"], Q["l1d"], Q["a11"], Q["a12"], Q["albedo"], want, rtol=1e-12, atol=1e-30, maxit=3000)
    assert info.reason == 2 and np.abs(x - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
    s.close()


@pytest.mark.parametrize("solver", ["3_10", "8_16"])
@pytest.mark.parametrize("on_device", [False, True])
def test_real32_vectors_cross_the_diffuse_seam_as_they_are(gpu, solver, on_device):
    """ireals = real32 (src/data_parameters.F90; the reference's CI builds it): x, y, b are float arrays at the boundary.  The
    device widens them, computes what the real64 entries compute, and narrows the result -- bit-identical to doing the two
    conversions on the host around the real64 entry."""
    import torch

    P = synthetic.make_problem(solver, Nx=10, Ny=8, Nz=6, n1d=1)
    s = DiffuseSolver(solver, 6, 10, 8)
    s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    x32 = np.random.default_rng(1).standard_normal(s.vec_shape).astype(np.float32)
    y64 = s.apply(x32.astype(np.float64))
    b32 = P["b"].astype(np.float32)
    sol64 = np.zeros(s.vec_shape)
    i64 = s.solve(b32.astype(np.float64), sol64, rtol=1e-9, atol=1e-30)
    if on_device:
        dev = torch.device("cuda", 0)
        y32 = s.apply(torch.tensor(x32, device=dev)).cpu().numpy()
        sol = torch.zeros(s.vec_shape, dtype=torch.float32, device=dev)
        i32 = s.solve(torch.tensor(b32, device=dev), sol, rtol=1e-9, atol=1e-30)
        sol32 = sol.cpu().numpy()
    else:
        y32 = s.apply(x32)
        sol32 = np.zeros(s.vec_shape, dtype=np.float32)
        i32 = s.solve(b32, sol32, rtol=1e-9, atol=1e-30)
    assert y32.dtype == np.float32 and np.array_equal(y32, y64.astype(np.float32))
    assert i32.reason == 2 and i32.niter == i64.niter and sol32.dtype == np.float32
    assert np.array_equal(sol32, sol64.astype(np.float32))
    # a real32 warm start: the float guess is widened, not dropped (one iteration or none is left to do)
    again = sol32.copy()
    i2 = s.solve(b32, again)   # reference default tolerances: the rounded solution's residual is below atol at once
    assert i2.reason in (2, 3) and i2.niter <= 1
    s.close()


def _seam_worker(rank, world, port, solver, Nx, Ny, Nz, phi0, theta0, transport, ret):
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist

    from tenstream_amd import coord, hostcomm

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("TSX_PEER_TIMEOUT_S", "10")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Q = _case(solver, Nx, Ny, Nz, phi0, theta0, 1)   # the global domain's blocks (same on every rank)
        co = coord.coord(rank, world, Nx, Ny)
        sl = (slice(co.ys, co.ys + co.ym), slice(co.xs, co.xs + co.xm))
        loc = lambda a: np.ascontiguousarray(a[sl])
        s = DiffuseSolver(solver, Nz, co.xm, co.ym, xs=co.xs, ys=co.ys, glob_xm=Nx, glob_ym=Ny, rank=rank, nranks=world,
                          neighbors=(co.west, co.east, co.south, co.north), device=0)
        if transport == "peer":
            hostcomm.attach_peer(s)
        else:
            hostcomm.attach(s, rank)
        s.set_angles(phi0, theta0)
        s.dir_set_coeffs(loc(Q["dir2dir"]), loc(Q["dir2diff"]), Q["l1d"], DX, DY, a33=loc(Q["a33"]), a13=loc(Q["a13"]), a23=loc(Q["a23"]))
        edir = np.zeros((co.ym, co.xm, Nz + 1, Q["S"]))
        niter, res, conv = s.dir_solve(700.0, edir, rtol=1e-12, atol=1e-6, maxit=60)   # (|edir| ~ 7e6 W: atol 1e-6 is 1e-13 of it)
        b = s.setup_b_solar(np.zeros(s.vec_shape), albedo=loc(Q["albedo"]))   # the beam the sweep left on the device, its exchanged faces
        refused = False
        try:   # a beam handed in on several ranks would lack the faces the sweep exchanged
            s.setup_b_solar(np.zeros(s.vec_shape), edir=edir)
        except TsxError:
            refused = True
        want_e, _ = O.explicit_edir(Q["lay"], Q["dlay"], Q["sun"], Q["dir2dir"], Q["l1d"], Q["a33"], 700.0, DX, DY, rtol=1e-14, atol=1e-12,
                                    maxit=500)
        want_b = O.setup_b_solar(Q["lay"], Q["dlay"], Q["sun"], Q["dir2diff"], Q["l1d"], Q["a13"], Q["a23"], Q["albedo"], want_e)
        ret[rank] = (conv, niter, float(np.abs(edir - want_e[sl]).max() / np.abs(want_e).max()),
                     float(np.abs(b - want_b[sl]).max() / np.abs(want_b).max()), refused)
        s.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,solver,Nx,Ny,phi0,theta0,transport", [(2, "3_10", 10, 8, 200.0, 50.0, "host"), (4, "3_10", 12, 10, 30.0, 35.0, "peer"),
                                                                         (2, "8_16", 8, 6, 300.0, 40.0, "host")])
def test_direct_seam_on_several_ranks_equals_the_global_oracle(gpu, world, solver, Nx, Ny, phi0, theta0, transport):
    """tsx_dir_set_coeffs / tsx_dir_solve / tsx_setup_b_solar with the domain split over 2 / 4 rank processes (sharing cuda:0):
    every sweep exchanges the downwind faces (exchange_direct_boundary, src/pprts_explicit.F90:1076-1140), the stop rule is on
    the mean over ranks of the local norms -- against the oracle's explicit_edir and setup_b on the global domain."""
    from test_gpu_multirank import _spawn

    ret = _spawn(_seam_worker, world, (solver, Nx, Ny, 7, phi0, theta0, transport))
    its = {v[1] for v in ret.values()}
    assert len(its) == 1   # one (all-reduced) residual decides for every rank
    for conv, niter, e_edir, e_b, refused in ret.values():
        assert conv and niter <= 40 and e_edir <= 1e-9 and e_b <= 1e-9 and refused, (conv, niter, e_edir, e_b, refused)
