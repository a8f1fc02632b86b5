"""The reference's own system-level cases restated on the device result:
 * config 1 of BASELINE.json: `examples/pprts/ex_pprts_ex1.F90:39-84` with `-Nx 4 -Ny 4 -Nz 20 -dtau_cld 0` (homogeneous
   clear-sky box, dtau 1, w0 .5, g 0, Ag .1, phi 180, theta 0, S0 1, dx = dy = dz = 100) through TenStream's own C-ABI
   `pprts_f2c_*` (libtsx_f2c.so) against the oracle's restatement of the pipeline on identical inputs;
 * `tests/test_pprts_symmetry/test_pprts_symmetry.F90:394-516` (`test_pprts_symmetry_ex1`): mirroring the sun azimuth by
   180 degrees mirrors the flux fields, atol 0.1 W/m2 at S0 = 1000;
 * config 3 of BASELINE.json (512x512x64 on 2x4 ranks): one rank's 256x128x64 block with rank faces routed through the halo
   buffers, and a 2-process x split, through size-independent properties."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O
from tenstream_amd import lut, synthetic
from tenstream_amd.pprts import PprtsSolver

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_luts(tmp_path, solver="3_10"):
    base = str(tmp_path / "LUT")
    D, tag = (10, "3_10") if solver == "3_10" else (16, "8_16")
    lut.write_mmap4(base + f"_diffuse_{D}.tau31.w020.aspect_zx23.g6.ds1000.nc.Sdiff.mmap4", lut.synthetic_diffuse_table(solver))
    dax = lut.direct_axes()
    Tdir, Sdir = lut.synthetic_direct_tables(dax, solver)
    dims = "tau{}.w0{}.aspect_zx{}.g{}.phi{}.theta{}".format(*[len(a) for a in dax])
    tpath = f"{base}_direct_{tag}.{dims}.ds1000.nc.Tdir.mmap4"
    lut.write_mmap4(tpath, Tdir)
    lut.write_mmap4(f"{base}_direct_{tag}.{dims}.ds1000.nc.Sdir.mmap4", Sdir)
    with open(tpath + ".axes", "w") as f:
        f.write(f"{len(dax)}\n")
        for a in dax:
            f.write(f"{len(a)} " + " ".join(repr(float(v)) for v in a) + "\n")
    return base, dims, dax, Tdir, Sdir


def test_config1_pprts_ex1_through_the_reference_c_abi(gpu, tmp_path, monkeypatch):
    from test_gpu_pipeline import _oracle_pipeline

    Nx, Ny, Nz = 4, 4, 20
    dx = dz = 100.0
    dtau, w0, g0, Ag, S0, phi0, theta0 = 1.0, 0.5, 0.0, 0.1, 1.0, 180.0, 0.0
    base, dims, dax, Tdir, Sdir = _write_luts(tmp_path)
    monkeypatch.setenv("LUT_BASENAME", base)
    monkeypatch.setenv("TSX_LUT_DIRECT_DIMS", dims)
    f2c = C.CDLL(os.path.join(ROOT, "tenstream_amd", "lib", "libtsx_f2c.so"))
    i32 = lambda v: C.byref(C.c_int(v))
    hhl = (np.float32(dz) * (Nz - np.arange(Nz + 1))).astype(np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    kabs = np.full((Ny, Nx, Nz), dtau / dz / Nz * (1.0 - w0), dtype=np.float32)   # ex_pprts_ex1: pprts_ex1.F90:85-87
    ksca = np.full((Ny, Nx, Nz), dtau / dz / Nz * w0, dtype=np.float32)
    g = np.full((Ny, Nx, Nz), g0, dtype=np.float32)
    f2c.pprts_f2c_init(0, i32(310), i32(Nz), i32(Nx), i32(Ny), C.byref(C.c_double(dx)), C.byref(C.c_double(dx)), fp(hhl),
                       C.byref(C.c_float(phi0)), C.byref(C.c_float(theta0)), i32(1))
    alb = C.c_float(Ag)
    f2c.pprts_f2c_set_global_optical_properties(Nz, Nx, Ny, C.byref(alb), fp(kabs), fp(ksca), fp(g), None)
    f2c.pprts_f2c_solve.argtypes = [C.c_int, C.c_float]
    f2c.pprts_f2c_solve(0, S0)
    edn, eup, edir = (np.zeros((Ny, Nx, Nz + 1), dtype=np.float32) for _ in range(3))
    abso = np.zeros((Ny, Nx, Nz), dtype=np.float32)
    f2c.pprts_f2c_get_result(Nz, Nx, Ny, fp(edn), fp(eup), fp(abso), fp(edir))
    f2c.pprts_f2c_destroy.argtypes = [C.c_int]
    f2c.pprts_f2c_destroy(0)

    # the oracle's pipeline on the same (float32-valued) inputs; P only provides the host mirror of the derived fields
    P = PprtsSolver(Nz, Nx, Ny, dx, dx, phi0, theta0)
    P.set_lut_diffuse(lut.synthetic_diffuse_table("3_10"), lut.diffuse_axes("3_10"))
    P.set_lut_direct(Tdir, Sdir, dax)
    dzf = np.broadcast_to(hhl[:-1].astype(np.float64) - hhl[1:].astype(np.float64), (Ny, Nx, Nz))
    P.set_optical_properties(float(np.float32(Ag)), kabs.astype(np.float64), ksca.astype(np.float64), g.astype(np.float64), dzf)
    I = dict(dx=dx, dy=dx, dax=dax, Tdir=Tdir, Sdir=Sdir)
    ref = _oracle_pipeline(P, I, Ag, S0, True)
    P.close()
    assert not P.l1d.any()   # dz / dx = 1 < twostr_ratio: a 3-D box
    # diffuse fluxes: the C-ABI solve stops at the reference's default tolerances (rtol 1e-5); direct: the sweep's rule
    for name, got, want, tol in (("edir", edir, ref["redir"], 2e-4), ("edn", edn, ref["edn"], 5e-4), ("eup", eup, ref["eup"], 5e-4),
                                 ("abso", abso, ref["abso"], 5e-4)):
        err = np.abs(got - want).max() / np.abs(want).max()
        assert err <= tol, (name, err)
    # homogeneous box, overhead sun: every column is the same, the direct beam is Beer-Lambert with dtau = 1 in total
    assert np.ptp(edn, axis=(0, 1)).max() <= 1e-5 * edn.max() and np.ptp(edir, axis=(0, 1)).max() <= 1e-6
    assert abs(edir[0, 0, 0] - 1.0) < 1e-6 and edir[0, 0, -1] < edir[0, 0, 0]


@pytest.mark.parametrize("phi_a,phi_b", [(10.0, 190.0), (100.0, 280.0)])
def test_pprts_symmetry_ex1(gpu, phi_a, phi_b):
    """tests/test_pprts_symmetry/test_pprts_symmetry.F90:394-516: 5x5x5 box (dx = dy = dz = 100, albedo 0, S0 = 1000,
    theta 60), one cloudy cell in the middle column; the sun at phi and at phi + 180 gives flux fields that are mirror
    images in x and y, within 0.1 W/m2.  rtol 1e-8 / atol 1e-30 as in that test's tenstream.options."""
    nxp = nyp = nv = 5
    dx = dz = 100.0
    kabs = np.full((nyp, nxp, nv), 1.0 / nv / dz)
    ksca = np.full((nyp, nxp, nv), 1.0 / nv / dz)
    g = np.zeros((nyp, nxp, nv))
    cx = cy = nxp // 2      # 0-based index of cx = int(nxp / 2) + 1
    kabs[cy, cx, 1] = 1.0 / dz
    ksca[cy, cx, 1] = 1.0 / dz
    g[cy, cx, 1] = 0.9
    dax = lut.direct_axes()
    Tdir, Sdir = lut.synthetic_direct_tables(dax)
    res = []
    for phi in (phi_a, phi_b):
        P = PprtsSolver(nv, nxp, nyp, dx, dx, phi, 60.0)
        P.set_lut_diffuse(lut.synthetic_diffuse_table("3_10"), lut.diffuse_axes("3_10"))
        P.set_lut_direct(Tdir, Sdir, dax)
        P.set_optical_properties(0.0, kabs, ksca, g, np.full((nyp, nxp, nv), dz))
        info = P.solve(1000.0, rtol=1e-8, atol=1e-30)
        assert info.reason == 2
        res.append(P.get_result())
        P.close()
    names = ("edn", "eup", "abso", "edir")
    for name, a, b in zip(names, res[0], res[1]):
        flipped = b[::-1, ::-1, :]
        scale = dz if name == "abso" else 1.0     # fdiv is W/m3 here; the reference's tolerance applies to W/m2 per layer
        assert np.abs(a - flipped).max() * scale <= 0.1, name
    # and the check is not vacuous: the un-mirrored fields differ by far more than the mirrored ones
    d_flip = max(np.abs(a - b[::-1, ::-1, :]).max() for a, b in zip(res[0][:2], res[1][:2]))
    d_same = max(np.abs(a - b).max() for a, b in zip(res[0][:2], res[1][:2]))
    assert d_same > 1.0 and d_same > 20.0 * d_flip


def test_config3_local_block_with_halo_faces(gpu):
    """One rank's share of config 3 (512x512x64 on 2x4: 256x128x64 per rank) with every rank face routed through the
    exchange buffers (force_halo: self neighbours, interior / frame split of the operator): linearity, the recurrence
    residual equals the true residual, default tolerances met -- and the same numbers as the wrapping single-rank run."""
    import torch

    Nx, Ny, Nz = 256, 128, 64
    dev = torch.device("cuda", 0)
    kabs, ksca, g = synthetic.cloud_field(Nx, Ny, Nz, seed=20240611)
    kabs, ksca, g = synthetic.delta_scale(kabs, ksca, g)
    b = torch.tensor(synthetic.solar_source("3_10", kabs, ksca, g, 50.0, 100.0, np.full((Ny, Nx), 0.1)), device=dev)
    from tenstream_amd import DiffuseSolver

    out = []
    for fh in (1, 0):
        s = DiffuseSolver("3_10", Nz, Nx, Ny, force_halo=fh)
        s.set_lut_diffuse(lut.synthetic_diffuse_table("3_10"), lut.diffuse_axes("3_10"))
        t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
        z = torch.zeros((Ny, Nx, Nz), dtype=torch.float64, device=dev)
        s.set_optprop(t(kabs), t(ksca), t(g), torch.full((Ny, Nx, Nz), 50.0, dtype=torch.float64, device=dev), 100.0,
                      torch.zeros(Nz, dtype=torch.uint8, device=dev), z, z, torch.full((Ny, Nx), 0.1, dtype=torch.float64, device=dev))
        gen = torch.Generator(device=dev).manual_seed(3)
        x = torch.randn(b.shape, dtype=torch.float64, device=dev, generator=gen)
        y = torch.randn(b.shape, dtype=torch.float64, device=dev, generator=gen)
        Ax, Ay = s.apply(x), s.apply(y)
        lin = s.apply(0.75 * x - 2.5 * y) - (0.75 * Ax - 2.5 * Ay)
        assert float(lin.abs().max()) <= 1e-13 * float(Ax.abs().max())
        sol = torch.zeros_like(b)
        info = s.solve(b, sol)
        assert info.reason in (2, 3)
        r = b - s.apply(sol)
        rn, bn = float(torch.linalg.vector_norm(r)), float(torch.linalg.vector_norm(b))
        assert abs(rn - info.rnorm) <= 1e-6 * bn
        rt, at, _ = s.default_tolerances()
        assert rn / bn <= rt or rn <= at
        out.append((Ax.clone(), sol.clone(), info.niter))
        s.close()
    assert float((out[0][0] - out[1][0]).abs().max()) <= 1e-13 * float(out[1][0].abs().max())   # same operator
    # the preconditioner drops the couplings across rank faces (block-Jacobi over ranks like PCBJACOBI): a few more
    # iterations, the same fixed point within the stop rule
    assert out[0][2] <= out[1][2] + 6
    assert float((out[0][1] - out[1][1]).abs().max()) <= 2e-4 * float(out[1][1].abs().max())


def test_config5_8_16_through_the_reference_c_abi(gpu, tmp_path, monkeypatch):
    """pprts_f2c_init with SOLVER_ID_PPRTS_8_16 = 816 (c_wrapper/f2c_solver_ids.h; f2c_pprts.F90:270-272): tables
    `_diffuse_16...` / `_direct_8_16...` from $LUT_BASENAME, a cloudy solar g-point; equal to the PprtsSolver-driven
    pipeline (which tests/test_gpu_pipeline.py pins to the oracle) to float32 rounding."""
    Nx, Ny, Nz, phi0, theta0 = 6, 5, 10, 200.0, 35.0
    base, dims, dax, Tdir, Sdir = _write_luts(tmp_path, "8_16")
    monkeypatch.setenv("LUT_BASENAME", base)
    monkeypatch.setenv("TSX_LUT_DIRECT_DIMS", dims)
    f2c = C.CDLL(os.path.join(ROOT, "tenstream_amd", "lib", "libtsx_f2c.so"))
    i32 = lambda v: C.byref(C.c_int(v))
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    hhl = (np.float32(40.41) * (Nz - np.arange(Nz + 1))).astype(np.float32)
    kabs = np.full((Ny, Nx, Nz), 1e-4, dtype=np.float32)
    ksca = np.full((Ny, Nx, Nz), 1e-4, dtype=np.float32)
    g = np.zeros((Ny, Nx, Nz), dtype=np.float32)
    for j in range(Ny):
        for i in range(Nx):
            if (i + 2 * j) % 5 < 2:
                ksca[j, i, Nz // 3:Nz // 2] = 2e-2
                kabs[j, i, Nz // 3:Nz // 2] = 1e-5
                g[j, i, Nz // 3:Nz // 2] = 0.85
    f2c.pprts_f2c_init(0, i32(816), i32(Nz), i32(Nx), i32(Ny), C.byref(C.c_double(100.0)), C.byref(C.c_double(100.0)), fp(hhl),
                       C.byref(C.c_float(phi0)), C.byref(C.c_float(theta0)), i32(1))
    alb = C.c_float(0.1)
    f2c.pprts_f2c_set_global_optical_properties(Nz, Nx, Ny, C.byref(alb), fp(kabs), fp(ksca), fp(g), None)
    f2c.pprts_f2c_solve.argtypes = [C.c_int, C.c_float]
    f2c.pprts_f2c_solve(0, 1000.0)
    edn, eup, edir = (np.zeros((Ny, Nx, Nz + 1), dtype=np.float32) for _ in range(3))
    abso = np.zeros((Ny, Nx, Nz), dtype=np.float32)
    f2c.pprts_f2c_get_result(Nz, Nx, Ny, fp(edn), fp(eup), fp(abso), fp(edir))
    f2c.pprts_f2c_destroy.argtypes = [C.c_int]
    f2c.pprts_f2c_destroy(0)
    P = PprtsSolver(Nz, Nx, Ny, 100.0, 100.0, np.float32(phi0), np.float32(theta0), solver="8_16")
    P.set_lut_diffuse(lut.synthetic_diffuse_table("8_16"), lut.diffuse_axes("8_16"))
    P.set_lut_direct(Tdir, Sdir, dax)
    dz = np.broadcast_to(hhl[:-1].astype(np.float64) - hhl[1:].astype(np.float64), (Ny, Nx, Nz))
    P.set_optical_properties(float(np.float32(0.1)), kabs.astype(np.float64), ksca.astype(np.float64), g.astype(np.float64), dz)
    P.solve(1000.0)
    want = P.get_result()
    P.close()
    for got, w in zip((edn, eup, abso, edir), want):
        assert np.abs(got - w.astype(np.float32)).max() <= 2e-6 * np.abs(w).max() + 1e-30
    assert np.ptp(edir[:, :, -1]) > 10.0   # the clouds cast shadows (W/m2)
