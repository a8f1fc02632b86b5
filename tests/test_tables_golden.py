"""Every table-like restatement on the path against fixtures derived from the reference's TEXT
(tests/golden/make_tables_golden.py interprets the Fortran assignment lines; the fixtures are data):

 * stream relabelling for the sun's quadrant: src/optprop.F90:1009-1045, 1186-1240, 1256-1266, 1268-1302
 * is_inward / area_divider / streams / inv_dof: src/pprts.F90:332-349, 413-425, 5739-5752
 * LUT axis presets and the dimension lists of LUT_3_10 / LUT_8_16: src/optprop_parameters.F90:91-245,
   src/optprop_base.F90:200-212, 228-240

CPU part: the oracle and the host-side mirrors.  GPU part (marked): the device path through the reference's own C-ABI
(`pprts_f2c_opp_get_coeff`) and the per-cell kernels, compared with the fixture DIRECTLY, not via the oracle.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from tenstream_amd import lut, synthetic

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _golden(name):
    with open(os.path.join(HERE, "golden", name)) as f:
        return json.load(f)


SYM = _golden("coeff_symmetry.json")["routines"]
ROUTINES = [  # fixture key, is_dir2dir, S, D
    ("dir3_to_diff10_coeff_symmetry", 0, 3, 10),
    ("dir8_to_diff16_coeff_symmetry", 0, 8, 16),
    ("dir2dir8_coeff_symmetry", 1, 8, 16),
    ("dir2dir_coeff_symmetry_none", 1, 3, 10),
]


# ------------------------------------------------------------------------------------------------ CPU: oracle and host mirrors
@pytest.mark.parametrize("name,is_dir2dir,S,D", ROUTINES)
def test_oracle_stream_relabelling_equals_the_reference_text(name, is_dir2dir, S, D):
    fx = SYM[name]
    n = fx["n"]
    for east in (0, 1):
        for north in (0, 1):
            got = O.dir_coeff_symmetry(np.arange(n, dtype=np.float32), is_dir2dir, S, D, east, north)
            assert got.astype(int).tolist() == fx["image"][f"e{east}n{north}"], (name, east, north)


def test_dir8_to_diff16_leaves_the_unassigned_blocks_alone():
    """The property the round-2 restatement got wrong (src/optprop.F90:1193-1236: the lines for dst 1,2,5,6,13-16 (east) and
    3,4,7-12 (north) are commented out): those blocks keep values AND source order."""
    img = np.array(SYM["dir8_to_diff16_coeff_symmetry"]["image"]["e1n0"]).reshape(16, 8)
    for d in (0, 1, 4, 5, 12, 13, 14, 15):
        assert img[d].tolist() == list(range(8 * d, 8 * d + 8))
    assert img[2].tolist() == [8 * 6 + q for q in (1, 0, 3, 2, 4, 5, 6, 7)]
    img = np.array(SYM["dir8_to_diff16_coeff_symmetry"]["image"]["e0n1"]).reshape(16, 8)
    for d in (2, 3, 6, 7, 8, 9, 10, 11):
        assert img[d].tolist() == list(range(8 * d, 8 * d + 8))
    assert img[0].tolist() == [8 * 4 + q for q in (2, 3, 0, 1, 4, 5, 6, 7)]


@pytest.mark.parametrize("solver,cls", [("3_10", "t_solver_3_10"), ("8_16", "t_solver_8_16")])
def test_stream_tables_equal_the_reference_text(solver, cls):
    fx = _golden("solver_tables.json")["solvers"][cls]
    lay = O.layout(solver, 4, 3, 3)
    assert (lay.ntop, lay.nside) == (fx["difftop"]["dof"], fx["diffside"]["dof"])
    assert list(lay.top_inward[: lay.ntop]) == fx["difftop"]["is_inward"]
    assert list(lay.side_inward[: lay.nside]) == fx["diffside"]["is_inward"]
    f = O.lib().orc_inv_dof
    f.restype = C.c_int
    assert [f(C.byref(lay), d) for d in range(lay.ntop)] == fx["inv_dof"]
    dl = O.dir_layout(solver)
    assert (dl.dtop, dl.dside, dl.top_div, dl.side_div) == (fx["dirtop"]["dof"], fx["dirside"]["dof"],
                                                           fx["dirtop"]["area_divider"], fx["dirside"]["area_divider"])
    assert all(fx["dirtop"]["is_inward"]) and all(fx["dirside"]["is_inward"])   # what the direct sweep's order assumes
    # host mirror used by the synthetic generator and the product's parity-of-index rule (tsx_dev.hpp:14)
    ntop, nside, top_in, side_in = synthetic.stream_layout(solver)
    assert (ntop, nside) == (lay.ntop, lay.nside)
    assert [int(v) for v in top_in] == fx["difftop"]["is_inward"] and [int(v) for v in side_in] == fx["diffside"]["is_inward"]
    assert fx["difftop"]["is_inward"] == [d & 1 for d in range(ntop)] and fx["diffside"]["is_inward"] == [d & 1 for d in range(nside)]
    assert fx["difftop"]["streams"] == ntop // 2 and fx["difftop"]["area_divider"] == 1


def _preset(fx, name):
    return np.frombuffer(bytes.fromhex("".join(fx["presets"][name]["f32_hex"])), dtype=np.float32)


def test_lut_axis_presets_equal_the_reference_text():
    fx = _golden("lut_presets.json")
    for solver, key in (("3_10", "LUT_3_10"), ("8_16", "LUT_8_16")):
        cfg = fx["configs"][key]
        assert [d["dim"] for d in cfg["diffconfig"]] == ["tau", "w0", "aspect_zx", "g"]
        assert [d["dim"] for d in cfg["dirconfig"]] == ["tau", "w0", "aspect_zx", "g", "phi", "theta"]
        want = [_preset(fx, d["preset"]) for d in cfg["diffconfig"]]
        got = lut.diffuse_axes(solver)
        assert len(got) == 4
        for a, b in zip(got, want):
            assert a.dtype == np.float32 and a.tobytes() == b.tobytes()
        full = lut.direct_axes(full=True)
        for a, d in zip(full, cfg["dirconfig"]):
            if "preset" in d:
                assert a.tobytes() == _preset(fx, d["preset"]).tobytes()
            else:   # populate_op_dim(vrange): linspace(lo, hi, n), src/optprop_base.F90
                assert a.tobytes() == np.linspace(d["vrange"][0], d["vrange"][1], d["n"], dtype=np.float32).tobytes()
        name = lut.diffuse_lut_filename("LUT", solver)
        assert name.endswith(".tau31.w020.aspect_zx23.g6.ds1000.nc.Sdiff.mmap4")


# ------------------------------------------------------------------------------------------------ GPU: the device path
def _index_valued_luts(tmp_path, solver):
    """Tables whose coefficient q of EVERY entry is q: a lookup on a lattice node returns arange -> after the relabelling the
    device's output IS the image the fixture holds."""
    S, D, tag = (3, 10, "3_10") if solver == "3_10" else (8, 16, "8_16")
    base = str(tmp_path / "LUT")
    dfx = lut.diffuse_axes(solver)
    nd = int(np.prod([len(a) for a in dfx]))
    lut.write_mmap4(base + f"_diffuse_{D}.tau31.w020.aspect_zx23.g6.ds1000.nc.Sdiff.mmap4",
                    np.tile(np.arange(D * D, dtype=np.float32), (nd, 1)))
    dax = lut.direct_axes()
    n = int(np.prod([len(a) for a in dax]))
    dims = "tau{}.w0{}.aspect_zx{}.g{}.phi{}.theta{}".format(*[len(a) for a in dax])
    tpath = f"{base}_direct_{tag}.{dims}.ds1000.nc.Tdir.mmap4"
    Tdir = np.tile(np.arange(S * S, dtype=np.float32), (n, 1))
    Sdir = np.tile(np.arange(S * D, dtype=np.float32), (n, 1))
    lut.write_mmap4(tpath, Tdir)
    lut.write_mmap4(f"{base}_direct_{tag}.{dims}.ds1000.nc.Sdir.mmap4", Sdir)
    with open(tpath + ".axes", "w") as f:
        f.write(f"{len(dax)}\n")
        for a in dax:
            f.write(f"{len(a)} " + " ".join(repr(float(v)) for v in a) + "\n")
    return base, dims, dfx, dax, Tdir, Sdir


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["3_10", "8_16"])
def test_f2c_opp_relabelling_equals_the_reference_text(gpu, tmp_path, monkeypatch, solver):
    """pprts_f2c_opp_get_coeff (c_wrapper/f2c_pprts.h:54-83) on index-valued tables against coeff_symmetry.json directly."""
    S, D, sid = (3, 10, 310) if solver == "3_10" else (8, 16, 816)
    base, dims, dfx, dax, _, _ = _index_valued_luts(tmp_path, solver)
    monkeypatch.setenv("LUT_BASENAME", base)
    monkeypatch.setenv("TSX_LUT_DIRECT_DIMS", dims)
    f2c = C.CDLL(os.path.join(ROOT, "tenstream_amd", "lib", "libtsx_f2c.so"))
    opp, ierr = C.c_void_p(), C.c_int(-1)
    f2c.pprts_f2c_opp_init(0, sid, C.byref(opp), C.byref(ierr))
    assert ierr.value == 0 and opp.value
    f2c.pprts_f2c_opp_get_coeff.argtypes = [C.c_void_p] + [C.c_float] * 6 + [C.c_int] * 4 + [C.c_void_p, C.POINTER(C.c_int)]
    fam = {1: "dir2dir8_coeff_symmetry" if S == 8 else "dir2dir_coeff_symmetry_none",
           2: "dir8_to_diff16_coeff_symmetry" if S == 8 else "dir3_to_diff10_coeff_symmetry"}
    pts = [(1, 2, 1, 1, 1, 1), (3, 1, 2, 0, 0, 2), (2, 3, 4, 2, 2, 4)]   # lattice nodes of the (thinned) direct axes
    for p in pts:
        tauz, w0, asp, g, phi, theta = (np.float32(dax[d][i]) for d, i in enumerate(p))
        for imode in (1, 2):
            n = S * S if imode == 1 else S * D
            for east in (0, 1):
                for north in (0, 1):
                    out = np.full(n, -1, dtype=np.float32)
                    f2c.pprts_f2c_opp_get_coeff(opp, tauz, w0, g, asp, phi, theta, imode, east, north, n, out.ctypes.data,
                                                C.byref(ierr))
                    assert ierr.value == 0
                    assert out.astype(int).tolist() == SYM[fam[imode]]["image"][f"e{east}n{north}"], (imode, east, north)
        out = np.full(D * D, -1, dtype=np.float32)
        f2c.pprts_f2c_opp_get_coeff(opp, tauz, w0, g, asp, phi, theta, 3, 1, 1, D * D, out.ctypes.data, C.byref(ierr))
        assert ierr.value == 0 and out.astype(int).tolist() == list(range(D * D))   # diff2diff: no relabelling
    f2c.pprts_f2c_opp_destroy(opp, C.byref(ierr))


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["3_10", "8_16"])
def test_cell_kernels_relabel_like_the_reference_text(gpu, tmp_path, solver):
    """The per-cell lookup kernels (tsx_k_lut_dir, what a solve uses) on index-valued tables: every cell's dir2dir / dir2diff
    vector equals the fixture image for the quadrant the sun stands in (lswitch_east = xinc == 0, lswitch_north = yinc == 0:
    src/pprts.F90:1157-1167, src/pprts_base.F90:1531-1539)."""
    from tenstream_amd.pprts import PprtsSolver

    S, D = (3, 10) if solver == "3_10" else (8, 16)
    _, _, dfx, dax, Tdir, Sdir = _index_valued_luts(tmp_path, solver)
    fam1 = "dir2dir8_coeff_symmetry" if S == 8 else "dir2dir_coeff_symmetry_none"
    fam2 = "dir8_to_diff16_coeff_symmetry" if S == 8 else "dir3_to_diff10_coeff_symmetry"
    Nz, Nx, Ny = 5, 6, 4
    sc = 2.0 ** -12   # index-valued and still a contraction: the direct tables enter a (tiny) solve before they can be read back
    for phi in (20.0, 110.0, 200.0, 290.0):
        sun = O.suninfo(phi, 40.0)
        east, north = int(sun.xinc == 0), int(sun.yinc == 0)
        ps = PprtsSolver(Nz, Nx, Ny, 100.0, 100.0, phi, 40.0, solver=solver)
        ps.set_lut_diffuse(lut.synthetic_diffuse_table(solver), dfx)
        ps.set_lut_direct(Tdir * np.float32(sc), Sdir * np.float32(sc), dax)
        ps.set_optical_properties(0.1, 1e-4, 2e-3, 0.3, 50.0, ldelta_scaling=False)
        ps.solve(1000.0)   # the direct coefficients are looked up by the first solar solve
        d2d = ps.get_field("dir2dir") / sc
        d2f = ps.get_field("dir2diff") / sc
        w1 = np.array(SYM[fam1]["image"][f"e{east}n{north}"], dtype=np.float64)
        w2 = np.array(SYM[fam2]["image"][f"e{east}n{north}"], dtype=np.float64)
        assert np.abs(d2d - w1).max() <= 64 * 2e-6, (phi, east, north)      # interpolation of a constant: sum of weights
        assert np.abs(d2f - w2).max() <= 128 * 2e-6, (phi, east, north)
        assert np.array_equal(np.rint(d2f), np.broadcast_to(w2, d2f.shape))
        assert np.array_equal(np.rint(d2d), np.broadcast_to(w1, d2d.shape))
        ps.close()
