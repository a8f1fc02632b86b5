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


# ------------------------------------------------------------------------------------------------ 8_16 stream geometry
# Known answers of the reference's Monte-Carlo box model for t_boxmc_8_16 (tests/test_boxmc_8_16/test_boxmc_8_16.F90:58-134,
# committed as data in tests/golden/boxmc_8_16_streams.json): which of the 4 top direct sub-streams feeds which, for the sun in
# the north / west / south / east, and which diffuse streams a downward top stream of each azimuth sector feeds.
BMC = _golden("boxmc_8_16_streams.json")


def _sun_switches(phi):
    sun = O.suninfo(float(phi), 60.0)
    return int(sun.xinc == 0), int(sun.yinc == 0), round(float(sun.symmetry_phi), 6)   # lswitch_east, lswitch_north (src/pprts_base.F90:1531-1539)


def _close(a, b):
    t = BMC["tolerance"]
    return abs(a - b) <= t["atol"] + t["rtol"] * max(abs(a), abs(b))


def _native_tables_from_boxmc():
    """The LUT holds one orientation per (symmetric) azimuth; a solve relabels its vectors for the sun's quadrant
    (dir2dir8_coeff_symmetry / dir8_to_diff16_coeff_symmetry, src/optprop.F90:1186-1302).  Undo that for every known-answer
    vector: native[image[q]] = relabelled[q].  Two vectors that address the same native entry must agree within the Monte-Carlo
    tolerance of the reference's own test -- that is what ties the relabelling tables to the reference's stream geometry."""
    S = 8
    native = {}   # (symmetry phi, family) -> {native flat index: value}
    for c in BMC["direct_src_top_face"]["cases"]:
        east, north, symphi = _sun_switches(c["phi"])
        src = c["src"] - 1
        for fam, vec in (("dir2dir8_coeff_symmetry", c["T_target"]), ("dir8_to_diff16_coeff_symmetry", c["S_target"])):
            img = SYM[fam]["image"][f"e{east}n{north}"]
            slot = native.setdefault((symphi, fam), {})
            for dst, v in enumerate(vec):
                q = img[dst * S + src]
                if q in slot:
                    assert _close(slot[q], v), (c["phi"], fam, dst, slot[q], v)
                else:
                    slot[q] = v
    return native


def test_relabelling_is_consistent_with_the_boxmc_8_16_known_answers():
    native = _native_tables_from_boxmc()
    assert sorted(k[0] for k in native) == [0.0, 0.0, 90.0, 90.0]
    # the sun in the west (phi 270, no switch) and in the east (phi 90, both switches) read the SAME native dir2dir entries:
    # (dst 0 <- src 1) and (dst 1 <- src 1), 0.063 and 0.301 in both vectors
    t90 = {q: v for q, v in native[(90.0, "dir2dir8_coeff_symmetry")].items() if v > 0}
    assert sorted(t90) == [0 * 8 + 1, 1 * 8 + 1] and _close(t90[1], 0.0630) and _close(t90[9], 0.3013)
    # the sun in the north (phi 0) and in the south (phi 180) read the same native dir2diff entry (dst 5 <- src 2)
    s0 = {q: v for q, v in native[(0.0, "dir8_to_diff16_coeff_symmetry")].items() if v > 0}
    assert list(s0) == [5 * 8 + 2] and _close(s0[42], 0.625)
    # the oracle's relabelling reproduces every known-answer vector from those native tables
    for c in BMC["direct_src_top_face"]["cases"]:
        east, north, symphi = _sun_switches(c["phi"])
        for fam, is_t, n, vec in (("dir2dir8_coeff_symmetry", 1, 64, c["T_target"]), ("dir8_to_diff16_coeff_symmetry", 0, 128, c["S_target"])):
            tab = np.zeros(n, dtype=np.float32)
            for q, v in native[(symphi, fam)].items():
                tab[q] = v
            got = O.dir_coeff_symmetry(tab, is_t, 8, 16, east, north).reshape(-1, 8)[:, c["src"] - 1]
            assert all(_close(float(g), w) for g, w in zip(got, vec)), (c["phi"], fam, got, vec)


def test_side_stream_orientation_matches_the_boxmc_8_16_known_answers():
    """A downward top stream of the north / west / south / east sector loses a few per cent to the side stream that moves the
    same way: dof 14 (+y), 9 (-x), 13 (-y), 10 (+x), 1-based -- all in the first (downward-tilted) half of their side group, and
    `is_inward` of t_solver_8_16 (src/pprts.F90:416-419, fixture solver_tables.json) says which of a pair moves towards +x / +y."""
    tab = _golden("solver_tables.json")["solvers"]["t_solver_8_16"]
    top_in, side_in = tab["difftop"]["is_inward"], tab["diffside"]["is_inward"]
    want = {"north": (14, True), "west": (9, False), "south": (13, False), "east": (10, True)}
    for c in BMC["diffuse_src_top_face"]["cases"]:
        S = np.array(c["S_target"])
        src = c["src"]
        assert top_in[src - 1] and np.argmax(S) == src - 1 and S[src - 1] > 0.9            # a downward stream stays itself
        side = S[8:]
        dom = 9 + int(np.argmax(side))
        dof, inward = want[c["quadrant"]]
        assert dom == dof and side_in[(dof - 9) % 4] == inward, (c["quadrant"], dom)
        # nothing reaches the upward-tilted halves (dofs 11, 12, 15, 16) or the upward top streams beyond Monte-Carlo noise
        assert S[[10, 11, 14, 15]].max() < 1e-3 and S[[0, 2, 4, 6]].max() < 1e-3
        assert abs(S.sum() - 1.0) < 5e-3 + 1e-3    # bg = (1e-6, 1e-3, .99): nearly nothing is absorbed in a 5 m layer


@pytest.mark.gpu
def test_device_relabelling_reproduces_the_boxmc_8_16_known_answers(gpu):
    """The device's coefficient probe (tsx_opp_get_coeff: table lookup + quadrant relabelling, what pprts_f2c_opp_get_coeff serves)
    on direct tables that hold the native vectors derived from the known answers: for the sun in the north, west, south and
    east it returns the reference's T_target / S_target for the source stream of that case."""
    from tenstream_amd import DiffuseSolver, _lib
    from tenstream_amd.pprts import PprtsSolver

    native = _native_tables_from_boxmc()
    ax2 = lambda lo, hi: np.array([lo, hi], dtype=np.float32)
    axes = [ax2(1e-10, 100.0), ax2(0.0, 0.99999), ax2(0.02, 7.451), ax2(0.0, 0.85), ax2(0.0, 90.0), ax2(0.0, 60.0)]
    n_inner = 16
    Tdir = np.zeros((64, 64), dtype=np.float32)    # (entries, S*S); entry = inner + 16 * (i_phi + 2 * i_theta)
    Sdir = np.zeros((64, 128), dtype=np.float32)
    for (symphi, fam), vals in native.items():
        ip = 0 if symphi == 0.0 else 1
        rows = slice(n_inner * (ip + 2 * 1), n_inner * (ip + 2 * 1) + n_inner)   # theta = 60: the upper node
        for q, v in vals.items():
            (Tdir if fam.startswith("dir2dir") else Sdir)[rows, q] = v
    P = PprtsSolver(4, 4, 4, 100.0, 100.0, 0.0, 60.0, solver="8_16")
    P.set_lut_direct(Tdir, Sdir, axes)
    for c in BMC["direct_src_top_face"]["cases"]:
        east, north, symphi = _sun_switches(c["phi"])
        for imode, n, vec in ((1, 64, c["T_target"]), (2, 128, c["S_target"])):
            out = np.full(n, -1, dtype=np.float32)
            _lib.check(P.lib.tsx_opp_get_coeff(P.h, C.c_float(1.0), C.c_float(0.5), C.c_float(0.3), C.c_float(0.05), C.c_float(symphi),
                                               C.c_float(60.0), imode, east, north, n, C.c_void_p(out.ctypes.data)))
            got = out.reshape(-1, 8)[:, c["src"] - 1]
            assert all(_close(float(g), w) for g, w in zip(got, vec)), (c["phi"], imode, got, vec)
    P.close()
