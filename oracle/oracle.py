"""ctypes front-end of the CPU parity oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under tenstream_amd/ may import this module.

Arrays use the reference's Fortran layouts (see pprts_oracle.h); numpy arrays here are
C-contiguous with *reversed* axis order, e.g. a diffuse vector is ``x[j, i, k, d]`` and a
coefficient field is ``c[j, i, k, dst*D+src]``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ORC_MAXDOF = 32


class Layout(C.Structure):
    _fields_ = [
        ("ntop", C.c_int),
        ("nside", C.c_int),
        ("top_inward", C.c_int * ORC_MAXDOF),
        ("side_inward", C.c_int * ORC_MAXDOF),
        ("Nz", C.c_int),
        ("xm", C.c_int),
        ("ym", C.c_int),
    ]

    @property
    def D(self):
        return self.ntop + 2 * self.nside


class KspTol(C.Structure):
    _fields_ = [("rtol", C.c_double), ("atol", C.c_double), ("dtol", C.c_double), ("maxit", C.c_int)]


class SorOpts(C.Structure):
    _fields_ = [
        ("rtol", C.c_double),
        ("atol", C.c_double),
        ("maxit", C.c_int),
        ("omega", C.c_double),
        ("adaptive_omega", C.c_int),
    ]


class Csr(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("nnz", C.c_int64),
        ("rowptr", C.POINTER(C.c_int64)),
        ("col", C.POINTER(C.c_int32)),
        ("val", C.POINTER(C.c_double)),
    ]


def build(force: bool = False) -> str:
    """Compile oracle/liboracle.so with gcc (building the checker is not using it)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("pprts_oracle.c", "pprts_oracle_phys.c", "pprts_oracle_pipe.c",
                                             "pprts_oracle.h", "pprts_oracle_phys.h")]
    srcs = [s for s in srcs if os.path.exists(s)]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_fbcgs.restype = C.c_int
        _LIB.orc_diff_solve_matfree.restype = C.c_int
        _LIB.orc_diff_solve_ilu.restype = C.c_int
        _LIB.orc_explicit_ediff_1rank.restype = C.c_int
        _LIB.orc_diff_assemble_csr_1rank.restype = C.c_int
    return _LIB


def _p(a, ct=C.c_double):
    return a.ctypes.data_as(C.POINTER(ct))


def layout(solver: str, Nz: int, xm: int, ym: int) -> Layout:
    lay = Layout()
    if solver in ("3_10", 310):
        lib().orc_layout_3_10(C.byref(lay), Nz, xm, ym)
    elif solver in ("8_16", 816):
        lib().orc_layout_8_16(C.byref(lay), Nz, xm, ym)
    else:
        raise ValueError(solver)
    return lay


def _chk(lay: Layout, coeff, l1d, a11, a12, albedo):
    D, Nz, xm, ym = lay.D, lay.Nz, lay.xm, lay.ym
    coeff = np.ascontiguousarray(coeff, dtype=np.float64)
    assert coeff.shape == (ym, xm, Nz, D * D), coeff.shape
    l1d = np.ascontiguousarray(l1d, dtype=np.uint8)
    assert l1d.shape == (Nz,)
    a11 = np.ascontiguousarray(a11, dtype=np.float64)
    a12 = np.ascontiguousarray(a12, dtype=np.float64)
    assert a11.shape == (ym, xm, Nz) and a12.shape == (ym, xm, Nz)
    albedo = np.ascontiguousarray(albedo, dtype=np.float64)
    assert albedo.shape == (ym, xm)
    return coeff, l1d, a11, a12, albedo


def diff_apply(lay: Layout, coeff, l1d, a11, a12, albedo, x):
    """y = (I - T) x, matrix-free, restating pprts_shell.F90:366-541 (one periodic rank)."""
    coeff, l1d, a11, a12, albedo = _chk(lay, coeff, l1d, a11, a12, albedo)
    x = np.ascontiguousarray(x, dtype=np.float64)
    assert x.shape == (lay.ym, lay.xm, lay.Nz + 1, lay.D)
    y = np.empty_like(x)
    lib().orc_diff_apply_1rank(C.byref(lay), _p(coeff), _p(l1d, C.c_uint8), _p(a11), _p(a12), _p(albedo), _p(x), _p(y))
    return y


def op_local(lay: Layout, coeff, l1d, a11, a12, albedo, xx_g):
    """Cell loop only (pprts_shell.F90:413-508) on a ghosted local array; returns ghosted xb."""
    coeff, l1d, a11, a12, albedo = _chk(lay, coeff, l1d, a11, a12, albedo)
    xx_g = np.ascontiguousarray(xx_g, dtype=np.float64)
    assert xx_g.shape == (lay.ym + 2, lay.xm + 2, lay.Nz + 1, lay.D)
    xb = np.zeros_like(xx_g)
    lib().orc_op_mat_mult_ediff_local(C.byref(lay), _p(coeff), _p(l1d, C.c_uint8), _p(a11), _p(a12), _p(albedo),
                                      _p(xx_g), _p(xb))
    return xb


def assemble_csr(lay: Layout, coeff, l1d, a11, a12, albedo):
    """CSR of A = I - T as set_diff_coeff builds it (pprts.F90:5511-5796). Returns scipy CSR."""
    import scipy.sparse as sp

    coeff, l1d, a11, a12, albedo = _chk(lay, coeff, l1d, a11, a12, albedo)
    A = Csr()
    rc = lib().orc_diff_assemble_csr_1rank(C.byref(lay), _p(coeff), _p(l1d, C.c_uint8), _p(a11), _p(a12), _p(albedo),
                                           C.byref(A))
    if rc:
        raise RuntimeError(f"assemble rc={rc}")
    n, nnz = A.n, A.nnz
    rowptr = np.ctypeslib.as_array(A.rowptr, shape=(n + 1,)).copy()
    col = np.ctypeslib.as_array(A.col, shape=(nnz,)).copy()
    val = np.ctypeslib.as_array(A.val, shape=(nnz,)).copy()
    lib().orc_csr_free(C.byref(A))
    return sp.csr_matrix((val, col, rowptr), shape=(n, n))


def default_tolerances(glob_xm, glob_ym, glob_zm, unconstrained_fraction=1.0):
    rtol, atol, maxit = C.c_double(), C.c_double(), C.c_int()
    lib().orc_determine_ksp_tolerances(glob_xm, glob_ym, glob_zm, C.c_double(unconstrained_fraction),
                                       C.byref(rtol), C.byref(atol), C.byref(maxit))
    return rtol.value, atol.value, maxit.value


def solve_matfree(lay, coeff, l1d, a11, a12, albedo, b, x0=None, rtol=1e-5, atol=1e-8, maxit=1000, dtol=1e4):
    coeff, l1d, a11, a12, albedo = _chk(lay, coeff, l1d, a11, a12, albedo)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64, order="C", copy=True)
    tol = KspTol(rtol, atol, dtol, maxit)
    nit = C.c_int()
    hist = np.full(maxit + 2, -1.0)
    reason = lib().orc_diff_solve_matfree(C.byref(lay), _p(coeff), _p(l1d, C.c_uint8), _p(a11), _p(a12), _p(albedo),
                                          _p(b), _p(x), C.byref(tol), C.byref(nit), _p(hist), len(hist))
    return x, dict(reason=reason, niter=nit.value, res_hist=hist[: nit.value + 1])


def solve_ilu(lay, coeff, l1d, a11, a12, albedo, b, x0=None, rtol=1e-5, atol=1e-8, maxit=1000, dtol=1e4):
    """The reference's default 1-rank path: assembled AIJ + KSPFBCGS + ILU(0)."""
    coeff, l1d, a11, a12, albedo = _chk(lay, coeff, l1d, a11, a12, albedo)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64, order="C", copy=True)
    tol = KspTol(rtol, atol, dtol, maxit)
    nit = C.c_int()
    hist = np.full(maxit + 2, -1.0)
    ta, tf, ts = C.c_double(), C.c_double(), C.c_double()
    reason = lib().orc_diff_solve_ilu(C.byref(lay), _p(coeff), _p(l1d, C.c_uint8), _p(a11), _p(a12), _p(albedo),
                                      _p(b), _p(x), C.byref(tol), C.byref(nit), _p(hist), len(hist),
                                      C.byref(ta), C.byref(tf), C.byref(ts))
    return x, dict(reason=reason, niter=nit.value, res_hist=hist[: nit.value + 1], t_assemble=ta.value,
                   t_factor=tf.value, t_solve=ts.value)


def solve_bjacobi_ilu_mt(lay, coeff, l1d, a11, a12, albedo, b, npx, npy, x0=None, rtol=1e-5, atol=1e-8, maxit=1000,
                         dtol=1e4, tighten=None):
    """The reference's default on npx*npy ranks (FBCGS + PCBJACOBI/ILU(0), pprts.F90:4415-4425), one host thread per
    subdomain (pprts_oracle_mt.c).  tighten=(rtol2, atol2, maxit2): a second solve continues from the first solution with
    the same factors; its result is info['x_tight'] (reason_tight, niter_tight, t_solve_tight)."""
    coeff, l1d, a11, a12, albedo = _chk(lay, coeff, l1d, a11, a12, albedo)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64, order="C", copy=True)
    tol = KspTol(rtol, atol, dtol, maxit)
    nit = C.c_int()
    hist = np.full(maxit + 2, -1.0)
    ta, tf, ts = C.c_double(), C.c_double(), C.c_double()
    if tighten is None:
        reason = lib().orc_diff_solve_bjacobi_ilu_mt(C.byref(lay), _p(coeff), _p(l1d, C.c_uint8), _p(a11), _p(a12),
                                                     _p(albedo), _p(b), _p(x), C.byref(tol), int(npx), int(npy),
                                                     C.byref(nit), _p(hist), len(hist), C.byref(ta), C.byref(tf),
                                                     C.byref(ts))
        extra = {}
    else:
        tol2 = KspTol(tighten[0], tighten[1], dtol, tighten[2])
        x2 = np.zeros_like(b)
        nit2, r2, ts2 = C.c_int(), C.c_int(), C.c_double()
        reason = lib().orc_diff_solve_bjacobi_ilu_mt2(C.byref(lay), _p(coeff), _p(l1d, C.c_uint8), _p(a11), _p(a12),
                                                      _p(albedo), _p(b), _p(x), C.byref(tol), int(npx), int(npy),
                                                      C.byref(nit), _p(hist), len(hist), C.byref(ta), C.byref(tf),
                                                      C.byref(ts), C.byref(tol2), _p(x2), C.byref(nit2), C.byref(r2),
                                                      C.byref(ts2))
        extra = dict(x_tight=x2, reason_tight=r2.value, niter_tight=nit2.value, t_solve_tight=ts2.value)
    return x, dict(reason=reason, niter=nit.value, res_hist=hist[: nit.value + 1], t_assemble=ta.value,
                   t_factor=tf.value, t_solve=ts.value, **extra)


def solve_sor(lay, coeff, l1d, a11, a12, albedo, b, x0=None, rtol=1e-5, atol=1e-8, maxit=10000, omega=1.0,
              adaptive=True):
    """The reference's PETSc-free explicit solver (pprts_explicit.F90:461-713)."""
    coeff, l1d, a11, a12, albedo = _chk(lay, coeff, l1d, a11, a12, albedo)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64, order="C", copy=True)
    o = SorOpts(rtol, atol, maxit, omega, int(adaptive))
    nit = C.c_int()
    hist = np.zeros(100)
    rc = lib().orc_explicit_ediff_1rank(C.byref(lay), _p(coeff), _p(l1d, C.c_uint8), _p(a11), _p(a12), _p(albedo),
                                        _p(b), _p(x), C.byref(o), C.byref(nit), _p(hist), len(hist))
    return x, dict(converged=(rc == 0), niter=nit.value, res_hist=hist[: min(nit.value, 100)])


# ---- coefficient / source pieces (pprts_oracle_phys.h) ---------------------------------------------------
ORC_LUT_MAXDIM = 8


class Lut(C.Structure):
    _fields_ = [("ndim", C.c_int), ("n", C.c_int * ORC_LUT_MAXDIM), ("axis", C.POINTER(C.c_float) * ORC_LUT_MAXDIM),
                ("nvec", C.c_int), ("table", C.POINTER(C.c_float))]


def search_sorted_bisection(arr, val, dtype=np.float64):
    """src/search.fypp:177-228: 1-based fractional location of val in the sorted array."""
    a = np.ascontiguousarray(arr, dtype=dtype)
    if dtype == np.float64:
        f = lib().orc_search_sorted_bisection_f64
        f.restype = C.c_double
        return f(_p(a), len(a), C.c_double(val))
    f = lib().orc_search_sorted_bisection_f32
    f.restype = C.c_float
    return f(_p(a, C.c_float), len(a), C.c_float(val))


def interp_vec_nd(pti, db, shape):
    """interp_vec_simplex_nd -> interp_vec_bilinear_iterative (src/interpolation.F90:317-360).
    db: (nentries, nvec) C-order == Fortran (nvec, nentries); shape: extents of the unravelled dims."""
    pti = np.ascontiguousarray(pti, dtype=np.float32)
    db = np.ascontiguousarray(db, dtype=np.float32)
    nvec = db.shape[1]
    shp = (C.c_int * len(shape))(*shape)
    offs = (C.c_int64 * len(shape))()
    lib().orc_ndarray_offsets(shp, len(shape), offs)
    out = np.zeros(nvec, dtype=np.float32)
    lib().orc_interp_vec_nd_f32(_p(pti, C.c_float), len(shape), _p(db, C.c_float), nvec, offs, _p(out, C.c_float))
    return out


def make_lut(axes, table):
    """axes: list of float32 arrays; table: (prod(n), nvec) C-order float32 (== Fortran (nvec, nentries))."""
    lut = Lut()
    lut.ndim = len(axes)
    keep = []
    for d, a in enumerate(axes):
        a = np.ascontiguousarray(a, dtype=np.float32)
        keep.append(a)
        lut.n[d] = len(a)
        lut.axis[d] = _p(a, C.c_float)
    table = np.ascontiguousarray(table, dtype=np.float32)
    keep.append(table)
    lut.nvec = table.shape[1]
    lut.table = _p(table, C.c_float)
    lut._keep = keep
    return lut


def get_coeff_diff2diff(lut, kabs, ksca, g, dz, dx):
    out = np.zeros(lut.nvec, dtype=np.float32)
    lib().orc_get_coeff_diff2diff(C.byref(lut), C.c_double(kabs), C.c_double(ksca), C.c_double(g), C.c_double(dz),
                                  C.c_double(dx), _p(out, C.c_float))
    return out


def alloc_coeff_diff2diff(lut, kabs, ksca, g, dz, dx, l1d):
    """alloc_coeff_diff2diff (src/pprts.F90:3433-3462) for fields shaped (ym, xm, Nz)."""
    kabs, ksca, g, dz = (np.ascontiguousarray(a, dtype=np.float64) for a in (kabs, ksca, g, dz))
    ym, xm, Nz = kabs.shape
    l1d = np.ascontiguousarray(l1d, dtype=np.uint8)
    out = np.zeros((ym, xm, Nz, lut.nvec), dtype=np.float64)
    lib().orc_alloc_coeff_diff2diff(C.byref(lut), Nz, xm, ym, _p(kabs), _p(ksca), _p(g), _p(dz), C.c_double(dx),
                                    _p(l1d, C.c_uint8), _p(out))
    return out


def delta_scale(kabs, ksca, g, f=None):
    a, b, c = C.c_double(kabs), C.c_double(ksca), C.c_double(g)
    lib().orc_delta_scale(C.byref(a), C.byref(b), C.byref(c), 0 if f is None else 1, C.c_double(0.0 if f is None else f))
    return a.value, b.value, c.value


def eddington_coeff_ec(dtau, w0, g, mu0):
    o = [C.c_double() for _ in range(5)]
    lib().orc_eddington_coeff_ec(C.c_double(dtau), C.c_double(w0), C.c_double(g), C.c_double(mu0), *[C.byref(v) for v in o])
    return [v.value for v in o]


def B_eff(B_far, B_near, tau):
    f = lib().orc_B_eff
    f.restype = C.c_double
    return f(C.c_double(B_far), C.c_double(B_near), C.c_double(tau))


# ---- direct beam / source / post-processing (pprts_oracle_pipe.c) --------------------------------------------
class DirLayout(C.Structure):
    _fields_ = [("dtop", C.c_int), ("dside", C.c_int), ("top_div", C.c_int), ("side_div", C.c_int)]


class SunInfo(C.Structure):
    _fields_ = [("phi", C.c_double), ("theta", C.c_double), ("mu", C.c_double), ("costheta", C.c_double),
                ("symmetry_phi", C.c_double), ("xinc", C.c_int), ("yinc", C.c_int)]


def dir_layout_3_10():
    d = DirLayout()
    lib().orc_dir_layout_3_10(C.byref(d))
    return d


def dir_layout(solver="3_10"):
    d = DirLayout()
    (lib().orc_dir_layout_3_10 if solver in ("3_10", 310) else lib().orc_dir_layout_8_16)(C.byref(d))
    return d


def suninfo(phi, theta):
    s = SunInfo()
    lib().orc_setup_suninfo(C.c_double(phi), C.c_double(theta), C.byref(s))
    return s


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def dir_coeff_symmetry(coeff, is_dir2dir, S, D, east, north):
    """src/optprop.F90:1009-1045, 1186-1240, 1268-1302 on one coefficient vector (flat dst*S + src); returns a copy."""
    v = np.array(coeff, dtype=np.float32, order="C")
    lib().orc_dir_coeff_symmetry(int(is_dir2dir), int(S), int(D), int(east), int(north), _p(v, C.c_float))
    return v


def alloc_coeff_dir(lut_, is_dir2dir, kabs, ksca, g, dz, dx, sun, l1d, S=3, D=10):
    kabs, ksca, g, dz = (_c64(a) for a in (kabs, ksca, g, dz))
    ym, xm, Nz = kabs.shape
    l1d = np.ascontiguousarray(l1d, dtype=np.uint8)
    out = np.zeros((ym, xm, Nz, lut_.nvec))
    lib().orc_alloc_coeff_dir(C.byref(lut_), int(is_dir2dir), S, D, Nz, xm, ym, _p(kabs), _p(ksca), _p(g), _p(dz),
                              C.c_double(dx), C.byref(sun), _p(l1d, C.c_uint8), _p(out))
    return out


def explicit_edir(lay, dlay, sun, dir2dir, l1d, a33, edirTOA, dx, dy, rtol=1e-5, atol=1e-8, maxit=1002, edir0=None):
    S = dlay.dtop + 2 * dlay.dside
    dir2dir, a33 = _c64(dir2dir), _c64(a33)
    l1d = np.ascontiguousarray(l1d, dtype=np.uint8)
    edir = np.zeros((lay.ym, lay.xm, lay.Nz + 1, S)) if edir0 is None else np.array(edir0, dtype=np.float64, order="C")
    nit = C.c_int()
    f = lib().orc_explicit_edir_1rank
    f.restype = C.c_int
    rc = f(C.byref(lay), C.byref(dlay), C.byref(sun), _p(dir2dir), _p(l1d, C.c_uint8), _p(a33), C.c_double(edirTOA),
           C.c_double(dx), C.c_double(dy), C.c_double(rtol), C.c_double(atol), maxit, _p(edir), C.byref(nit))
    return edir, dict(converged=rc == 0, niter=nit.value)


def setup_b_solar(lay, dlay, sun, dir2diff, l1d, a13, a23, albedo, edir):
    dir2diff, a13, a23, albedo, edir = (_c64(a) for a in (dir2diff, a13, a23, albedo, edir))
    l1d = np.ascontiguousarray(l1d, dtype=np.uint8)
    b = np.zeros((lay.ym, lay.xm, lay.Nz + 1, lay.D))
    lib().orc_setup_b_solar_1rank(C.byref(lay), C.byref(dlay), C.byref(sun), _p(dir2diff), _p(l1d, C.c_uint8), _p(a13),
                                  _p(a23), _p(albedo), _p(edir), _p(b))
    return b


def setup_b_thermal(lay, diff2diff, l1d, a11, a12, albedo, planck, kabs, dz, dx, dy, planck_srfc=None):
    """planck_srfc (ym, xm) = atm%Bsrfc (set_optical_properties' optional argument, src/pprts.F90:1823-1829) or None"""
    diff2diff, a11, a12, albedo, planck, kabs, dz = (_c64(a) for a in (diff2diff, a11, a12, albedo, planck, kabs, dz))
    srfc = None if planck_srfc is None else _c64(np.broadcast_to(planck_srfc, (lay.ym, lay.xm)))
    l1d = np.ascontiguousarray(l1d, dtype=np.uint8)
    b = np.zeros((lay.ym, lay.xm, lay.Nz + 1, lay.D))
    lib().orc_setup_b_thermal_1rank(C.byref(lay), _p(diff2diff), _p(l1d, C.c_uint8), _p(a11), _p(a12), _p(albedo),
                                    _p(planck), None if srfc is None else _p(srfc), _p(kabs), _p(dz), C.c_double(dx),
                                    C.c_double(dy), _p(b))
    return b


def scale_diff(lay, dz, dx, dy, to_Wm2, ediff):
    e = np.array(ediff, dtype=np.float64, order="C")
    lib().orc_scale_diff(C.byref(lay), _p(_c64(dz)), C.c_double(dx), C.c_double(dy), int(to_Wm2), _p(e))
    return e


def scale_dir(lay, dlay, dz, dx, dy, to_Wm2, edir):
    e = np.array(edir, dtype=np.float64, order="C")
    lib().orc_scale_dir(C.byref(lay), C.byref(dlay), _p(_c64(dz)), C.c_double(dx), C.c_double(dy), int(to_Wm2), _p(e))
    return e


def calc_flx_div(lay, dlay, sun, dir2dir, dir2diff, diff2diff, l1d, a11, a12, kabs, dz, dx, dy, edir, ediff, b_thermal=None):
    l1d = np.ascontiguousarray(l1d, dtype=np.uint8)
    abso = np.zeros((lay.ym, lay.xm, lay.Nz))
    none = C.POINTER(C.c_double)()
    keep = [_c64(a) if a is not None else None for a in (dir2dir, dir2diff, diff2diff, a11, a12, kabs, dz, edir, ediff, b_thermal)]
    ptr = [(_p(a) if a is not None else none) for a in keep]
    lib().orc_calc_flx_div_1rank(C.byref(lay), C.byref(dlay) if dlay is not None else None,
                                 C.byref(sun) if sun is not None else None, ptr[0], ptr[1], ptr[2], _p(l1d, C.c_uint8),
                                 ptr[3], ptr[4], ptr[5], ptr[6], C.c_double(dx), C.c_double(dy), ptr[7], ptr[8], ptr[9],
                                 _p(abso))
    return abso


def get_result(lay, dlay, sun, lsolar, edir, ediff, abso):
    L = lay.Nz + 1
    redir = np.zeros((lay.ym, lay.xm, L))
    redn = np.zeros_like(redir)
    reup = np.zeros_like(redir)
    rabso = np.zeros((lay.ym, lay.xm, lay.Nz))
    none = C.POINTER(C.c_double)()
    ed = _c64(edir) if edir is not None else None
    lib().orc_get_result(C.byref(lay), C.byref(dlay), C.byref(sun), int(lsolar), _p(ed) if ed is not None else none,
                         _p(_c64(ediff)), _p(_c64(abso)), _p(redir), _p(redn), _p(reup), _p(rabso))
    return redn, reup, rabso, redir
