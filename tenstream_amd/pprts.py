"""Host-side mirror of the reference's pprts call sequence for one solver instance on one GPU:

    init_pprts -> set_angles -> set_optical_properties -> solve_pprts -> pprts_get_result
    (src/pprts.F90:213, 1100, 1764, 2487, 5799; C-ABI c_wrapper/f2c_pprts.h:48-52)

Everything between "optical properties in" and "edn/eup/abso/edir out" runs on the device through the C-ABI
(tsx_pprts_*).  Host-side pieces restated here are the cheap per-call preparations the reference does in
set_optical_properties: delta scaling (src/pprts.F90:1903-1917), which layers are 1-D (:669-677) and their
Eddington coefficients (:1962-1992, src/eddington.F90:173-241).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .solver import DiffuseSolver, KspInfo, _ptr
from .synthetic import delta_scale

TWOSTR_RATIO = 2.0  # -twostr_ratio default, src/tenstream_options.F90:64-158


def eddington_coeff_ec(dtau, w0, g, mu0):
    """eddington_coeff_ec (src/eddington.F90:173-241), vectorised; returns a11, a12, a13, a23, a33."""
    dtau, w0, g = (np.asarray(a, dtype=np.float64) for a in (dtau, w0, g))
    f = 0.75 * g
    g1 = 2.0 - w0 * (1.25 + f)
    g2 = w0 * (0.75 - f)
    g3 = 0.5 - mu0 * f
    slant = np.maximum(dtau / max(np.sqrt(np.finfo(np.float64).tiny), mu0), 0.0)
    g4 = 1.0 - g3
    alpha1 = g1 * g4 + g2 * g3
    alpha2 = g1 * g3 + g2 * g4
    A = np.sqrt(np.maximum((g1 - g2) * (g1 + g2), 1e-12))
    k_mu0 = A * mu0
    k_mu0 = np.where(np.abs(k_mu0 - 1.0) <= 10 * np.finfo(np.float64).eps, 1 - 10 * np.finfo(np.float64).eps, k_mu0)
    k_g3, k_g4 = A * g3, A * g4
    e0 = np.exp(-slant)
    e = np.exp(-A * dtau)
    e2 = e * e
    k_2_e = 2 * A * e
    beta = 1 / (A + g1 + (A - g1) * e2)
    r = g2 * (1 - e2) * beta
    t = k_2_e * beta
    beta2 = w0 * beta / (1 - k_mu0 * k_mu0)
    sdir = beta2 * (k_2_e * (g4 + alpha1 * mu0) - e0 * ((1 + k_mu0) * (alpha1 + k_g4) - (1 - k_mu0) * (alpha1 - k_g4) * e2))
    rdir = beta2 * ((1 - k_mu0) * (alpha2 + k_g3) - (1 + k_mu0) * (alpha2 - k_g3) * e2 - k_2_e * (g3 - alpha2 * mu0) * e0)
    thin = slant <= 1e-6
    t = np.where(thin, 1.0 - g1 * dtau, t)
    r = np.where(thin, g2 * dtau, r)
    sdir = np.where(thin, (1.0 - g3) * (w0 * dtau), sdir)
    rdir = np.where(thin, g3 * (w0 * dtau), rdir)
    tdir = np.where(thin, 1.0 - slant, e0)
    return t, r, rdir, sdir, tdir


class PprtsSolver:
    """One pprts solver (3_10) on one GPU, driven like the reference's Fortran/C API."""

    def __init__(self, Nz, Nx, Ny, dx, dy, phi0, theta0, solver="3_10", device=-1, **decomposition):
        """Nx, Ny: the columns this rank owns; decomposition: xs, ys, glob_xm, glob_ym, rank, nranks, neighbors (W, E, S, N)
        as DiffuseSolver takes them (several ranks: call core.comm_init / core.comm_set_callbacks before the first
        set_optical_properties), force_halo for tests."""
        self.Nz, self.Nx, self.Ny, self.dx, self.dy = int(Nz), int(Nx), int(Ny), float(dx), float(dy)
        self.core = DiffuseSolver(solver, Nz, Nx, Ny, device=device, **decomposition)
        self.lib = self.core.lib
        self.h = self.core.h
        self.phi0, self.theta0 = float(phi0), float(theta0)
        _lib.check(self.lib.tsx_pprts_set_angles(self.h, self.phi0, self.theta0))
        self.mu0 = max(np.cos(np.deg2rad(theta0)), 0.0) if theta0 < 90 else 0.0

    # -- look-up tables ------------------------------------------------------------------------------
    def set_lut_diffuse(self, table, axes):
        self.core.set_lut_diffuse(table, axes)

    def set_lut_direct(self, Tdir, Sdir, axes):
        Tdir = np.ascontiguousarray(Tdir, dtype=np.float32)
        Sdir = np.ascontiguousarray(Sdir, dtype=np.float32)
        n = (C.c_int32 * len(axes))(*[len(a) for a in axes])
        ax = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float32) for a in axes]))
        _lib.check(self.lib.tsx_lut_set_direct(self.h, C.c_void_p(Tdir.ctypes.data), C.c_void_p(Sdir.ctypes.data),
                                               int(Tdir.shape[0]), len(axes), n, C.c_void_p(ax.ctypes.data), 0))

    # -- set_optical_properties ------------------------------------------------------------------------
    def set_optical_properties(self, albedo, kabs, ksca, g, dz, planck=None, ldelta_scaling=True, planck_srfc=None):
        """Fields (Ny, Nx, Nz) float64 (numpy, or CUDA tensors to stay on the device), k = 0 at TOA; albedo scalar or
        (Ny, Nx); planck (Ny, Nx, Nz+1) or None; planck_srfc scalar or (Ny, Nx) or None: the surface's own Planck emission
        (atm%Bsrfc, src/pprts.F90:1773, 1823-1829, used at :4958-4970).  Delta scaling, 1-D layer detection, Eddington coefficients and the
        coefficient lookups all run on the device (tsx_pprts_set_optical_properties)."""
        from .solver import _is_torch

        on_dev = _is_torch(kabs)
        shape = (self.Ny, self.Nx, self.Nz)
        if on_dev:
            import torch

            f = lambda a, shp: (a if _is_torch(a) else torch.as_tensor(np.asarray(a, dtype=np.float64), device=kabs.device)
                                ).to(torch.float64).expand(shp).contiguous()
        else:
            f = lambda a, shp: np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), shp))
        raw = dict(kabs=f(kabs, shape), ksca=f(ksca, shape), g=f(g, shape), dz=f(dz, shape),
                   albedo=f(albedo, (self.Ny, self.Nx)),
                   planck=None if planck is None else f(planck, (self.Ny, self.Nx, self.Nz + 1)),
                   planck_srfc=None if planck_srfc is None else f(planck_srfc, (self.Ny, self.Nx)))
        self._raw, self._ldelta, self._fields = raw, bool(ldelta_scaling), None
        ptr = lambda a: None if a is None else _ptr(a, np.float64)[0]
        _lib.check(self.lib.tsx_pprts_set_optical_properties(
            self.h, ptr(raw["albedo"]), ptr(raw["kabs"]), ptr(raw["ksca"]), ptr(raw["g"]), ptr(raw["dz"]),
            ptr(raw["planck"]), ptr(raw["planck_srfc"]), self.dx, self.dy, int(self._ldelta), 1 if on_dev else 0))

    @property
    def fields(self):
        """Host mirror of what the device derived (delta-scaled properties, Eddington coefficients): for tests and
        diagnostics only, computed on demand from the raw inputs."""
        if self._fields is None:
            r = {k: (None if v is None else (v.cpu().numpy() if hasattr(v, "cpu") else v)) for k, v in self._raw.items()}
            kabs, ksca, g = (np.array(r[k], dtype=np.float64, copy=True) for k in ("kabs", "ksca", "g"))
            if self._ldelta:
                kabs, ksca, g = delta_scale(kabs, ksca, g)
            ext = np.maximum(np.finfo(np.float64).tiny, kabs + ksca)
            a11, a12, a13, a23, a33 = eddington_coeff_ec(r["dz"] * ext, ksca / ext, g, self.mu0)
            self._fields = dict(kabs=kabs, ksca=ksca, g=g, dz=r["dz"], a11=a11, a12=a12, a13=a13, a23=a23, a33=a33,
                                albedo=r["albedo"], planck=r["planck"], planck_srfc=r["planck_srfc"])
        return self._fields

    @property
    def l1d(self):
        """1-D layers as the reference flags them (src/pprts.F90:670-677, 708-719); host mirror of the device logic"""
        dz = self._raw["dz"]
        dz = dz.cpu().numpy() if hasattr(dz, "cpu") else dz
        ex = (dz / self.dx > TWOSTR_RATIO).any(axis=(0, 1))
        l1d = np.zeros(self.Nz, dtype=np.uint8)
        l1d[-1] = ex[-1]
        upper = np.nonzero(ex[:-1])[0]
        if upper.size:
            l1d[: upper.max() + 1] = 1
        l1d[: int(l1d.sum())] = 1  # the count of 1-D layers is applied from the top (:708-719)
        return l1d

    # -- solve_pprts -------------------------------------------------------------------------------------
    def solve(self, edirTOA, lsolar=None, zero_guess=False, uid=None, **opts) -> KspInfo:
        """uid: solve_pprts' opt_solution_uid -- the solution slot whose previous content is the initial guess"""
        lsolar = (edirTOA > 0) if lsolar is None else lsolar  # pprts_f2c_solve, c_wrapper/f2c_pprts.F90:340-341
        if uid is not None:
            _lib.check(self.lib.tsx_pprts_select_solution(self.h, int(uid)))
        if zero_guess:
            _lib.check(self.lib.tsx_pprts_zero_guess(self.h))
        o = None
        if opts:
            o = _lib.KspOpts()
            self.lib.tsx_default_ksp_opts(C.byref(o))
            drt, dat, dmx = self.core.default_tolerances()
            o.rtol, o.atol, o.maxit, o.pc, o.pc_sweeps = drt, dat, dmx, 3, 0
            for k, v in opts.items():
                setattr(o, k, v)
        r = _lib.KspResult()
        _lib.check(self.lib.tsx_pprts_solve(self.h, float(edirTOA), int(bool(lsolar)), None if o is None else C.byref(o),
                                            C.byref(r)))
        return KspInfo(r.reason, r.niter, r.rnorm0, r.rnorm, np.array(r.res_hist[: r.nhist]), r.solve_ms, 0.0, 0.0)

    # -- pprts_get_result -----------------------------------------------------------------------------------
    def get_result(self, out=None):
        """edn, eup, edir (Ny, Nx, Nz+1) [W/m2] and abso (Ny, Nx, Nz) [W/m3]; `out` = (edn, eup, abso, edir) CUDA tensors
        keeps the result on the device."""
        L = self.Nz + 1
        if out is None:
            edn = np.empty((self.Ny, self.Nx, L))
            eup = np.empty_like(edn)
            edir = np.empty_like(edn)
            abso = np.empty((self.Ny, self.Nx, self.Nz))
            where = 0
        else:
            edn, eup, abso, edir = out
            where = 1
        _lib.check(self.lib.tsx_pprts_get_result(self.h, _ptr(edn, np.float64)[0], _ptr(eup, np.float64)[0],
                                                 _ptr(abso, np.float64)[0], _ptr(edir, np.float64)[0], where))
        return edn, eup, abso, edir

    def get_field(self, which):
        S, D = (3, 10) if self.core.D == 10 else (8, 16)
        shapes = {"edir": (0, (self.Ny, self.Nx, self.Nz + 1, S)), "b": (1, self.core.vec_shape),
                  "ediff": (2, self.core.vec_shape), "dir2dir": (3, (self.Ny, self.Nx, self.Nz, S * S)),
                  "dir2diff": (4, (self.Ny, self.Nx, self.Nz, S * D))}
        cell = (self.Ny, self.Nx, self.Nz)   # what the device derived in set_optical_properties (delta scaling, Eddington)
        shapes.update({n: (5 + q, cell) for q, n in enumerate(("kabs", "ksca", "g", "a11", "a12", "a13", "a23", "a33"))})
        idx, shp = shapes[which]
        out = np.empty(shp)
        _lib.check(self.lib.tsx_pprts_get_field(self.h, idx, _ptr(out, np.float64)[0], 0))
        return out

    def close(self):
        self.core.close()
