"""Host-side handle of the MI355X diffuse back-end (thin wrapper over the C-ABI, include/tsx.h).

Mirrors the seam in the reference's `pprts()` (src/pprts.F90:2794-2813): coefficients in
(`solver%diff2diff`, `atm%a11/a12/albedo/l1d`), `solver%b` and `solution%ediff` in/out, iteration
count and residual history out.  Arrays are numpy (host) or torch CUDA tensors (device, used in
place) in the reference's layouts with reversed (C-order) axes:
    vectors      x[j, i, k, d]            <->  Fortran (0:D-1, zs:ze, xs:xe, ys:ye)
    coefficients c[j, i, k, dst*D + src]  <->  Fortran (1:D*D, zs:ze-1, xs:xe, ys:ye)
    a11/a12      a[j, i, k];  albedo[j, i];  l1d[k]
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import TSX_DEVICE, TSX_HOST

SOLVER_IDS = {"3_10": 310, "8_16": 816, 310: 310, 816: 816}
STREAMS = {310: (2, 4), 816: (8, 4)}  # (difftop%dof, diffside%dof)  src/pprts.F90:332-349, 413-425


@dataclass
class KspInfo:
    reason: int
    niter: int
    rnorm0: float
    rnorm: float
    res_hist: np.ndarray
    solve_ms: float
    import_ms: float
    export_ms: float


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _kind(a):
    """real kind (bytes per element) of a vector handed to the seam: 8 = real64, 4 = real32 (ireals of a single-precision
    TenStream build)"""
    if _is_torch(a):
        import torch

        return {torch.float64: 8, torch.float32: 4}[a.dtype]
    return {np.dtype(np.float64): 8, np.dtype(np.float32): 4}[a.dtype]


def _ptr(a, dtype):
    """(void*, where) of a numpy array or torch CUDA tensor; checks dtype and contiguity."""
    if a is None:
        return None, None
    if _is_torch(a):
        import torch

        want = {np.float64: torch.float64, np.float32: torch.float32, np.uint8: torch.uint8}[dtype]
        if a.dtype != want or not a.is_contiguous() or not a.is_cuda:
            raise TypeError("device arrays must be contiguous CUDA tensors of the right dtype")
        # the library works on its own stream: whatever torch still has queued on this tensor must be done first
        torch.cuda.current_stream(a.device).synchronize()
        return C.c_void_p(a.data_ptr()), TSX_DEVICE
    if a.dtype != dtype or not a.flags.c_contiguous:
        raise TypeError(f"host arrays must be C-contiguous {dtype}")
    return C.c_void_p(a.ctypes.data), TSX_HOST


_LIVE = None  # solvers still holding a device handle: destroyed at interpreter exit *before* HIP / RCCL unload


def _close_all():
    for s in list(_LIVE or ()):
        try:
            s.close()
        except Exception:
            pass


class DiffuseSolver:
    """One diffuse system (I - T) x = b on one rank/GPU."""

    def __init__(self, solver, Nz, xm, ym, *, xs=0, ys=0, glob_xm=None, glob_ym=None, rank=0, nranks=1,
                 neighbors=None, device=-1, force_halo=False):
        self.lib = _lib.load()
        sid = SOLVER_IDS[solver]
        self.ntop, self.nside = STREAMS[sid]
        self.D = self.ntop + 2 * self.nside
        self.Nz, self.xm, self.ym = int(Nz), int(xm), int(ym)
        w, e, s_, n = neighbors if neighbors is not None else (rank, rank, rank, rank)
        self.rank, self.nranks = int(rank), int(nranks)
        self.grid = _lib.Grid(sid, Nz, xm, ym, xs, ys, glob_xm or xm, glob_ym or ym, rank, nranks, w, e, s_, n,
                              device, int(force_halo))
        h = C.c_void_p()
        _lib.check(self.lib.tsx_create(C.byref(self.grid), C.byref(h)))
        self.h = h
        self._keep = []
        global _LIVE
        if _LIVE is None:
            import atexit
            import weakref

            _LIVE = weakref.WeakSet()
            atexit.register(_close_all)
        _LIVE.add(self)

    # -- shapes ------------------------------------------------------------------------------------
    @property
    def vec_shape(self):
        return (self.ym, self.xm, self.Nz + 1, self.D)

    @property
    def coeff_shape(self):
        return (self.ym, self.xm, self.Nz, self.D * self.D)

    @property
    def n_unknowns(self):
        return self.D * (self.Nz + 1) * self.xm * self.ym

    @property
    def n_cells(self):
        return self.Nz * self.xm * self.ym

    def close(self):
        if getattr(self, "h", None):
            self.lib.tsx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- communicator --------------------------------------------------------------------------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        _lib.check(self.lib.tsx_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, uid: bytes):
        buf = C.create_string_buffer(uid, 128)
        _lib.check(self.lib.tsx_comm_init(self.h, buf))

    def comm_peer_init(self, allgather):
        """Device-resident peer transport (tsx_comm_peer_export / _attach): `allgather(blob: bytes) -> list of every rank's
        blob in rank order` is the host's all-gather (torch.distributed.all_gather_object, MPI_Allgather ...)."""
        buf = C.create_string_buffer(_lib.PEER_BLOB_BYTES)
        mine, err = bytes(_lib.PEER_BLOB_BYTES), None
        try:
            _lib.check(self.lib.tsx_comm_peer_export(self.h, buf))
            mine = buf.raw
        except Exception as e:   # this rank still takes part in the all-gather (an empty blob): the others must not wait for it
            err = e
        blobs = allgather(mine)
        if err is not None:
            raise err
        if len(blobs) != self.nranks or any(len(b) != _lib.PEER_BLOB_BYTES for b in blobs):
            raise ValueError("comm_peer_init: the all-gather must return one blob per rank")
        if any(not any(b) for b in blobs):
            raise RuntimeError("comm_peer_init: a rank could not export its mailbox")
        allb = C.create_string_buffer(b"".join(blobs), _lib.PEER_BLOB_BYTES * len(blobs))
        _lib.check(self.lib.tsx_comm_peer_attach(self.h, allb))

    def comm_peer_selftest(self, rounds=64):
        """collective; number of things this rank found wrong (0.0 = fine): tsx_comm_peer_selftest"""
        bad = C.c_double(-1.0)
        _lib.check(self.lib.tsx_comm_peer_selftest(self.h, int(rounds), C.byref(bad)))
        return bad.value

    def comm_peer_disable(self):
        _lib.check(self.lib.tsx_comm_peer_disable(self.h))

    def comm_peer_set_fences(self, heavy):
        """1: full system-scope fences around every flag of the peer transport (tsx_comm_peer_set_fences); 0: the light ordering"""
        _lib.check(self.lib.tsx_comm_peer_set_fences(self.h, int(heavy)))

    def comm_peer_reset(self):
        """back to the state after attach; barrier over all ranks before and after (tsx_comm_peer_reset)"""
        _lib.check(self.lib.tsx_comm_peer_reset(self.h))

    def comm_set_callbacks(self, exchange, allreduce):
        """Host-staged transport.  exchange(send: list of 4 numpy views W,E,S,N, recv: list of 4 writable views,
        peers: list of 4 ranks) and allreduce(buf: writable numpy view) operate on pinned host memory."""
        def _x(ctx, send, recv, count, peer):
            try:
                n = [int(count[q]) for q in range(4)]
                sv = [np.ctypeslib.as_array(send[q], shape=(n[q],)) if n[q] else np.empty(0) for q in range(4)]
                rv = [np.ctypeslib.as_array(recv[q], shape=(n[q],)) if n[q] else np.empty(0) for q in range(4)]
                exchange(sv, rv, [int(peer[q]) for q in range(4)])
                return 0
            except Exception:  # never unwind through C
                import traceback

                traceback.print_exc()
                return 1

        def _a(ctx, buf, n):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(int(n),)))
                return 0
            except Exception:
                import traceback

                traceback.print_exc()
                return 1

        self._cb = (_lib.EXCHANGE_FN(_x), _lib.ALLREDUCE_FN(_a))
        _lib.check(self.lib.tsx_comm_set_callbacks(self.h, self._cb[0], self._cb[1], None))

    def set_stream(self, stream_ptr):
        _lib.check(self.lib.tsx_set_stream(self.h, C.c_void_p(stream_ptr) if stream_ptr else None))

    # -- operator ------------------------------------------------------------------------------------
    def set_coeffs(self, diff2diff, l1d, a11, a12, albedo):
        """Replaces set_diff_coeff (src/pprts.F90:5511-5796): hand over the per-cell blocks."""
        if tuple(diff2diff.shape) != self.coeff_shape:
            raise ValueError(f"diff2diff shape {tuple(diff2diff.shape)} != {self.coeff_shape}")
        if _is_torch(diff2diff):
            import torch

            kind = 8 if diff2diff.dtype == torch.float64 else 4
        else:
            kind = 8 if diff2diff.dtype == np.float64 else 4
        cp, where = _ptr(diff2diff, np.float64 if kind == 8 else np.float32)
        lp, w2 = _ptr(l1d, np.uint8)
        ap, w3 = _ptr(albedo, np.float64)
        p11, w4 = _ptr(a11, np.float64)
        p12, w5 = _ptr(a12, np.float64)
        ws = {w for w in (where, w2, w3, w4, w5) if w is not None}
        if len(ws) != 1:
            raise TypeError("all arrays of one call must live on the same side (host or device)")
        _lib.check(self.lib.tsx_diff_set_coeffs(self.h, cp, kind, lp, p11, p12, ap, where))

    def set_lut_diffuse(self, table, axes):
        """Upload a diffuse LUT: table (nentries, D*D) float32 (tau fastest), axes = [tau, w0, aspect_zx, g]."""
        t = table if _is_torch(table) else np.ascontiguousarray(table, dtype=np.float32)
        tp, where = _ptr(t, np.float32)
        n = (C.c_int32 * len(axes))(*[len(a) for a in axes])
        ax = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float32) for a in axes]))
        if where == TSX_DEVICE:
            import torch

            ax_dev = torch.from_numpy(ax).to(table.device)
            axp = C.c_void_p(ax_dev.data_ptr())
        else:
            axp = C.c_void_p(ax.ctypes.data)
        _lib.check(self.lib.tsx_lut_set_diffuse(self.h, tp, int(t.shape[1]), int(t.shape[0]), len(axes), n, axp, where))

    def load_lut_diffuse_mmap4(self, path):
        _lib.check(self.lib.tsx_lut_load_diffuse_mmap4(self.h, os.fsencode(path)))

    def set_optprop(self, kabs, ksca, g, dz, dx, l1d, a11, a12, albedo):
        """Replaces alloc_coeff_diff2diff + set_diff_coeff (src/pprts.F90:3396-3490, 5511-5796): coefficient
        planes are interpolated from the LUT on the device.  Fields are (ym, xm, Nz) float64 (delta-scaled)."""
        shp = (self.ym, self.xm, self.Nz)
        for a in (kabs, ksca, g, dz):
            if tuple(a.shape) != shp:
                raise ValueError(f"optical property shape {tuple(a.shape)} != {shp}")
        ptrs, wheres = [], set()
        for a, dt in ((kabs, np.float64), (ksca, np.float64), (g, np.float64), (dz, np.float64), (l1d, np.uint8),
                      (a11, np.float64), (a12, np.float64), (albedo, np.float64)):
            p, w = _ptr(a, dt)
            ptrs.append(p)
            if w is not None:
                wheres.add(w)
        if len(wheres) != 1:
            raise TypeError("all arrays of one call must live on the same side (host or device)")
        _lib.check(self.lib.tsx_diff_set_optprop(self.h, ptrs[0], ptrs[1], ptrs[2], ptrs[3], float(dx), ptrs[4], ptrs[5],
                                                 ptrs[6], ptrs[7], wheres.pop()))

    def get_coeffs(self, out=None):
        """The coefficient blocks in the reference layout, float64 (what solver%diff2diff holds)."""
        if out is None:
            out = np.empty(self.coeff_shape, dtype=np.float64)
        p, where = _ptr(out, np.float64)
        _lib.check(self.lib.tsx_diff_get_coeffs(self.h, p, where))
        return out

    def apply(self, x, out=None):
        """y = (I - T) x  (op_mat_mult_ediff, src/pprts_shell.F90:366-541)."""
        if tuple(x.shape) != self.vec_shape:
            raise ValueError(f"x shape {tuple(x.shape)} != {self.vec_shape}")
        if out is None:
            if _is_torch(x):
                import torch

                out = torch.empty_like(x)
            else:
                out = np.empty_like(x)
        kind = _kind(x)   # real32 vectors (a single-precision ireals build) go through tsx_diff_apply_r as they are
        dt = np.float64 if kind == 8 else np.float32
        xp, where = _ptr(x, dt)
        yp, w2 = _ptr(out, dt)
        if where != w2:
            raise TypeError("x and out must live on the same side")
        if kind == 8:
            _lib.check(self.lib.tsx_diff_apply(self.h, xp, yp, where))
        else:
            _lib.check(self.lib.tsx_diff_apply_r(self.h, xp, yp, 4, where))
        return out

    def pc_apply(self, v, pc=1, sweeps=1, out=None, mixed=False):
        """z = M^-1 v with the solver's preconditioner (test hook).  mixed: the path the solver uses by default (fp32
        directions computed from the packed fp16 blocks) instead of the exact fp64 one."""
        if out is None:
            if _is_torch(v):
                import torch

                out = torch.empty_like(v)
            else:
                out = np.empty_like(v)
        vp_, where = _ptr(v, np.float64)
        zp, w2 = _ptr(out, np.float64)
        if where != w2:
            raise TypeError("v and out must live on the same side")
        _lib.check(self.lib.tsx_diff_pc_apply(self.h, vp_, zp, where, pc, sweeps, int(bool(mixed))))
        return out

    def default_tolerances(self, unconstrained_fraction=1.0):
        rtol, atol, maxit = C.c_double(), C.c_double(), C.c_int32()
        _lib.check(self.lib.tsx_determine_ksp_tolerances(self.h, unconstrained_fraction, C.byref(rtol), C.byref(atol),
                                                         C.byref(maxit)))
        return rtol.value, atol.value, maxit.value

    def solve(self, b, x, *, rtol=None, atol=None, maxit=None, dtol=None, pc=None, pc_sweeps=None,
              check_every=None, fp32_directions=None, pc_coeff_fp16=None, explicit_solver=None,
              accept_incomplete_solve=None, initial_guess_zero=None) -> KspInfo:
        """Solve in place: x holds the initial guess on entry (src/pprts.F90:4343) and the solution on exit.
        explicit_solver=1: explicit_ediff's stationary iteration (-<prefix>explicit, src/pprts.F90:2799) instead of FBCGS.
        accept_incomplete_solve=1: -accept_incomplete_solve (src/pprts.F90:4271-4273): no retry from zero after a failed solve.
        initial_guess_zero=1: x is not read (KSPSetInitialGuessNonzero(FALSE)); its content on entry is ignored."""
        if tuple(b.shape) != self.vec_shape or tuple(x.shape) != self.vec_shape:
            raise ValueError("b/x shape mismatch")
        o = _lib.KspOpts()
        self.lib.tsx_default_ksp_opts(C.byref(o))
        drt, dat, dmx = self.default_tolerances()
        o.rtol, o.atol, o.maxit = drt, dat, dmx
        for name, val in (("rtol", rtol), ("atol", atol), ("maxit", maxit), ("dtol", dtol), ("pc", pc),
                          ("pc_sweeps", pc_sweeps), ("check_every", check_every), ("fp32_directions", fp32_directions),
                          ("pc_coeff_fp16", pc_coeff_fp16), ("explicit_solver", explicit_solver),
                          ("accept_incomplete_solve", accept_incomplete_solve), ("initial_guess_zero", initial_guess_zero)):
            if val is not None:
                setattr(o, name, val)
        kind = _kind(b)
        if _kind(x) != kind:
            raise TypeError("b and x must have the same real kind")
        dt = np.float64 if kind == 8 else np.float32
        bp, where = _ptr(b, dt)
        xp, w2 = _ptr(x, dt)
        if where != w2:
            raise TypeError("b and x must live on the same side")
        r = _lib.KspResult()
        if kind == 8:
            _lib.check(self.lib.tsx_diff_solve(self.h, bp, xp, where, C.byref(o), C.byref(r)))
        else:   # ireals = real32: the vectors cross the seam as they are (tsx_diff_solve_r)
            _lib.check(self.lib.tsx_diff_solve_r(self.h, bp, xp, 4, where, C.byref(o), C.byref(r)))
        return KspInfo(r.reason, r.niter, r.rnorm0, r.rnorm, np.array(r.res_hist[: r.nhist]), r.solve_ms,
                       r.import_ms, r.export_ms)

    # -- the direct seam and setup_b on their own (src/pprts.F90:2698-2755, 4641-4987) ------------------------
    @property
    def S(self):
        return 3 if self.D == 10 else 8   # dirtop%dof + 2 dirside%dof, src/pprts.F90:332-349, 413-425

    def set_angles(self, phi0, theta0):
        _lib.check(self.lib.tsx_pprts_set_angles(self.h, float(phi0), float(theta0)))

    def dir_set_coeffs(self, dir2dir, dir2diff, l1d, dx, dy, a33=None, a13=None, a23=None):
        """set_dir_coeff's input (src/pprts.F90:4493-4630): solver%dir2dir (ym, xm, Nz, S*S) and solver%dir2diff
        (ym, xm, Nz, S*D) or None, float64 or float32, flat index dst*S + src; l1d (Nz); a33 / a13 / a23 (ym, xm, Nz) float64
        where layers are 1-D.  set_angles comes first."""
        kind = _kind(dir2dir)
        dt = np.float64 if kind == 8 else np.float32
        tp, where = _ptr(dir2dir, dt)
        sp = _ptr(dir2diff, dt)[0] if dir2diff is not None else None
        l1d = np.ascontiguousarray(l1d, dtype=np.uint8) if not _is_torch(l1d) else l1d
        if _is_torch(l1d) != (where == TSX_DEVICE):
            raise TypeError("all arrays of one call must live on the same side (host or device)")
        p = lambda a: None if a is None else _ptr(a, np.float64)[0]
        _lib.check(self.lib.tsx_dir_set_coeffs(self.h, tp, sp, kind, _ptr(l1d, np.uint8)[0], p(a33), p(a13), p(a23), float(dx),
                                               float(dy), where))

    def dir_solve(self, edirTOA, edir, rtol=0.0, atol=0.0, maxit=0):
        """explicit_edir (src/pprts_explicit.F90:60-459) in place on edir (ym, xm, Nz+1, S) [W], float64 or float32;
        returns (niter, residual, converged)"""
        kind = _kind(edir)
        ep, where = _ptr(edir, np.float64 if kind == 8 else np.float32)
        it, cv, res = C.c_int32(), C.c_int32(), C.c_double()
        _lib.check(self.lib.tsx_dir_solve(self.h, float(edirTOA), ep, kind, where, float(rtol), float(atol), int(maxit),
                                          C.byref(it), C.byref(res), C.byref(cv)))
        return it.value, res.value, bool(cv.value)

    def setup_b_solar(self, b, edir=None, albedo=None):
        """set_solar_source (src/pprts.F90:4684-4846) into b (vec_shape); edir None = the beam dir_solve left on the device"""
        kind = _kind(b)
        dt = np.float64 if kind == 8 else np.float32
        bp, where = _ptr(b, dt)
        ep = None if edir is None else _ptr(edir, dt)[0]
        ap = None if albedo is None else _ptr(albedo, np.float64)[0]
        _lib.check(self.lib.tsx_setup_b_solar(self.h, ep, ap, bp, kind, where))
        return b

    def setup_b_thermal(self, b, planck, kabs, dz, dx, dy, planck_srfc=None):
        """set_thermal_source (src/pprts.F90:4848-4987) into b; planck (ym, xm, Nz+1), planck_srfc (ym, xm) or None, kabs, dz
        (ym, xm, Nz) float64; the diffuse coefficients (set_coeffs / set_optprop) come first"""
        kind = _kind(b)
        bp, where = _ptr(b, np.float64 if kind == 8 else np.float32)
        p = lambda a: None if a is None else _ptr(a, np.float64)[0]
        _lib.check(self.lib.tsx_setup_b_thermal(self.h, p(planck), p(planck_srfc), p(kabs), p(dz), float(dx), float(dy), bp, kind,
                                                where))
        return b

    # -- measurement ---------------------------------------------------------------------------------
    def dedup_info(self):
        """(in use?, number of distinct transport blocks) of the current coefficients"""
        on, n = C.c_int32(), C.c_int64()
        _lib.check(self.lib.tsx_dedup_info(self.h, C.byref(on), C.byref(n)))
        self.dedup_mode = int(on.value)   # bit 0: bit-identical blocks shared; bit 1: near-identical blocks grouped for M^-1;
        #                                   bit 2: grouping taken over from the previous coefficient set (validated)
        return bool(on.value & 1), int(n.value)

    def pc_info(self):
        """(TSX_PC_* id, pc_sweeps, scan kernels?, identical recurrence records shared?) of the preconditioner the last
        solve actually ran"""
        pc, sw, scan = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self.lib.tsx_pc_info(self.h, C.byref(pc), C.byref(sw), C.byref(scan)))
        return int(pc.value), int(sw.value), bool(scan.value & 1), bool(scan.value & 2)

    def flow_info(self):
        """how the last application of M^-1 ran its intermediate passes (tsx_flow_info)"""
        o = (C.c_int32 * 8)()
        _lib.check(self.lib.tsx_flow_info(self.h, o))
        return {"in_use": bool(o[0]), "first_pass": int(o[1]), "end_pass": int(o[2]), "columns_per_tile": int(o[3]),
                "fat": bool(o[4]), "granules": bool(o[5]), "tiles_per_pass": int(o[6]), "workgroups": int(o[7])}

    def log_enable(self, on=True):
        """the reference's log events for this path (+ roctx ranges): tsx_log_enable"""
        _lib.check(self.lib.tsx_log_enable(self.h, int(bool(on))))

    def log_get(self):
        """{event name: (count, device milliseconds)} (tsx_log_get; synchronises the stream)"""
        n = C.c_int32(0)
        names = (C.c_char_p * 16)()
        counts = (C.c_int64 * 16)()
        ms = (C.c_double * 16)()
        _lib.check(self.lib.tsx_log_get(self.h, C.byref(n), names, counts, ms))
        return {names[q].decode(): (int(counts[q]), float(ms[q])) for q in range(n.value)}

    def bench_kernel(self, kernel: int, reps: int) -> float:
        ms = C.c_float()
        _lib.check(self.lib.tsx_bench_kernel(self.h, kernel, reps, C.byref(ms)))
        return ms.value

    def algorithmic_bytes(self, kernel: int) -> float:
        b = C.c_double()
        _lib.check(self.lib.tsx_algorithmic_bytes(self.h, kernel, C.byref(b)))
        return b.value

    def probe_bandwidth(self, nbytes=1 << 30, reps=5):
        """{copy_GBps, read_GBps, copy_variant, read_variant}: the best of the streaming probes (tsx_probe_bandwidth)"""
        o = (C.c_double * 4)()
        _lib.check(self.lib.tsx_probe_bandwidth(self.h, nbytes, reps, o))
        return {"copy_GBps": o[0], "read_GBps": o[1], "copy_variant": int(o[2]), "read_variant": int(o[3])}

    def probe_copy_bandwidth(self, nbytes=1 << 30, reps=10) -> float:
        g = C.c_double()
        _lib.check(self.lib.tsx_probe_copy_bandwidth(self.h, nbytes, reps, C.byref(g)))
        return g.value
