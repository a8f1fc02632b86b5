"""Look-up-table plumbing on the host: the `.mmap4` container of src/mmap.F90 and the axis presets of the
3_10 / 8_16 tables, plus the synthetic stand-in table used when no real LUT is at hand.

File format (src/mmap.F90:63-127 writer, :129-203 reader): one page (sysconf PAGESIZE) of `size_t`
    [dtype_size = 4, n_elems, n_bytes, dim1 = Ncoeff, dim2 = Nentries, 0, ...]
followed by the raw real32 array, column-major (Ncoeff, Nentries), entry index = tau fastest, then w0,
aspect_zx, g (src/optprop_base.F90:438-442).  File name: <basename>_diffuse_<D>.tau31.w020.aspect_zx23.g6.ds1000.nc.Sdiff.mmap4
(gen_lut_basename, src/optprop_LUT.F90:364-374, 453).
"""
from __future__ import annotations

import mmap

import numpy as np

from . import synthetic as S
from . import synthetic as S_

PAGESIZE = mmap.PAGESIZE


def diffuse_axes(solver="3_10"):
    """Axes of LUT_3_10 / LUT_8_16 diffuse tables (src/optprop_base.F90:200-212, 228-240)."""
    return [S.PRESET_TAU31, S.PRESET_W020, S.PRESET_ASPECT23, S.PRESET_G6]


def diffuse_lut_filename(basename, solver="3_10"):
    D = {"3_10": 10, "8_16": 16}[solver]
    ax = diffuse_axes(solver)
    return f"{basename}_diffuse_{D}.tau{len(ax[0])}.w0{len(ax[1])}.aspect_zx{len(ax[2])}.g{len(ax[3])}.ds1000.nc.Sdiff.mmap4"


def write_mmap4(path, table):
    """table: (nentries, ncoeff) C-order float32 == Fortran (ncoeff, nentries)."""
    t = np.ascontiguousarray(table, dtype=np.float32)
    header = np.zeros(PAGESIZE // 8, dtype=np.uint64)
    header[0] = 4
    header[1] = t.size
    header[2] = 4 * t.size
    header[3] = t.shape[1]
    header[4] = t.shape[0]
    with open(path, "wb") as f:
        f.write(header.tobytes())
        f.write(t.tobytes())


def write_mmap4_generated(path, nentries, ncoeff, gen, chunk=1 << 20):
    """The same container written in pieces: gen(lo, hi) -> (hi - lo, ncoeff) float32 for entries lo..hi-1.  For tables of
    the reference's full preset size (LUT_direct_3_10 Tdir: 30.9 M entries x 9 = 1.1 GB, Sdir x 30 = 3.7 GB) that should
    not be held in memory twice."""
    header = np.zeros(PAGESIZE // 8, dtype=np.uint64)
    header[0] = 4
    header[1] = nentries * ncoeff
    header[2] = 4 * nentries * ncoeff
    header[3] = ncoeff
    header[4] = nentries
    with open(path, "wb") as f:
        f.write(header.tobytes())
        for lo in range(0, nentries, chunk):
            hi = min(nentries, lo + chunk)
            t = np.ascontiguousarray(gen(lo, hi), dtype=np.float32)
            assert t.shape == (hi - lo, ncoeff)
            f.write(t.tobytes())


def hashed_table_values(lo, hi, ncoeff, salt=0):
    """Deterministic pseudo-random float32 values in [0, 1/ncoeff) for entries lo..hi-1 (a stand-in payload for tables of
    the preset size: values differ from entry to entry and from coefficient to coefficient, sums over a block stay <= 1)."""
    n = (hi - lo) * ncoeff
    h = np.arange(n, dtype=np.uint32)
    h += np.uint32((lo * ncoeff + salt * 0x51ED27) & 0xFFFFFFFF)   # flat index of (entry, coefficient), wrapping
    h *= np.uint32(2654435761)
    h ^= h >> np.uint32(15)
    h *= np.uint32(2246822519)
    h ^= h >> np.uint32(13)
    h >>= np.uint32(8)
    v = h.astype(np.float32)
    v *= np.float32(1.0 / (1 << 24) / ncoeff)
    return v.reshape(hi - lo, ncoeff)


def read_mmap4(path):
    """Returns a read-only (nentries, ncoeff) float32 memory map of a `.mmap4` table."""
    header = np.fromfile(path, dtype=np.uint64, count=PAGESIZE // 8)
    if header[0] != 4 or header[2] != 4 * header[1] or header[3] * header[4] != header[1] or header[5] != 0:
        raise ValueError(f"{path}: not a 2-D real32 mmap4 table")
    return np.memmap(path, dtype=np.float32, mode="r", offset=PAGESIZE, shape=(int(header[4]), int(header[3])))


def synthetic_diffuse_table(solver="3_10"):
    """The closed-form surrogate (synthetic.diff2diff_surrogate) evaluated on the real LUT's nodes:
    (nentries, D*D) float32 with tau fastest, i.e. the exact shape/ordering a downloaded table has."""
    tau, w0, asp, g = diffuse_axes(solver)
    D = {"3_10": 10, "8_16": 16}[solver]
    out = np.empty((len(g), len(asp), len(w0), len(tau), D * D), dtype=np.float32)
    T, W = np.meshgrid(tau, w0, indexing="xy")  # (w0, tau)
    for ig, gv in enumerate(g):
        for ia, av in enumerate(asp):
            out[ig, ia] = S.diff2diff_surrogate(solver, T, W, float(av), np.full_like(T, gv))
    return out.reshape(-1, D * D)


# ---- direct tables (6-D: tau, w0, aspect_zx, g, phi, theta) ---------------------------------------------------
def direct_axes(full=False):
    """LUT_3_10 direct axes (src/optprop_base.F90:228-235): the 4 diffuse axes + phi19, theta19 = linspace(0, 90).
    full=False gives a thinned version (same ranges) for tests and benches: the real Tdir/Sdir are 1.1 / 3.7 GB."""
    if full:
        return diffuse_axes() + [np.linspace(0, 90, 19, dtype=np.float32), np.linspace(0, 90, 19, dtype=np.float32)]
    tau, w0, asp, g = diffuse_axes()
    return [tau[::3].copy(), w0[::4].copy(), asp[[0, 6, 10, 13, 16, 22]].copy(), g[[0, 2, 5]].copy(),
            np.array([0, 45, 90], dtype=np.float32), np.array([0, 20, 40, 60, 80], dtype=np.float32)]


def synthetic_direct_tables(axes, solver="3_10"):
    """Closed-form, energy-conserving stand-ins for Tdir (S*S) and Sdir (S*D), memory order [dst*S + src]:
    beam geometry decides which face a ray leaves through, exp(-tau/mu) what survives, the scattered part
    w0*(1-t) goes to the diffuse streams with forward bias g.  sum_dst(T) + sum_dst(S) = t + w0 (1 - t) <= 1."""
    if solver == "8_16":
        return _synthetic_direct_tables_8_16(axes)
    assert solver == "3_10"
    S, D = 3, 10
    tau, w0, asp, g, phi, theta = [np.asarray(a, dtype=np.float64) for a in axes]
    TH, PH, G, A, W, T = np.meshgrid(theta, phi, g, asp, w0, tau, indexing="ij")  # tau fastest in C-order flatten
    mu = np.maximum(np.cos(np.deg2rad(np.minimum(TH, 89.0))), 0.02)
    tant = np.tan(np.deg2rad(np.minimum(TH, 89.0)))
    ux = np.minimum(1.0, A * tant * np.sin(np.deg2rad(PH)))  # horizontal shift across the box / dx
    uy = np.minimum(1.0, A * tant * np.cos(np.deg2rad(PH)))
    t = np.exp(-T / mu)
    geo = np.zeros(TH.shape + (S, S))  # [src, dst]
    geo[..., 0, 0] = (1 - ux) * (1 - uy)
    geo[..., 0, 1] = ux * (1 - uy / 2)
    geo[..., 0, 2] = uy * (1 - ux / 2)
    qbx = np.minimum(1.0, 1.0 / np.maximum(A * tant * np.sin(np.deg2rad(PH)), 1e-6))
    qby = np.minimum(1.0, 1.0 / np.maximum(A * tant * np.cos(np.deg2rad(PH)), 1e-6))
    geo[..., 1, 0] = qbx
    geo[..., 1, 1] = (1 - qbx) * (1 - uy)
    geo[..., 1, 2] = (1 - qbx) * uy
    geo[..., 2, 0] = qby
    geo[..., 2, 2] = (1 - qby) * (1 - ux)
    geo[..., 2, 1] = (1 - qby) * ux
    Tt = (t[..., None, None] * geo)  # [src, dst]
    Tdir = np.swapaxes(Tt, -1, -2).reshape(-1, S * S).astype(np.float32)  # [dst*S + src]
    _, I = S_.geometric_blocks(solver, 0.5)
    fwd = np.zeros(D)
    fwd[1] = 0.6
    fwd[[2, 3, 6, 7]] = 0.1
    share = G[..., None] * fwd + (1 - G[..., None]) * I  # [dst]
    sc = (W * (1 - t))[..., None, None] * share[..., :, None] * np.ones(S)  # [dst, src]
    Sdir = sc.reshape(-1, D * S).astype(np.float32)
    return Tdir, Sdir


def _synthetic_direct_tables_8_16(axes):
    """Stand-ins for the 8_16 tables (Tdir 64, Sdir 128 per entry) built from the 3_10 surrogate: every stream behaves
    like its parent stream of 3_10 (4 top sub-streams -> top, 2 per side -> side; the 8 top diffuse streams -> Eup / Edn),
    and what arrives on a parent is spread over its sub-streams with weights that depend on source and beam geometry --
    deliberately not invariant under the stream relabellings of dir2dir8 / dir8_to_diff16_coeff_symmetry, so that a
    wrong permutation shows.  Column sums equal the 3_10 surrogate's: sum_dst(T) + sum_dst(S) = t + w0 (1 - t)."""
    T3, S3 = synthetic_direct_tables(axes, "3_10")
    n = T3.shape[0]
    T3 = T3.reshape(n, 3, 3)      # [dst, src]
    S3 = S3.reshape(n, 10, 3)
    phi = np.asarray(axes[4], dtype=np.float64)
    theta = np.asarray(axes[5], dtype=np.float64)
    inner = n // (len(phi) * len(theta))
    ph = np.repeat(np.tile(phi, len(theta)), inner) / 90.0        # entry order: tau fastest ... phi, theta slowest
    th = np.repeat(theta, len(phi) * inner) / 90.0
    pdir = np.array([0, 0, 0, 0, 1, 1, 2, 2])                      # parent of a direct stream
    sdir = np.array([0, 1, 2, 3, 0, 1, 0, 1])                      # its sub-index
    ndir = np.array([4, 4, 4, 4, 2, 2, 2, 2])
    pdif = np.array([0, 1, 0, 1, 0, 1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
    sdif = np.array([0, 0, 1, 1, 2, 2, 3, 3, 0, 0, 0, 0, 0, 0, 0, 0])
    ndif = np.array([4, 4, 4, 4, 4, 4, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1])

    def weights(sub, nsub, src):
        """share of sub-stream `sub` (of nsub) for source stream `src`: positive, sums to 1 over sub"""
        if nsub == 1:
            return np.ones(n)
        raw = [1.0 + 0.6 * np.sin(1.3 * q + 0.7 * src + 2.0 * ph) + 0.3 * np.cos(0.9 * q * (1 + src) + 1.5 * th) for q in range(nsub)]
        return raw[sub] / np.sum(raw, axis=0)

    T8 = np.empty((n, 8, 8), dtype=np.float64)
    S8 = np.empty((n, 16, 8), dtype=np.float64)
    for s_ in range(8):
        for d in range(8):
            T8[:, d, s_] = T3[:, pdir[d], pdir[s_]] * weights(sdir[d], ndir[d], s_)
        for d in range(16):
            S8[:, d, s_] = S3[:, pdif[d], pdir[s_]] * weights(sdif[d], ndif[d], s_ + 8)
    return T8.reshape(n, 64).astype(np.float32), S8.reshape(n, 128).astype(np.float32)
