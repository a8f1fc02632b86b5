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
