"""2-D x/y domain decomposition of the pprts grid: host-side restatement of `setup_coord_native`
(src/pprts_base.F90:721-828): z is never split, ranks are laid out x-fastest (rank = xi + yi*nxp),
even integer split xs = (xi*Nx)/nxp, periodic W/E/S/N neighbours (neighbors(10|16|4|22))."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Coord:
    rank: int
    nxp: int
    nyp: int
    xi: int
    yi: int
    xs: int
    xm: int
    ys: int
    ym: int
    glob_xm: int
    glob_ym: int
    west: int
    east: int
    south: int
    north: int

    @property
    def xe(self):
        return self.xs + self.xm - 1

    @property
    def ye(self):
        return self.ys + self.ym - 1


def _dims_create(nproc: int):
    """MPI_Dims_create(nproc, 2): most balanced factor pair in non-increasing order (dims(1) >= dims(2))."""
    best = (nproc, 1)
    f = 1
    while f * f <= nproc:
        if nproc % f == 0:
            best = (nproc // f, f)
        f += 1
    return best


def decompose(nproc: int):
    """(nxp, nyp): dims = [nyp, nxp] with nyp >= nxp (src/pprts_base.F90:757-763)."""
    nyp, nxp = _dims_create(nproc)
    return nxp, nyp


def coord(rank: int, nproc: int, Nx: int, Ny: int, nxp: int | None = None, nyp: int | None = None) -> Coord:
    if nxp is None or nyp is None:
        nxp, nyp = decompose(nproc)
    if nxp * nyp != nproc:
        raise ValueError("nxp*nyp != nproc")
    yi, xi = divmod(rank, nxp)
    xs = (xi * Nx) // nxp
    xm = ((xi + 1) * Nx) // nxp - xs
    ys = (yi * Ny) // nyp
    ym = ((yi + 1) * Ny) // nyp - ys
    west = ((xi - 1) % nxp) + yi * nxp
    east = ((xi + 1) % nxp) + yi * nxp
    south = xi + ((yi - 1) % nyp) * nxp
    north = xi + ((yi + 1) % nyp) * nxp
    return Coord(rank, nxp, nyp, xi, yi, xs, xm, ys, ym, Nx, Ny, west, east, south, north)


decompose.coord = coord  # convenience for bench.py
