"""Diagnostics that ship with the library (round 6): the reference's log events for this path and the comparison of the
device code with the file."""
import os
import sys

import numpy as np
import pytest

from tenstream_amd import lut, synthetic
from tenstream_amd.pprts import PprtsSolver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _solver(Nx=12, Ny=10, Nz=8):
    P = PprtsSolver(Nz, Nx, Ny, 100.0, 100.0, 30.0, 55.0)
    P.set_lut_diffuse(lut.synthetic_diffuse_table("3_10"), lut.diffuse_axes("3_10"))
    dax = lut.direct_axes()
    Tdir, Sdir = lut.synthetic_direct_tables(dax)
    P.set_lut_direct(Tdir, Sdir, dax)
    kabs, ksca, g = synthetic.cloud_field(Nx, Ny, Nz, seed=5)
    dz = np.full((Ny, Nx, Nz), 50.0)
    return P, (kabs * 20.0, ksca, g, dz)


def test_log_events_carry_the_reference_names_and_nest_like_the_reference(gpu):
    """solver%logs (src/pprts_base.F90:176-209): one g-point leaves one count per phase; solve_Mdiff lies inside compute_Ediff
    (src/pprts.F90:2760-2818 around :3012-3021), solve_Mdir inside compute_Edir (:2694-2756 around :2903-2912)."""
    P, (kabs, ksca, g, dz) = _solver()
    with pytest.raises(Exception):
        P.core.log_get()   # off by default: no event records on the hot path
    P.core.log_enable(True)
    for rep in range(2):
        P.set_optical_properties(0.15, kabs, ksca, g, dz)
        info = P.solve(1000.0)
        assert info.reason in (2, 3)
        P.get_result()
    ev = P.core.log_get()
    want = {"set_optprop", "get_coeff_diff2diff", "get_coeff_dir2dir", "compute_Edir", "solve_Mdir", "setup_diff_src", "compute_Ediff",
            "setup_Mdiff", "solve_Mdiff", "compute_absorption", "get_result"}
    assert set(ev) == want
    for name in want:
        assert ev[name][0] == 2 and ev[name][1] >= 0.0, (name, ev[name])
    assert 0.0 < ev["solve_Mdiff"][1] <= ev["compute_Ediff"][1]
    assert 0.0 < ev["solve_Mdir"][1] <= ev["compute_Edir"][1]
    assert ev["get_coeff_diff2diff"][1] <= ev["set_optprop"][1]
    P.core.log_enable(False)
    with pytest.raises(Exception):
        P.core.log_get()
    P.close()


def test_device_code_equals_the_file(gpu):
    """Every code object of libtsx.so, read back from device memory through its probe kernel, equals the .text of the file."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import code_verify

    P, (kabs, ksca, g, dz) = _solver()
    P.set_optical_properties(0.15, kabs, ksca, g, dz)
    info = code_verify.units()
    assert set(info) == set(code_verify.UNITS)
    bad = code_verify.verify(P.lib, info=info)
    assert bad == [], bad[:5]
    P.close()
