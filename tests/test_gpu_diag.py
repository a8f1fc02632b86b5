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


def test_pool_takes_driver_memory_once_and_reuses_it(gpu, monkeypatch):
    """tsx_pool.hip: solvers come and go, the driver is asked once -- the second and third solver of the same size run in the
    first one's memory (no new slab), the pool has spent time in quarantine for what it did take, and never saw a fresh slab's
    pattern damaged; TSX_POOL=0 sends allocations straight to hipMalloc (the A/B switch of profiles/r06/DEFECT.md)."""
    import ctypes

    def stats():
        st = (ctypes.c_int64 * 8)()
        assert gpu.tsx_pool_stats(-1, st) == 0
        return [int(v) for v in st]

    P, fields = _solver(16, 12, 10)
    P.set_optical_properties(0.15, *fields)
    assert P.solve(1000.0).reason in (2, 3)
    P.close()
    s1 = stats()
    assert s1[0] >= 1 and s1[1] >= s1[2] and s1[7] > 0 and s1[4] == 0, s1
    for _ in range(2):
        P, fields = _solver(16, 12, 10)
        P.set_optical_properties(0.15, *fields)
        assert P.solve(1000.0).reason in (2, 3)
        edn = P.get_result()[0]
        P.close()
    s2 = stats()
    assert s2[0] == s1[0] and s2[1] == s1[1] and s2[2] <= s1[2], (s1, s2)   # same slabs, everything handed back
    monkeypatch.setenv("TSX_POOL", "0")
    P, fields = _solver(16, 12, 10)
    P.set_optical_properties(0.15, *fields)
    assert P.solve(1000.0).reason in (2, 3)
    edn0 = P.get_result()[0]
    P.close()
    s3 = stats()
    assert s3[0] == s2[0] and s3[2] <= s2[2], (s2, s3)
    assert np.array_equal(np.asarray(edn), np.asarray(edn0))


def test_an_iteration_replayed_from_a_graph_leaves_the_solver_usable(gpu):
    """tsx_bench_kernel(5): one BiCGStab iteration captured from the solver's stream into a hipGraph and replayed (the measurement of
    what graph replay buys, scripts/graph_ab.py).  Here: it runs next to the eager timing, and a solve after it still reaches the same
    solution -- the capture left no host-side state behind."""
    from tenstream_amd import DiffuseSolver

    P = synthetic.make_problem("3_10", Nx=32, Ny=32, Nz=16)
    s = DiffuseSolver("3_10", 16, 32, 32)
    s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    x0 = np.zeros(s.vec_shape)
    assert s.solve(P["b"], x0, rtol=1e-10, atol=1e-30).reason == 2
    eager = s.bench_kernel(1, 5)
    graph = s.bench_kernel(5, 5)
    assert eager > 0 and graph > 0
    x1 = np.zeros(s.vec_shape)
    assert s.solve(P["b"], x1, rtol=1e-10, atol=1e-30).reason == 2
    assert np.abs(x1 - x0).max() <= 1e-8 * np.abs(x0).max()
    s.close()
