"""Regime stress (one GPU): perfectly reflecting surface with conservative scattering, optically thick / thin, purely
absorbing -- default solver against a sparse direct solve of the oracle CSR."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tenstream_amd import DiffuseSolver, synthetic
from oracle import oracle as O
import scipy.sparse.linalg as spla
for alb, scale_sca, scale_abs in ((1.0, 1.0, 0.0), (1.0, 30.0, 0.0), (0.0, 1.0, 1.0), (0.5, 0.0, 1.0), (0.999, 100.0, 1e-3), (0.1, 1e-4, 1e-4)):
    P = synthetic.make_problem("3_10", Nx=16, Ny=12, Nz=10, albedo=alb)
    # rebuild coefficients with scaled optical properties through the same surrogate
    kabs, ksca, g = synthetic.cloud_field(16, 12, 10)
    kabs = kabs * scale_abs; ksca = ksca * scale_sca
    kabs, ksca, g = synthetic.delta_scale(kabs, ksca, g)
    tau = np.clip(((kabs + ksca) * 50.0).astype(np.float32), synthetic.PRESET_TAU31[0], synthetic.PRESET_TAU31[-1])
    w0 = np.clip((ksca / np.maximum(kabs + ksca, 1e-300)).astype(np.float32), synthetic.PRESET_W020[0], synthetic.PRESET_W020[-1])
    coeff = synthetic.diff2diff_surrogate("3_10", tau, w0, np.float32(0.5), np.clip(g.astype(np.float32), 0, 0.85)).astype(np.float32)
    lay = O.layout("3_10", 10, 16, 12)
    albf = np.full((12, 16), alb)
    b = synthetic.solar_source("3_10", kabs, ksca, g, 50.0, 100.0, albf)
    s = DiffuseSolver("3_10", 10, 16, 12)
    s.set_coeffs(coeff, P["l1d"], P["a11"], P["a12"], albf)
    x = np.zeros(s.vec_shape)
    info = s.solve(b, x, rtol=1e-8, atol=1e-30, maxit=500)
    A = O.assemble_csr(lay, coeff.astype(np.float64), P["l1d"], P["a11"], P["a12"], albf)
    xr = spla.spsolve(A.tocsc(), b.ravel()).reshape(b.shape)
    print(json.dumps(dict(albedo=alb, sca=scale_sca, abs=scale_abs, reason=info.reason, its=info.niter, err=float(np.abs(x-xr).max()/max(np.abs(xr).max(),1e-300)), colsum_max=float(coeff.reshape(12,16,10,10,10).sum(axis=3).max()))))
    s.close()
