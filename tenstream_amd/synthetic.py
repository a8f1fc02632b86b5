"""Synthetic inputs for tests and bench.py ("data": "synthetic").

No LUT binaries ship with the reference (they are downloaded), so the transport
coefficients come from a closed-form, energy-conserving surrogate of the box Monte-Carlo tables:

    c(src, dst) = t * G0(src, dst) + E * (g * G0(src, dst) + (1 - g) * I(dst))

      t  = exp(-tau * l)                     un-collided transmission, l = 2 / (1 + 2 a) mean chord / dz
      E  = w0 (1 - t) (1 - q) / (1 - w0 q)   escape after >= 1 scattering, q = 1 - (1 - t) / (tau l)
      G0 = geometric (tau -> 0) face-to-face transfer from view factors (rows sum to 1)
      I  = isotropic re-emission by exit-face area share

Rows satisfy sum_dst c(src, dst) <= 1 (== 1 for w0 = 1), like the real tables
(src/optprop_LUT.F90:1598-1612 checks exactly that), and G0 at aspect 0.5 reproduces the structure of the
reference's known-answer block (tests/test_boxmc_3_10/test_boxmc_3_10.F90:151-235).  This is host-side
input generation, not part of the solver.
"""
from __future__ import annotations

import numpy as np

# LUT axes of the reference (src/optprop_parameters.F90:145-154 tau31, :194-199 w020, :107-110 aspect23, :245 g6)
PRESET_TAU31 = np.array([
    1e-10, 3.62266272998e-07, 7.04565803675e-06, 4.47545500233e-05, 0.000172126759821, 0.000495994753047,
    0.00119161313679, 0.00251026980343, 0.00480799264297, 0.00856221891924, 0.0143961482731, 0.0231530284254,
    0.0358868239775, 0.0541358315379, 0.079959118223, 0.11623968405, 0.167882053841, 0.246414427244,
    0.350199325489, 0.502459974196, 0.759082408765, 1.08083180518, 1.5415157991, 2.19832932733, 3.04549626819,
    4.27145477454, 6.16953841432, 9.43719309835, 15.7335501106, 29.5819342206, 100.0], dtype=np.float32)
PRESET_W020 = np.array([
    0.0, 0.152960717624, 0.295085090042, 0.416951893959, 0.521358613652, 0.610087211908, 0.684967634054,
    0.747886390181, 0.800286677013, 0.84336972609, 0.878674797098, 0.906377786525, 0.928097831502,
    0.943463164595, 0.954135786554, 0.963824066888, 0.972632134967, 0.981529289348, 0.990759644674, 0.99999],
    dtype=np.float32)
PRESET_ASPECT23 = np.array([
    0.02, 0.032, 0.042, 0.056, 0.075, 0.1, 0.133, 0.178, 0.237, 0.316, 0.422, 0.562, 0.75, 1., 1.25, 1.562, 1.953,
    2.441, 3.052, 3.815, 4.768, 5.96, 7.451], dtype=np.float32)
PRESET_G6 = np.array([0.0, 0.2424, 0.4137, 0.5717, 0.7144, 0.85], dtype=np.float32)


def stream_layout(solver):
    """(ntop, nside, top_inward, side_inward): src/pprts.F90:332-349 (3_10), :413-425 (8_16)."""
    if solver in ("3_10", 310):
        return 2, 4, [False, True], [False, True, False, True]
    if solver in ("8_16", 816):
        return 8, 4, [False, True] * 4, [False, True, False, True]
    raise ValueError(solver)


def _view_factor_parallel_rect(X, Y):
    """View factor between identical, directly opposed rectangles (a x b) a distance c apart, X=a/c, Y=b/c."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    t1 = np.log(np.sqrt((1 + X * X) * (1 + Y * Y) / (1 + X * X + Y * Y)))
    t2 = X * np.sqrt(1 + Y * Y) * np.arctan(X / np.sqrt(1 + Y * Y))
    t3 = Y * np.sqrt(1 + X * X) * np.arctan(Y / np.sqrt(1 + X * X))
    return 2.0 / (np.pi * X * Y) * (t1 + t2 + t3 - X * np.arctan(X) - Y * np.arctan(Y))


def geometric_blocks(solver, aspect):
    """G0[src, dst] (rows sum to 1) and I[dst] (sums to 1) for a box with dz/dx = aspect, dx = dy."""
    ntop, nside, tin, sin_ = stream_layout(solver)
    D = ntop + 2 * nside
    a = float(aspect)
    F_tt = float(_view_factor_parallel_rect(1.0 / a, 1.0 / a))  # top <-> bottom face
    f_v = 1.0 / (1.0 + 1.3875 * a)  # side-entering stream leaving through its top/bottom target face
    f_o = 0.269 * (1.0 - f_v)  # ... through the opposite side face
    f_p = 0.3655 * (1.0 - f_v)  # ... through each perpendicular side face
    nts = ntop // 2  # top stream pairs
    G = np.zeros((D, D))
    # helper: side dofs are [x: (-x,down) (+x,down) (-x,up) (+x,up)], [y: same]; first half downward-tilted
    # (src/boxmc_3_10.inc:36-52, used at src/pprts.F90:4922-4926)
    def side(dof_axis, q):
        return ntop + dof_axis * nside + q
    for s in range(ntop):
        up = not tin[s]
        G[s, s] = F_tt
        tilt = [2, 3] if up else [0, 1]  # upward stream feeds upward-tilted side streams
        for ax in (0, 1):
            for q in tilt:
                G[s, side(ax, q)] = (1.0 - F_tt) / 4.0
    for ax in (0, 1):
        for q in range(nside):
            s = side(ax, q)
            down = q < nside // 2
            # vertical exit: all top streams of that direction share it equally
            tops = [d for d in range(ntop) if tin[d] == down]
            for d in tops:
                G[s, d] = f_v / len(tops)
            G[s, s] = f_o  # keeps direction and tilt, leaves through the opposite face
            for q2 in ([0, 1] if down else [2, 3]):
                G[s, side(1 - ax, q2)] = f_p
    G /= G.sum(axis=1, keepdims=True)
    I = np.zeros(D)
    denom = 2.0 + 4.0 * a
    for d in range(ntop):
        I[d] = (1.0 / denom) / nts
    for ax in (0, 1):
        for q in range(nside):
            I[side(ax, q)] = a / (2.0 * denom)
    I /= I.sum()
    return G, I


def diff2diff_surrogate(solver, tau, w0, aspect, g):
    """Per-cell D*D blocks, float32, memory order c[..., dst*D + src] (src fastest, src/pprts.F90:3471)."""
    G0, I = geometric_blocks(solver, aspect)
    D = G0.shape[0]
    tau = np.asarray(tau, dtype=np.float64)
    w0 = np.asarray(w0, dtype=np.float64)
    g = np.asarray(g, dtype=np.float64)
    ell = 2.0 / (1.0 + 2.0 * float(aspect))
    tl = np.maximum(tau * ell, 1e-12)
    t = np.exp(-tl)
    q = 1.0 + np.expm1(-tl) / tl  # 1 - (1 - t)/tl
    E = w0 * (1.0 - t) * (1.0 - q) / (1.0 - w0 * q)
    A = (t + E * g).astype(np.float32)[..., None, None]
    B = (E * (1.0 - g)).astype(np.float32)[..., None, None]
    GT = G0.T.astype(np.float32)  # [dst, src]
    IT = np.repeat(I[:, None], D, axis=1).astype(np.float32)  # [dst, src]
    c = A * GT + B * IT
    return c.reshape(tau.shape + (D * D,))


def delta_scale(kabs, ksca, g, f=None):
    """delta_scale + delta_scale_optprop with f = g**2 (src/helper_functions.fypp:1622-1666)."""
    kabs = np.array(kabs, dtype=np.float64, copy=True)
    ksca = np.array(ksca, dtype=np.float64, copy=True)
    g = np.array(g, dtype=np.float64, copy=True)
    f = g ** 2 if f is None else np.broadcast_to(np.asarray(f, dtype=np.float64), g.shape).copy()
    dtau = kabs + ksca
    act = dtau >= np.finfo(np.float64).eps
    w0 = np.where(act, ksca / np.where(act, dtau, 1.0), 0.0)
    big = g >= 1.0 - np.finfo(np.float64).eps * 10
    dtau_b = dtau * (1.0 - w0)
    dtau_n = dtau * (1.0 - w0 * f)
    g_n = (g - f) / np.where(big, 1.0, 1.0 - f)
    w0_n = w0 * (1.0 - f) / np.where(big, 1.0, (1.0 - f * w0))
    dtau2 = np.where(big, dtau_b, dtau_n)
    w02 = np.where(big, 0.0, w0_n)
    g2 = np.where(big, 0.0, g_n)
    kabs_o = np.where(act, dtau2 * (1.0 - w02), kabs)
    ksca_o = np.where(act, dtau2 * w02, ksca)
    g_o = np.where(act, g2, g)
    return kabs_o, ksca_o, g_o


def cloud_field(Nx, Ny, Nz, seed=20240611, cover=0.3, cld_layers=None, sigma=8.0, heterogeneous=False):
    """clear sky kabs = ksca = 1e-5, g = 0; thresholded smooth random cloud field.
    heterogeneous=True: every cell gets optical properties of its own -- log-normal "humidity" noise (sigma 0.25, smooth over
    ~4 cells horizontally and 2 levels, times white noise of 3 %) on the absorption and scattering of the background and the
    clouds alike, as an LES humidity / aerosol field has it.  No two cells then share a transport block (the general case
    the shared-block storage does not help with)."""
    from scipy.ndimage import gaussian_filter

    rng = np.random.default_rng(seed)
    kabs = np.full((Ny, Nx, Nz), 1e-5)
    ksca = np.full((Ny, Nx, Nz), 1e-5)
    g = np.zeros((Ny, Nx, Nz))
    if cld_layers is None:
        k0 = max(0, int(round(Nz * 20 / 64)))
        k1 = max(k0 + 1, int(round(Nz * 36 / 64)))
        cld_layers = range(k0, min(Nz, k1))
    noise = gaussian_filter(rng.standard_normal((Ny, Nx)), sigma=min(sigma, max(1.0, min(Nx, Ny) / 8.0)), mode="wrap")
    thresh = np.quantile(noise, 1.0 - cover)
    mask2d = noise > thresh
    for k in cld_layers:
        ks = rng.uniform(5e-3, 3e-2, size=(Ny, Nx))
        ksca[..., k] = np.where(mask2d, ks, ksca[..., k])
        kabs[..., k] = np.where(mask2d, 1e-6 * ks, kabs[..., k])
        g[..., k] = np.where(mask2d, 0.85, g[..., k])
    if heterogeneous:
        rng2 = np.random.default_rng(seed + 1)
        for a in (kabs, ksca):
            smooth = gaussian_filter(rng2.standard_normal((Ny, Nx, Nz)), sigma=(4.0, 4.0, 2.0), mode="wrap")
            smooth *= 0.25 / max(float(smooth.std()), 1e-30)
            a *= np.exp(smooth + 0.03 * rng2.standard_normal((Ny, Nx, Nz)))
    return kabs, ksca, g


def solar_source(solver, kabs, ksca, g, dz, dx, albedo, theta0_deg=40.0, S0=1000.0):
    """A cheap stand-in for edir + setup_b (src/pprts.F90:4684-4846): column-wise Beer-Lambert direct beam,
    first-scatter source spread over the streams leaving each cell.  Gives a physically shaped, positive RHS in W."""
    ntop, nside, tin, sin_ = stream_layout(solver)
    D = ntop + 2 * nside
    Ny, Nx, Nz = kabs.shape
    mu0 = np.cos(np.deg2rad(theta0_deg))
    tau = (kabs + ksca) * dz
    w0 = ksca / np.maximum(kabs + ksca, 1e-300)
    A = dx * dx
    edir = np.empty((Ny, Nx, Nz + 1))
    edir[..., 0] = S0 * A
    edir[..., 1:] = S0 * A * np.exp(-np.cumsum(tau, axis=-1) / mu0)
    scat = (edir[..., :-1] - edir[..., 1:]) * w0  # W scattered in the cell
    _, I = geometric_blocks(solver, dz / dx)
    fwd = np.zeros(D)
    for d in range(ntop):
        if tin[d]:
            fwd[d] = 1.0 / (ntop // 2)
    share = g[..., None] * fwd + (1.0 - g[..., None]) * I  # (Ny,Nx,Nz,D)
    src = scat[..., None] * share
    b = np.zeros((Ny, Nx, Nz + 1, D))
    for d in range(ntop):
        if tin[d]:
            b[:, :, 1:, d] += src[..., d]
        else:
            b[:, :, :-1, d] += src[..., d]
    for ax, shift_axis in ((0, 1), (1, 0)):
        for q in range(nside):
            d = ntop + ax * nside + q
            if sin_[q]:
                b[:, :, :-1, d] += np.roll(src[..., d], 1, axis=shift_axis)  # leaves through face i+1 / j+1
            else:
                b[:, :, :-1, d] += src[..., d]
    for d in range(ntop):
        if not tin[d]:
            b[:, :, Nz, d] += edir[..., Nz] * albedo / (ntop // 2)
    return b


def make_problem(solver="3_10", Nx=32, Ny=32, Nz=16, dx=100.0, dz=50.0, albedo=0.1, seed=20240611, n1d=0,
                 coeff_dtype=np.float32):
    """Everything the diffuse seam needs, in the reference's layouts (reversed C-order axes)."""
    ntop, nside, tin, sin_ = stream_layout(solver)
    D = ntop + 2 * nside
    kabs, ksca, g = cloud_field(Nx, Ny, Nz, seed=seed)
    kabs, ksca, g = delta_scale(kabs, ksca, g)
    tau = np.clip(((kabs + ksca) * dz).astype(np.float32), PRESET_TAU31[0], PRESET_TAU31[-1])
    w0 = np.clip((ksca / np.maximum(kabs + ksca, np.finfo(np.float64).eps)).astype(np.float32), PRESET_W020[0],
                 PRESET_W020[-1])
    aspect = max(float(np.float32(dz / dx)), float(PRESET_ASPECT23[0]))
    coeff = diff2diff_surrogate(solver, tau, w0, aspect, g.astype(np.float32)).astype(coeff_dtype)
    alb = np.full((Ny, Nx), albedo, dtype=np.float64)
    l1d = np.zeros(Nz, dtype=np.uint8)
    l1d[:n1d] = 1
    # two-stream-like 1-D coefficients for the l1d layers (any 0 < a11 + a12 <= 1 is a valid operator input)
    t1 = np.exp(-1.66 * (kabs + ksca) * dz)
    a11 = np.ascontiguousarray(t1 * 0.9)
    a12 = np.ascontiguousarray((1.0 - t1) * 0.4)
    b = solar_source(solver, kabs, ksca, g, dz, dx, alb)
    # 1-D layers only couple the top streams; their side rows are identity rows with zero source
    # (setup_b writes a13/a23 terms to top streams only, src/pprts.F90:4709-4721)
    for k in range(n1d):
        top_only = b[:, :, k, :ntop].copy()
        b[:, :, k, :] = 0.0
        b[:, :, k, :ntop] = top_only
    return dict(solver=solver, Nx=Nx, Ny=Ny, Nz=Nz, D=D, dx=dx, dz=dz, coeff=np.ascontiguousarray(coeff), l1d=l1d,
                a11=a11, a12=a12, albedo=alb, b=np.ascontiguousarray(b), kabs=kabs, ksca=ksca, g=g)
