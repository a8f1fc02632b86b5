"""Timeline of the flow kernel's work items (see scripts/flow_trace.sh).  usage (GPU box, trace build): flow_trace.py [xm ym [Nz]]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tenstream_amd import DiffuseSolver, synthetic, _lib  # noqa: E402

xm, ym = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 64)
Nz = int(sys.argv[3]) if len(sys.argv) > 3 else 64
P = synthetic.make_problem("3_10", Nx=xm, Ny=ym, Nz=Nz)
s = DiffuseSolver("3_10", Nz, xm, ym)
s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
v = np.random.default_rng(1).standard_normal(P["b"].shape)
for _ in range(3):
    s.pc_apply(v, pc=3, sweeps=27, mixed=True)
lib = _lib.load()
h = xm // 2
cw = 32 if ym * h >= 8192 else 16
ntiles = (h // cw) * ym
npass = 24  # passes 2 .. 25 of 28
n = min(npass * ntiles, 32768)
buf = np.zeros((n, 12), dtype=np.uint64)
lib.tsx_debug_flow_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.tsx_debug_flow_trace(buf.ctypes.data, n)
assert rc == 0
t = buf[:, :10].astype(np.int64)
pp = (buf[:, 10] >> np.uint64(32)).astype(np.int64)
xcc = (buf[:, 11] & np.uint64(0xFF)).astype(np.int64)
wg = (buf[:, 11] >> np.uint64(8)).astype(np.int64)
t0 = t[:, 0].min()
names = ["ticket->polled", "barrier", "loads+scan1", "barrier1", "combine+scan2+barrier2", "phase3", "drain", "barrier", "(next ticket)"]
print(f"{xm}x{ym}x{Nz}: {ntiles} tiles per pass, cw {cw}, {n} items, {len(np.unique(wg))} workgroups on {len(np.unique(xcc))} XCDs")
d = np.diff(t[:, :9], axis=1) * 10.0  # ns
print("stage medians (ns):  " + "  ".join(f"{nm} {np.median(d[:, i]):.0f}" for i, nm in enumerate(names[:8])))
print("stage p90 (ns):      " + "  ".join(f"{nm} {np.percentile(d[:, i], 90):.0f}" for i, nm in enumerate(names[:8])))
print(f"item total median {np.median(t[:, 8] - t[:, 0]) * 10:.0f} ns, p90 {np.percentile(t[:, 8] - t[:, 0], 90) * 10:.0f} ns")
for q in range(int(pp.max()) + 1):
    m = pp == q
    if not m.any():
        continue
    if q < 4 or q == pp.max():
        print(f"pass {q}: first start {(t[m, 0].min() - t0) * 10:8.0f} ns  last published {(t[m, 8].max() - t0) * 10:8.0f} ns  "
              f"median poll wait {np.median(t[m, 1] - t[m, 0]) * 10:6.0f} ns")
ends = np.array([t[pp == q, 8].max() for q in range(int(pp.max()) + 1)])
print(f"per pass (last published, differences): median {np.median(np.diff(ends)) * 10:.0f} ns; whole launch {(t[:, 8].max() - t0) * 10 / 1e3:.1f} us")
