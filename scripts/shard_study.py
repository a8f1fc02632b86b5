"""What one rank of a strong-scaling run pays for its exchanges: the shard of an N-rank split of the metric domain, solved by ONE
rank whose four neighbours are itself (force_halo), so that every exchange kernel, event and launch of the multi-rank path is
issued -- on a one-GPU box.  The arithmetic is that of the periodic shard domain in all three modes:

  wrap   plain periodic domain, no halo machinery (what a single GPU does)
  copy   force_halo, self neighbours through device-to-device copies
  peer   force_halo, self neighbours through the rank's own mailbox (tsx_peer.hip: send / recv kernels, sequence words)

usage (GPU box): python scripts/shard_study.py [xm ym [Nz]]   -- default 128 64 64 (a rank of the 2 x 4 split of 256 x 256 x 64)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tenstream_amd import DiffuseSolver, synthetic  # noqa: E402

xm, ym = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 64)
Nz = int(sys.argv[3]) if len(sys.argv) > 3 else 64
P = synthetic.make_problem("3_10", Nx=xm, Ny=ym, Nz=Nz)
dev = torch.device("cuda", 0)
b = torch.tensor(P["b"], device=dev)
coeff = torch.tensor(P["coeff"], device=dev)
rest = [torch.tensor(P[k], device=dev) for k in ("l1d", "a11", "a12", "albedo")]
KW = dict(pc_sweeps=int(os.environ["PC_SWEEPS"])) if os.environ.get("PC_SWEEPS") else {}   # half-grid passes - 1 (A/B)
for mode in os.environ.get("SHARD_MODES", "wrap,copy,peer").split(","):
    s = DiffuseSolver("3_10", Nz, xm, ym, force_halo=mode != "wrap")
    if mode == "peer":
        s.comm_peer_init(lambda blob: [blob])
    s.set_coeffs(coeff, *rest)
    x = torch.zeros_like(b)
    for _ in range(3):
        info = s.solve(b, x, initial_guess_zero=1, **KW)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    dev_ms = 0.0
    for _ in range(n):
        info = s.solve(b, x, initial_guess_zero=1, **KW)
        dev_ms += info.solve_ms
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    print(f"{xm}x{ym}x{Nz} {mode:5s} wall {wall:7.3f} ms  device {dev_ms / n:7.3f} ms  its {info.niter} reason {info.reason} "
          f"rel {info.rnorm / info.rnorm0:.2e}  -> {xm * ym * Nz / wall / 1e3:.1f} M cells/s per rank", flush=True)
    s.close()
