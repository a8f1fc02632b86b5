"""What the first multi-GPU run of bench.py should show, from one-GPU measurements (no multi-GPU box on this pool).

For the strong-scaling metric (256 x 256 x 64 on N = 1, 2, 4, 8 GPUs; decompose(): 1 x 1, 2 x 1, 2 x 2, 2 x 4 ranks, x fastest) and for
config 3 (512 x 512 x 64 on 2 x 4) a rank's shard is solved by ONE rank whose four neighbours are itself over the peer transport
(scripts/shard_study.py: every pack, send, tag and wait of the multi-rank path is executed, the arithmetic is the periodic
shard's) -- that is the per-rank time with a zero-latency link.  The projection adds, per dependent hand-off that crosses a
device boundary, the extra latency of an xGMI hop over a local one (assumption, stated in the output: 1.5 us per hop, the
MI355X guide's cross-XCD / cross-device figures are 0.1-0.3 us and 1-2 us): per iteration 2 x (passes - 1) hand-offs of the
preconditioner, 2 of the operator and 3-4 all-reduces.

usage (GPU box): python scripts/scale_projection.py [out.json]"""
import json
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "r06", "scale_projection.json")
HOP_US = 1.5
PASSES = 28


def shard(xm, ym, mode):
    env = dict(os.environ, SHARD_MODES=mode)
    p = subprocess.run([sys.executable, os.path.join(root, "scripts", "shard_study.py"), str(xm), str(ym), "64"], env=env,
                       capture_output=True, text=True, timeout=600)
    m = re.search(r"wall\s+([\d.]+) ms\s+device\s+([\d.]+) ms\s+its (\d+)", p.stdout)
    if not m:
        raise RuntimeError(p.stdout + p.stderr)
    return {"wall_ms": float(m.group(1)), "device_ms": float(m.group(2)), "iterations": int(m.group(3))}


def project(label, gx, gy, grids):
    rows = []
    for n, (npx, npy) in grids:
        xm, ym = gx // npx, gy // npy
        if n == 1:
            r = shard(xm, ym, "wrap")
            rows.append({"gpus": 1, "grid": "1x1", "shard": f"{xm}x{ym}x64", "measured_ms": r["wall_ms"], "iterations": r["iterations"],
                         "projected_ms": r["wall_ms"], "cells_per_s": gx * gy * 64 / r["wall_ms"] * 1e3, "speedup": 1.0})
            continue
        w, p = shard(xm, ym, "wrap"), shard(xm, ym, "peer")
        hops = p["iterations"] * (2 * (PASSES - 1) + 2 + 4)
        proj = p["wall_ms"] + hops * HOP_US * 1e-3
        rows.append({"gpus": n, "grid": f"{npx}x{npy}", "shard": f"{xm}x{ym}x64", "shard_periodic_ms": w["wall_ms"],
                     "shard_with_its_exchanges_ms": p["wall_ms"], "iterations": p["iterations"], "dependent_handoffs": hops,
                     "projected_ms": proj, "cells_per_s": gx * gy * 64 / proj * 1e3})
    t1 = rows[0]["projected_ms"]
    for r in rows:
        r["speedup"] = t1 / r["projected_ms"] * (r["gpus"] if label.startswith("weak") else 1.0) if label.startswith("strong") else None
    return rows


doc = {"assumption": f"{HOP_US} us extra per dependent hand-off that crosses xGMI instead of staying on the device; everything else "
                     "as measured on ONE MI355X with self neighbours over the peer transport (scripts/shard_study.py)",
       "strong_256x256x64": project("strong", 256, 256, [(1, (1, 1)), (2, (2, 1)), (4, (2, 2)), (8, (2, 4))])}
# config 3: 512 x 512 x 64 on 2 x 4 (256 x 128 per rank) against ONE GPU solving 512 x 512 x 64
c3 = project("config3", 512, 512, [(1, (1, 1)), (8, (2, 4))])
c3[1]["speedup"] = c3[0]["projected_ms"] / c3[1]["projected_ms"]
doc["config3_512x512x64"] = c3
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(doc, open(out, "w"), indent=1)
for k in ("strong_256x256x64", "config3_512x512x64"):
    print(k)
    for r in doc[k]:
        print("  ", {a: (round(b, 3) if isinstance(b, float) else b) for a, b in r.items()})
