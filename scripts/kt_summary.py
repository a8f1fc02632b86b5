#!/usr/bin/env python3
"""Per-kernel durations from a rocprofv3 --kernel-trace CSV, separating *working* launches from the ones that return at
once on the `done` flag (the Krylov loop enqueues up to check_every iterations ahead of the host's convergence check).

usage: kt_summary.py <kernel_trace.csv> [out.json]
A launch counts as working when it lasts at least half of the kernel's longest launch."""
import csv
import json
import re
import statistics
import sys
from collections import defaultdict


def main():
    assert sys.argv[1].endswith(".csv") and (len(sys.argv) < 3 or sys.argv[2].endswith(".json")), __doc__
    acc = defaultdict(list)
    with open(sys.argv[1], newline="") as f:
        for r in csv.DictReader(f):
            m = re.search(r"(tsx_k_\w+)(<[^(]*>)?", r["Kernel_Name"])
            if m:
                acc[m.group(0).replace(" ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    doc = {}
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        w = [x for x in v if x >= 0.5 * max(v)]
        doc[k] = {"launches": len(v), "working_launches": len(w), "avg_working_us": statistics.mean(w) / 1e3,
                  "avg_listed_us": statistics.mean(v) / 1e3, "total_ms": sum(v) / 1e6}
        print(f"{k[:80]:80s} n={len(v):5d} working={len(w):5d} avg_working={doc[k]['avg_working_us']:8.1f} us  "
              f"listed={doc[k]['avg_listed_us']:8.1f} us")
    if len(sys.argv) > 2:
        json.dump(doc, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
