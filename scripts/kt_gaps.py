#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV: where the device waits for the host.

usage: kt_gaps.py <kernel_trace.csv> [burst_gap_us=300]
Launches are sorted by start time and cut into bursts wherever the device idles longer than burst_gap_us (between solves the
host does other work); inside the bursts every gap is attributed to the kernel that FOLLOWS it."""
import csv
import re
import sys
from collections import defaultdict


def main():
    cut = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 300e3
    rows = []
    with open(sys.argv[1], newline="") as f:
        for r in csv.DictReader(f):
            m = re.search(r"(tsx_k_\w+)(<[^(]*>)?", r["Kernel_Name"])
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0).replace(" ", "") if m else r["Kernel_Name"][:40]))
    rows.sort()
    busy = idle = 0
    nb = 1
    by = defaultdict(lambda: [0, 0, 0])  # launches, summed gap, gaps above 2 us
    hist = defaultdict(int)
    end = rows[0][1]
    for (s0, e0, k) in rows[1:]:
        g = s0 - end
        if g > cut:
            nb += 1
        else:
            g = max(g, 0)
            idle += g
            by[k][0] += 1
            by[k][1] += g
            by[k][2] += g > 2000
            hist[min(int(g // 1000), 50)] += 1
        busy += e0 - s0
        end = max(end, e0)
    print(f"{len(rows)} launches in {nb} bursts: busy {busy / 1e6:.2f} ms, idle inside bursts {idle / 1e6:.2f} ms ({100.0 * idle / (busy + idle):.1f} %)")
    print("gap histogram (us: launches):", " ".join(f"{k}:{v}" for k, v in sorted(hist.items())))
    for k, (n, g, big) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"{k[:84]:84s} n={n:5d} gap before: avg {g / n / 1e3:7.2f} us  total {g / 1e6:7.3f} ms  >2us: {big}")


if __name__ == "__main__":
    main()
