#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection CSVs into a per-kernel traffic summary.

usage: pmc_summary.py <workload-key> <out.json> <fetch_counter_collection.csv> <write_counter_collection.csv>

HBM bytes per launch = 2 * FETCH_SIZE * 1024  +  WRITE_SIZE * 1024
  * FETCH_SIZE / WRITE_SIZE are reported in KiB;
  * on gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md "HBM"; re-checked for this library's
    4/8/16 B-per-lane access widths by scripts/calib_fetch.hip -> profiles/r01/calib_pmc_*.csv), hence the factor 2;
  * Infinity-Cache hits are counted by FETCH_SIZE, so this is fabric traffic, an upper bound on HBM traffic.
The two counters are collected in separate passes (they do not fit one pass).
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(tsx_k_\w+)", name)
    if not m:
        return None
    base = m.group(1)
    tm = re.search(re.escape(base) + r"<([^(]*)>\(", name)
    return base + ("<" + tm.group(1).replace(" ", "") + ">" if tm else "")


def collect(path, counter):
    acc = defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            if k:
                acc[k].append((float(row["Counter_Value"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    return acc


def main():
    key, out, fcsv, wcsv = sys.argv[1:5]
    F, W = collect(fcsv, "FETCH_SIZE"), collect(wcsv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(F) | set(W)):
        f = [v for v, _ in F.get(k, [])]
        w = [v for v, _ in W.get(k, [])]
        fm = sum(f) / len(f) if f else 0.0
        wm = sum(w) / len(w) if w else 0.0
        kernels[k] = {
            "launches": max(len(f), len(w)),
            "FETCH_SIZE_KiB_mean": fm,
            "WRITE_SIZE_KiB_mean": wm,
            "read_bytes_per_launch": 2.0 * fm * 1024.0,
            "write_bytes_per_launch": wm * 1024.0,
            "traffic_bytes_per_launch": 2.0 * fm * 1024.0 + wm * 1024.0,
        }
    json.dump({"workload": key, "correction": "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950)", "kernels": kernels},
              open(out, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k[:70]:70s} n={v['launches']:5d} traffic={v['traffic_bytes_per_launch'] / 1e6:10.1f} MB")


if __name__ == "__main__":
    main()
