#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection CSVs into a per-kernel traffic summary.

usage: pmc_summary.py <workload-key> <out.json> <fetch_counter_collection.csv> <write_counter_collection.csv>

HBM bytes per launch = 2 * FETCH_SIZE * 1024  +  WRITE_SIZE * 1024
  * FETCH_SIZE / WRITE_SIZE are reported in KiB;
  * on gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md "HBM"; re-checked for this library's
    4/8/16 B-per-lane access widths by scripts/calib_fetch.hip -> profiles/r01/calib_pmc_*.csv), hence the factor 2;
  * Infinity-Cache hits are counted by FETCH_SIZE, so this is fabric traffic, an upper bound on HBM traffic.
The two counters are collected in separate passes (they do not fit one pass).
Launches that exit at once (the Krylov loop enqueues up to check_every iterations ahead of the host's convergence check;
those kernels return on the `done` flag) are excluded: only launches with at least half of the kernel's largest counter
value are averaged.  "iteration" = the sum over the kernels of one default BiCGStab iteration (red-black, argv[5] passes per application).
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


def compose(kernels, P):
    """traffic of one default BiCGStab iteration from the per-kernel figures; P = half-grid passes per application of M^-1"""
    # one default iteration: 2 applications of M^-1 (pass 0 without neighbours, P - 3 intermediate, the fp32 pass, the last
    # pass; P = argv[5], default 14), 2 operator applies with fused dots, 3 vector updates
    rb = r"tsx_k_pcsh?_rb<\d+,\d+,\d+,"
    if any(re.match(rb + r"true,0,\w+,2", k) for k in kernels):
        # bf16 right-hand side words: per application pass 0 (no neighbours, leaves the words), pass 1 (leaves the words),
        # P - 4 passes that read them, the fp32 pass, the last pass
        passes = [(rb + r"false,0,\w+,1", 2), (rb + r"true,0,\w+,1", 2), (rb + r"true,0,\w+,2", 2 * (P - 4)),
                  (rb + r"true,1", 2), (rb + r"true,2", 2)]
    else:
        passes = [(rb + r"false,0", 2), (rb + r"true,0", 2 * (P - 3)), (rb + r"true,1", 2), (rb + r"true,2", 2)]
    if any(k.startswith("tsx_k_xrupdate_k32") for k in kernels):   # fp32 Krylov vectors (round 3 default)
        per_iter = passes + [(r"tsx_k_spmv_w<\d+,\d+,\w+,1,\d,float,float,.*float>$", 1),
                             (r"tsx_k_spmv_w<\d+,\d+,\w+,5,\d,float,float,.*float>$", 1),
                             (r"tsx_k_psupdate_k32c<\d+,0>|tsx_k_pupdate_k32", 1), (r"tsx_k_psupdate_k32c<\d+,1>|tsx_k_supdate_k32", 1),
                             (r"tsx_k_xrupdate_k32", 1)]
        if any(k.startswith("tsx_k_psupdate_k32c") for k in kernels):
            # the vector updates leave the bf16 words: every intermediate pass of an application reads them (P - 2 of them, the
            # first without neighbours), none writes them
            passes = [(rb + r"false,0,\w+,2", 2), (rb + r"true,0,\w+,2", 2 * (P - 3)), (rb + r"true,1", 2), (rb + r"true,2", 2)]
            if any(k.startswith("tsx_k_pcs_flow") for k in kernels):
                # round 5: the P - 3 intermediate passes with neighbours run as ONE launch per application (tsx_k_pcs_flow)
                passes = [(rb + r"false,0,\w+,2", 2), (r"tsx_k_pcs_flow<", 2), (rb + r"true,1", 2), (rb + r"true,2", 2)]
            per_iter = passes + per_iter[len(per_iter) - 5:]
    else:
        per_iter = passes + [(r"tsx_k_spmv_w<\d+,\d+,\w+,1,\d,float,float", 1), (r"tsx_k_spmv_w<\d+,\d+,\w+,5,\d,float,double", 1),
                             (r"tsx_k_pupdate32", 1), (r"tsx_k_supdate", 1), (r"tsx_k_xrupdate", 1)]
    it_bytes, missing = 0.0, []
    for pat, mult in per_iter:
        hit = [v for name, v in kernels.items() if re.match(pat, name)]
        if hit:
            it_bytes += mult * hit[0]["traffic_bytes_per_launch"]
        else:
            missing.append(pat)
    return it_bytes, missing, per_iter


def main():
    if sys.argv[1] == "--recompose":   # pmc_summary.py --recompose traffic.json [passes]: renew the `iteration` entry of an existing file
        doc = json.load(open(sys.argv[2]))
        P = int(sys.argv[3]) if len(sys.argv) > 3 else 28
        it_bytes, missing, per_iter = compose(doc["kernels"], P)
        assert not missing, missing
        doc["iteration"] = {"traffic_bytes": it_bytes, "passes_per_application": P, "composition": [f"{m} x {p}" for p, m in per_iter]}
        json.dump(doc, open(sys.argv[2], "w"), indent=1)
        print(sys.argv[2], it_bytes / 1e9, "GB per iteration,", P, "passes per application")
        return
    key, out, fcsv, wcsv = sys.argv[1:5]
    assert out.endswith(".json") and fcsv.endswith(".csv") and wcsv.endswith(".csv"), "usage: key out.json fetch.csv write.csv [passes]"
    F, W = collect(fcsv, "FETCH_SIZE"), collect(wcsv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(F) | set(W)):
        f = [v for v, _ in F.get(k, [])]
        w = [v for v, _ in W.get(k, [])]
        f = [v for v in f if v >= 0.5 * max(f)] if f and max(f) > 0 else f   # working launches only
        w = [v for v in w if v >= 0.5 * max(w)] if w and max(w) > 0 else w
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
    P = int(sys.argv[5]) if len(sys.argv) > 5 else 28   # passes per application: the library's default where the scan kernels run
    it_bytes, missing, per_iter = compose(kernels, P)
    doc = {"workload": key, "correction": "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950)", "kernels": kernels}
    if not missing:
        doc["iteration"] = {"traffic_bytes": it_bytes, "passes_per_application": P, "composition": [f"{m} x {p}" for p, m in per_iter]}
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k[:70]:70s} n={v['launches']:5d} traffic={v['traffic_bytes_per_launch'] / 1e6:10.1f} MB")


if __name__ == "__main__":
    main()
