#!/usr/bin/env python3
"""Cross-check of the FETCH_SIZE x 2 correction with the L2 -> fabric request-size counters.

usage: rdreq_summary.py <rdreq_counter_collection.csv> <traffic_xxx.json> <out.json>
The csv comes from  rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum
(one pass of its own).  Read bytes per working launch = 32 n32 + 64 n64 + 128 n128, beside 2 * FETCH_SIZE * 1024 of the
traffic summary (scripts/pmc_summary.py)."""
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


def main():
    src, traffic, out = sys.argv[1:4]
    assert src.endswith(".csv") and traffic.endswith(".json") and out.endswith(".json"), __doc__
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(src, newline="")):
        k = short(r["Kernel_Name"])
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    T = json.load(open(traffic))["kernels"]

    def wmean(v):
        mx = max(v)
        w = [x for x in v if x >= 0.5 * mx] if mx > 0 else v
        return sum(w) / len(w)

    doc = {}
    for k in sorted(acc):
        a = acc[k]
        n32, n64, n128, n = (wmean(a[c]) for c in ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum",
                                                   "TCC_EA0_RDREQ_sum"))
        exact = 32 * n32 + 64 * n64 + 128 * n128
        doc[k] = {"RDREQ": n, "RDREQ_32B": n32, "RDREQ_64B": n64, "RDREQ_128B": n128, "read_bytes_by_request_size": exact,
                  "read_bytes_2x_FETCH_SIZE": T.get(k, {}).get("read_bytes_per_launch")}
        f = doc[k]["read_bytes_2x_FETCH_SIZE"]
        print(f"{k[:64]:64s} 32B {n32/1e6:6.2f}M 64B {n64/1e6:6.2f}M 128B {n128/1e6:7.2f}M -> {exact/1e6:8.1f} MB   2xFETCH {0 if not f else f/1e6:8.1f} MB")
    json.dump(doc, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
