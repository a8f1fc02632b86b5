# usage (GPU box): bash scripts/c4_profile.sh -- kernel time by family for 24 g-points of config 4 (one instance, both calls)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/c4k
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4k -- python3 $GRAFT_REPO_ROOT/bench_specint.py --sw 12 --lw 12 --streams 1 > /tmp/c4_line.json 2>/dev/null
python3 - <<'PY'
import csv, glob, re, collections
f = glob.glob("/tmp/c4k/*/*_kernel_stats.csv")[0]
fam = collections.OrderedDict([("passes", r"tsx_k_pcsh?_rb"), ("operator", r"tsx_k_spmv"), ("vector updates", r"tsx_k_(pupdate|supdate|xrupdate|residual|scalar|asum|defect|xplus)"),
       ("direct sweep", r"tsx_k_edir"), ("LUT lookups", r"tsx_k_lut"), ("block sharing (dd)", r"tsx_k_(dd_|scan_)"), ("record sharing", r"tsx_k_rec_"),
       ("preconditioner packing", r"tsx_k_pcsh?_pack"), ("setup_b / flx_div / result", r"tsx_k_(setup_b|flx_div|get_result|scale|accum)"), ("optical properties", r"tsx_k_(optprop|delta|edd|l1d|prep)"),
       ("conversions / copies", r"tsx_k_(convert|copy|to_f|narrow|widen|halo)")])
tot = collections.Counter(); n = collections.Counter(); other = []
for r in csv.DictReader(open(f)):
    name = r["Name"]; t = float(r["TotalDurationNs"]) / 1e6
    for k, pat in fam.items():
        if re.search(pat, name):
            tot[k] += t; n[k] += int(r["Calls"]); break
    else:
        tot["other"] += t; other.append((t, name[:70]))
s = sum(tot.values())
for k, v in tot.most_common(): print(f"{k:32s} {v:9.1f} ms {100*v/s:5.1f} %  ({n[k]} launches)")
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:int(__import__("os").environ.get("C4_TOP", "0"))]:
    print("%9.2f ms %6d x %9.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:120]))
print("other top:", sorted(other, reverse=True)[:6])
import json
x = json.loads([l for l in open("/tmp/c4_line.json") if l.startswith("{")][-1])
print([(round(c["seconds"], 2), round(c["gpoints_per_s"], 1), c["iterations_min_med_max"]) for c in x["config"]["calls"]], "device total", round(s), "ms")
PY
