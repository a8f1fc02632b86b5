# config 4 in short: kernel time of the shared-storage steps (12 g-points under rocprofv3) and the full 252-g-point run
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/c4k
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4k -- python3 $GRAFT_REPO_ROOT/bench_specint.py --sw 6 --lw 6 --streams 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/c4k/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("tsx_k_dd_", "tsx_k_rec_", "tsx_k_scan", "pack", "lut_diff")):
        print(f"{r['Name'][:60]:60s} n={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:8.1f} us")
PY
cd $GRAFT_REPO_ROOT
python3 bench_specint.py > gpurun_out/r02/config4.json 2>/dev/null
python3 - <<'PY'
import json
x = json.loads(open("gpurun_out/r02/config4.json").read().strip().splitlines()[-1])
print([(round(c["seconds"], 2), round(c["gpoints_per_s"], 1), c["iterations_min_med_max"]) for c in x["config"]["calls"]])
PY
