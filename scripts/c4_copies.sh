# usage (GPU box): bash scripts/c4_copies.sh -- which buffer copies config 4's loop issues (rocclr copy kernels: durations, and the kernels before each)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/c4c
rocprofv3 --kernel-trace --output-format csv -d /tmp/c4c -- python3 $GRAFT_REPO_ROOT/bench_specint.py --sw 6 --lw 6 --streams 1 --no-cpu-baseline > /tmp/c4c_line.json 2>/dev/null
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/c4c/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ctx = collections.Counter(); dur = collections.defaultdict(float)
for i, r in enumerate(rows):
    if "copyBuffer" in r["Kernel_Name"]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        prev = next((rows[j]["Kernel_Name"][:48] for j in range(i - 1, -1, -1) if "copyBuffer" not in rows[j]["Kernel_Name"]), "-")
        nxt = next((rows[j]["Kernel_Name"][:48] for j in range(i + 1, len(rows)) if "copyBuffer" not in rows[j]["Kernel_Name"]), "-")
        k = (prev, nxt, "big" if d > 20 else "small", r.get("Grid_Size", r.get("Grid_Size_X", "?")))
        ctx[k] += 1; dur[k] += d
for k, n in sorted(ctx.items(), key=lambda kv: -dur[kv[0]])[:40]:
    print(f"{dur[k]/1e3:8.2f} ms {n:5d} x {dur[k]/n:8.1f} us grid {k[3]:>9s}  after {k[0]:48s} before {k[1]}")
PY
