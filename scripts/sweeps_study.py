"""Which pass count should be the library default?  Config 4's 252 g-points (optical depths scaled over 4.5 decades, solar
and thermal, 1-D layers on top) with pc_sweeps = 9 / 11 / 13 / 15: total time and iteration statistics of both calls."""
import json, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sw in [int(v) for v in os.environ.get("SWEEPS", "9,11,13,15").split(",")]:
    out = subprocess.run([sys.executable, os.path.join(root, "bench_specint.py"), "--pc-sweeps", str(sw), "--calls", "2"],
                         capture_output=True, text=True)
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    print(sw, [(round(c["seconds"], 2), c["iterations_min_med_max"], round(c["diffuse_solve_ms_total"])) for c in d["config"]["calls"]], flush=True)
