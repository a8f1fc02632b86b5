# usage (GPU box): bash scripts/bench_sizes.sh  -- bench.py's line at the metric domain (all legs) and at 128 x 128 / 64 x 64 (headline leg only), condensed
O=gpurun_out/r05; mkdir -p $O
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_quick_line.json 2> $O/bench_quick_err.log; tail -2 $O/bench_quick_err.log
python bench.py --no-cpu-baseline --skip-extra-legs --skip-no-sharing --steps 20 --warmup 5 --nx 128 --ny 128 > $O/c2_quick_line.json 2>/dev/null
python bench.py --no-cpu-baseline --skip-extra-legs --skip-no-sharing --steps 20 --warmup 5 --nx 64 --ny 64 > $O/s64_quick_line.json 2>/dev/null
python - <<PY
import json
for f in ("bench_quick","c2_quick","s64_quick"):
    d=json.loads(open(f"gpurun_out/r05/{f}_line.json").read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
    print(f, round(d["value"]/1e6,1), "M", round(d["ms_per_step"],3), "ms its", c["iterations"], "spread", c["step_ms_min_med_max"], "first", c["first_solve"], "nohint", c["no_hint"] and (round(c["no_hint"]["cells_per_s"]/1e6,1), c["no_hint"]["step_ms_min_med_max"]))
    print("   roofline", r["kernel"][:40], round(r["ms_per_launch"]*1e3,1), "us frac", round(r["frac"],3), "per pass", r.get("us_per_pass"), "own launch", r.get("the_same_pass_as_its_own_launch_us"))
    print("   flow", c["flow_kernel"])
    for k in ("no_sharing","heterogeneous"):
        if c.get(k): print("  ", k, round(c[k]["cells_per_s"]/1e6,1))
    if c.get("all_fp64"): print("   all_fp64", {k:(round(v["cells_per_s"]/1e6,1), v["iterations"]) for k,v in c["all_fp64"].items() if isinstance(v,dict)})
PY
