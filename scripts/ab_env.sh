# usage (GPU box): bash scripts/ab_env.sh VAR "v1 v2 ..." [bench.py arguments]  -- A/B of one environment switch in bench.py
VAR=$1; VALS=$2; shift 2
mkdir -p gpurun_out/ab
for v in $VALS; do
  env $VAR=$v python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" > gpurun_out/ab/${VAR}_$v.json 2>/dev/null
  python - "$VAR" "$v" <<'PY'
import json, sys
var, v = sys.argv[1:3]
d = json.loads(open(f"gpurun_out/ab/{var}_{v}.json").read().strip().splitlines()[-1]); c = d["config"]
r, rp = d["roofline"], d.get("roofline_pc") or {}
print(f"{var}={v}: {d['value']/1e6:7.1f} M cells/s  {d['ms_per_step']:6.2f} ms  its {c['iterations']} / tight {c['tight_run']['iterations']}"
      f"  iter {c['iter_ms']:.3f} ms  pass {r['ms_per_launch']*1e3:6.1f} us frac {r['frac']:.3f}  M^-1 {rp.get('ms_per_application', 0):.3f} ms")
PY
done
