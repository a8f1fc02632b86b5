# solve rate against the domain size on one GPU; usage (GPU box): bash scripts/size_scan.sh
for sz in "64 64" "128 128" "256 256" "512 512" "1024 512"; do
  set -- $sz
  python bench.py --no-cpu-baseline --steps 3 --warmup 1 --nx $1 --ny $2 > /tmp/sz.json 2>/dev/null
  python - $1 $2 <<'PY'
import json, sys
d = json.loads(open("/tmp/sz.json").read().strip().splitlines()[-1]); c = d["config"]; r = d["roofline"]
print(f"{sys.argv[1]}x{sys.argv[2]}x64: {d['value']/1e6:7.1f} M cells/s  {d['ms_per_step']:7.2f} ms  its {c['iterations']}  iter {c['iter_ms']:.3f} ms  "
      f"pass {r['ms_per_launch']*1e3:6.1f} us frac {r['frac']:.3f}  spmv frac {d['roofline_spmv']['frac']:.3f}  distinct blocks {c['coeff_dedup']['distinct_blocks']}")
PY
done
