# the default solver on other realisations of the synthetic field (cloud cover 0 .. 1, other seeds): iteration counts and rate
for args in "--cover 0.0" "--cover 0.1" "--cover 0.3" "--cover 0.3 --seed 7" "--cover 0.3 --seed 99" "--cover 0.6" "--cover 1.0" "--cover 0.3 --solver 8_16 --seed 7" "--cover 1.0 --solver 8_16"; do
  python bench.py --no-cpu-baseline --steps 3 --warmup 1 $args > /tmp/fs.json 2>/dev/null
  python - "$args" <<'PY'
import json, sys
d = json.loads(open("/tmp/fs.json").read().strip().splitlines()[-1]); c = d["config"]
print(f"{sys.argv[1]:36s} {d['value']/1e6:7.1f} M cells/s {d['ms_per_step']:7.2f} ms  its {c['iterations']} (reason {c['reason']})  tight {c['tight_run']['iterations']}  warm {c['warm_start']['iterations']}  distinct blocks {c['coeff_dedup']['distinct_blocks']} shared {c['coeff_dedup']['in_use']}")
PY
done
