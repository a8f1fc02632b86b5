"""usage (GPU box): python scripts/quick.py [bench.py arguments] -- one line per run: value, time, iterations, pass / iteration times"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--steps", "10", "--warmup", "2"] + sys.argv[1:],
                   capture_output=True, text=True)
lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
if not lines:
    print("FAILED", p.stderr[-800:])
    sys.exit(1)
d = json.loads(lines[0])
c, r, ns = d["config"], d["roofline"], d["config"].get("no_sharing")
msg = (f"{' '.join(sys.argv[1:]):34s} {d['value']/1e6:7.1f} M cells/s {d['ms_per_step']:7.2f} ms  its {c['iterations']} / tight "
       f"{c['tight_run']['iterations']} / warm {c['warm_start']['iterations']}  iter {c['iter_ms']:.3f} ms  {r['kernel'][:13]} "
       f"{r['ms_per_launch']*1e3:6.1f} us frac {r['frac']:.3f}  spmv {d['roofline_spmv']['ms_per_launch']:.3f} ms")
if ns:
    msg += f"  | no-sharing {ns['cells_per_s']/1e6:.1f} M, {ns['iterations']} its, iter {ns['iter_ms']:.3f}, pass {ns['pass_ms']*1e3:.1f} us"
print(msg)
