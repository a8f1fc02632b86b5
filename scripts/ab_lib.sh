# usage (GPU box): bash scripts/ab_lib.sh <reps> <dir> [<dir> ...]   -- A/B of builds of libtsx on ONE box: scripts/pcsbench.py
# (solve, application of M^-1 and intermediate pass on the metric domain) with tenstream_amd/<dir>/libtsx.so, round robin
reps=$1; shift
cd $GRAFT_REPO_ROOT
for r in $(seq $reps); do
  for v in "$@"; do
    TSX_PROBE_LIB=tenstream_amd/$v/libtsx.so CFGS="${CFGS:-4,16,32}" python3 - <<'PY' 2>&1 | grep -v amdgpu
import json, os, runpy, sys, io, contextlib
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import tenstream_amd._lib as L
L.LIB_PATH = os.path.join(os.environ["GRAFT_REPO_ROOT"], os.environ["TSX_PROBE_LIB"])
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path(os.path.join(os.environ["GRAFT_REPO_ROOT"], "scripts", "pcsbench.py"), run_name="__main__")
for line in buf.getvalue().splitlines():
    if line.startswith("{"):
        o = json.loads(line)
        print(f"{os.environ['TSX_PROBE_LIB']:36s} its {o['its']} solve {o['solve_ms']:.3f} ms  M^-1 {o['pc_apply_ms']:.4f} ms  pass {1e3 * o.get('pass_ms', 0):.2f} us  operator {o['spmv_ms']:.4f} ms")
PY
  done
done
