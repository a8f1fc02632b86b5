# iteration counts of 4 ranks (host-staged exchange, ranks share the GPU) against the frequency of the preconditioner's halo
# exchange; usage (GPU box): bash scripts/halo_every_study.sh [local nx] [local ny]
NX=${1:-64}; NY=${2:-64}
for v in "TSX_PC_HALO=0" "TSX_PC_HALO_EVERY=1" "TSX_PC_HALO_EVERY=2" "TSX_PC_HALO_EVERY=4" "TSX_PC_HALO_EVERY=8"; do
  env $v python bench.py --gpus 4 --transport host --nx $NX --ny $NY --steps 1 --warmup 1 --no-cpu-baseline --kernel-reps 2 > /tmp/he.json 2>/tmp/he.err
  python - "$v" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open("/tmp/he.json") if l.startswith("{")][-1]); c = d["config"]
    print(f"{sys.argv[1]:24s} {c['workload'][:60]} its {c['iterations']} tight {c['tight_run']['iterations']} relres {c['rel_residual']:.2e} ms {d['ms_per_step']:.1f}")
except Exception as e:
    print(sys.argv[1], "failed", e, open("/tmp/he.err").read()[-400:])
PY
done
