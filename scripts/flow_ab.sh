# A/B of the flow kernel (TSX_PC_FLOW=1, one launch for the intermediate passes) against a launch per pass (TSX_PC_FLOW=0) on
# periodic single-rank domains; usage (GPU box): bash scripts/flow_ab.sh [sizes...]   e.g. "128 64" "128 128" "256 256"
if [ $# -eq 0 ]; then set -- "64 64" "128 64" "128 128" "256 128" "256 256"; fi
for sz in "$@"; do
  for f in 0 1; do
    echo -n "TSX_PC_FLOW=$f  "
    TSX_PC_FLOW=$f SHARD_MODES=wrap timeout 300 python scripts/shard_study.py $sz 64
  done
done
