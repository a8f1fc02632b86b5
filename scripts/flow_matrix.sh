# flow-kernel variants side by side on one box: hand-off by progress words / by granules x columns per workgroup; usage (GPU box):
# bash scripts/flow_matrix.sh "64 64" "128 64" "128 128"
for sz in "$@"; do
  for cw in 16 32; do
    for gr in 0 1; do
      echo -n "cw $cw gran $gr  "
      TSX_PCS_CFG=4,16,$cw TSX_FLOW_GRAN=$gr SHARD_MODES=wrap timeout 300 python scripts/shard_study.py $sz 64 2>&1 | grep -v amdgpu.ids
    done
  done
done
