# one rank with itself as its four neighbours over the peer transport (scripts/shard_study.py): the passes as launches (TSX_FLOW_PEER=0)
# against the flow kernel with the faces inside it; usage (GPU box): bash scripts/flow_peer_ab.sh "128 64" ...
if [ $# -eq 0 ]; then set -- "128 64" "128 128" "256 128"; fi
for sz in "$@"; do
  for f in 0 1; do
    echo -n "TSX_FLOW_PEER=$f  "
    TSX_FLOW_PEER=$f SHARD_MODES=peer timeout 300 python scripts/shard_study.py $sz 64 2>&1 | grep -v amdgpu.ids
  done
done
