# one rank with itself as its four neighbours over the peer transport (scripts/shard_study.py) on shards whose passes are NOT resident at
# once: the passes as launches that send for themselves (TSX_FLOW_PEER=0: what round 5 ran there) against the flow kernel's lean body
# with the rank faces inside it (round 6).  usage (GPU box): bash scripts/flow_peer_lean_ab.sh ["256 128" ...]
if [ $# -eq 0 ]; then set -- "256 128" "128 256" "256 256" "128 64"; fi
for sz in "$@"; do
  for f in 0 1 0 1; do
    echo -n "TSX_FLOW_PEER=$f  "
    TSX_FLOW_PEER=$f SHARD_MODES=peer timeout 300 python scripts/shard_study.py $sz 64 2>&1 | grep -v amdgpu.ids
  done
  echo -n "periodic (no halo)  "; SHARD_MODES=wrap timeout 300 python scripts/shard_study.py $sz 64 2>&1 | grep -v amdgpu.ids
done
