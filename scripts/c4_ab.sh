# usage (GPU box): bash scripts/c4_ab.sh -- config 4 (four instances, then one) on ONE box: the round's library as it is, without the
# groupings taken over between g-points (TSX_DEDUP_REUSE=0), and with every allocation straight from the driver (TSX_POOL=0)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "" "TSX_DEDUP_REUSE=0" "TSX_POOL=0"; do
  for st in 4 1; do
    env $v python3 bench_specint.py --streams $st --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${v:-default}', 'streams', $st, round(d['value'],1), [round(x['gpoints_per_s'],1) for x in d['config']['calls']])"
  done
done
done
