# usage (on the GPU box): ROUND=r04 bash scripts/profile_round.sh <tag> [bench.py arguments ...]
# bench line (with the CPU baseline only for tag "bench"), rocprofv3 kernel trace and the two PMC passes of the same command,
# condensed into gpurun_out/$ROUND/<tag>_{line,kernel_stats,working_launches,traffic}.*
TAG=${1:-bench}
shift
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-r04}
O=$R/gpurun_out/$ROUND
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BASE="--no-cpu-baseline"
[ "$TAG" = "bench" ] && BASE=""
python3 $R/bench.py --steps 20 --warmup 5 $BASE "$@" > $O/${TAG}_line.json 2> $O/${TAG}_err.log
rm -rf /tmp/${TAG}_kt /tmp/${TAG}_fetch /tmp/${TAG}_write
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${TAG}_kt -- python3 $R/bench.py --no-cpu-baseline --skip-no-sharing --skip-extra-legs --steps 5 --warmup 1 "$@" > $O/${TAG}_kernel_stats_line.json 2> $O/${TAG}_kt_err.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/${TAG}_fetch -- python3 $R/bench.py --no-cpu-baseline --skip-no-sharing --skip-extra-legs --steps 2 --warmup 1 --kernel-reps 4 "$@" > /dev/null 2> $O/${TAG}_fetch_err.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/${TAG}_write -- python3 $R/bench.py --no-cpu-baseline --skip-no-sharing --skip-extra-legs --steps 2 --warmup 1 --kernel-reps 4 "$@" > /dev/null 2> $O/${TAG}_write_err.log
cp $(find /tmp/${TAG}_kt -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv
python3 $R/scripts/kt_summary.py $(find /tmp/${TAG}_kt -name "*kernel_trace.csv" | head -1) $O/${TAG}_working_launches.json > $O/${TAG}_working_launches.txt
python3 $R/scripts/pmc_summary.py "$TAG $*" $O/${TAG}_traffic.json $(find /tmp/${TAG}_fetch -name "*counter_collection.csv" | head -1) $(find /tmp/${TAG}_write -name "*counter_collection.csv" | head -1) > $O/${TAG}_traffic.txt
tail -3 $O/${TAG}_err.log
