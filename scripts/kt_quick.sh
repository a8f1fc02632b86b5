# usage (GPU box): bash scripts/kt_quick.sh <tag> [bench.py arguments]  -- rocprofv3 kernel trace of a short bench run, working launches per kernel
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/${TAG}_kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${TAG}_kt -- python3 $R/bench.py --no-cpu-baseline --skip-no-sharing --skip-extra-legs --steps 5 --warmup 1 "$@" > $O/${TAG}_kernel_stats_line.json 2> $O/${TAG}_kt_err.log
cp $(find /tmp/${TAG}_kt -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv
python3 $R/scripts/kt_summary.py $(find /tmp/${TAG}_kt -name "*kernel_trace.csv" | head -1) $O/${TAG}_working_launches.json | tee $O/${TAG}_working_launches.txt | head -30
