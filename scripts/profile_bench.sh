# usage (on the GPU box): bash scripts/profile_bench.sh [tag [bench.py arguments ...]]
# default bench line, rocprofv3 kernel trace, and the two PMC passes of the same command -> gpurun_out/r02/<tag>_*
set -x
TAG=${1:-bench}
shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 "$@" > $O/${TAG}_line.json 2> $O/${TAG}_err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 "$@" > $O/${TAG}_kt_line.json 2> $O/${TAG}_kt_err.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --kernel-reps 4 "$@" > /dev/null 2> $O/${TAG}_fetch_err.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --kernel-reps 4 "$@" > /dev/null 2> $O/${TAG}_write_err.log
find $O -name "*.csv" | head -40
