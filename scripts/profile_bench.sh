set -x
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/r02/bench_line.json 2> $R/gpurun_out/r02/bench_err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02/kt -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 > $R/gpurun_out/r02/kt_line.json 2> $R/gpurun_out/r02/kt_err.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r02/fetch -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --kernel-reps 4 > /dev/null 2> $R/gpurun_out/r02/fetch_err.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r02/write -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --kernel-reps 4 > /dev/null 2> $R/gpurun_out/r02/write_err.log
ls -R $R/gpurun_out/r02 | head -40
