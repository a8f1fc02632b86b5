# usage (GPU box): bash scripts/flaky_loop.sh N "<pytest -k expression>" [file] [outfile] -- runs the selection N times (or until $FLAKY_SECONDS have
# passed), keeps the output of the failing runs
N=${1:-20}; K=${2:-sharded_pipeline}; F=${3:-tests/test_gpu_multirank.py}
out=${4:-gpurun_out/r06/flaky_loop.txt}; mkdir -p $(dirname $out); : > $out
fail=0; ran=0; t0=$(date +%s)
for i in $(seq $N); do
  [ -n "$FLAKY_SECONDS" ] && [ $(( $(date +%s) - t0 )) -ge "$FLAKY_SECONDS" ] && break
  ran=$((ran+1))
  timeout 300 python -m pytest $F -x -q -k "$K" > /tmp/flaky_$i.log 2>&1 || { fail=$((fail+1)); echo "=== run $i" >> $out; grep -v "Gloo\|socket.cpp\|amdgpu.ids" /tmp/flaky_$i.log | grep -E "^E  |^tests/|Error|error|FAILED" | head -30 | cut -c1-7000 >> $out; }
done
echo "$fail of $ran runs failed ($(( $(date +%s) - t0 )) s; -k \"$K\"; TSX_POOL=${TSX_POOL:-1})" | tee -a $out
