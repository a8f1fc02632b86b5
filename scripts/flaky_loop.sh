# usage (GPU box): bash scripts/flaky_loop.sh N "<pytest -k expression>" [file] -- runs the selection N times, keeps the output of the failing runs
N=${1:-20}; K=${2:-sharded_pipeline}; F=${3:-tests/test_gpu_multirank.py}
mkdir -p gpurun_out/r05; out=gpurun_out/r05/flaky_loop.txt; : > $out
fail=0
for i in $(seq $N); do
  timeout 300 python -m pytest $F -x -q -k "$K" > /tmp/flaky_$i.log 2>&1 || { fail=$((fail+1)); echo "=== run $i" >> $out; grep -v "Gloo\|socket.cpp\|amdgpu.ids" /tmp/flaky_$i.log | grep -E "^E  |^tests/|Error|error|FAILED" | head -30 | cut -c1-7000 >> $out; }
done
echo "$fail of $N runs failed" | tee -a $out
