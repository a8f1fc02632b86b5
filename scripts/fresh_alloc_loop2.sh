# usage (GPU box): bash scripts/fresh_alloc_loop2.sh N [procs] [watch_ms] [exit_mb] -- N rounds of `procs` concurrent fresh processes of
# tests/c/bin/fresh_alloc_probe2, started in two staggered halves so that one half exits (its memory is released) while the other allocates
N=${1:-100}; P=${2:-8}; W=${3:-20}; X=${4:-2048}
mkdir -p gpurun_out/r06; out=gpurun_out/r06/fresh_alloc_loop2.txt; : > $out
[ -x tests/c/bin/fresh_alloc_probe2 ] || { mkdir -p tests/c/bin; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tests/c/bin/fresh_alloc_probe2 tests/c/fresh_alloc_probe2.hip; }
bad=0; t0=$(date +%s)
for i in $(seq $N); do
  for r in $(seq $P); do
    ( [ $((r % 2)) = 0 ] && sleep 0.15; tests/c/bin/fresh_alloc_probe2 $W $X > /tmp/fap2_$r.log 2>&1; echo $? > /tmp/fap2_$r.rc ) &
  done
  wait
  for r in $(seq $P); do rc=$(cat /tmp/fap2_$r.rc); if [ "$rc" != "0" ]; then bad=$((bad+1)); echo "=== round $i proc $r rc $rc" >> $out; head -8 /tmp/fap2_$r.log >> $out; fi; done
done
echo "fresh_alloc_loop2: $bad of $((N*P)) fresh processes saw fresh memory lose its contents ($(( $(date +%s) - t0 )) s; $P processes per round, watch $W ms, $X MB released at exit)" | tee -a $out
