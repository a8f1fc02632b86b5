# usage (GPU box): bash scripts/fresh_alloc_loop.sh N -- N rounds of 4 concurrent fresh processes of tests/c/bin/fresh_alloc_probe
N=${1:-200}; mkdir -p gpurun_out/r06; out=gpurun_out/r06/fresh_alloc_loop.txt; : > $out
[ -x tests/c/bin/fresh_alloc_probe ] || { mkdir -p tests/c/bin; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tests/c/bin/fresh_alloc_probe tests/c/fresh_alloc_probe.hip; }
bad=0; t0=$(date +%s)
for i in $(seq $N); do
  for r in 0 1 2 3; do ( tests/c/bin/fresh_alloc_probe 8 > /tmp/fap_$r.log 2>&1; echo $? > /tmp/fap_$r.rc ) & done
  wait
  for r in 0 1 2 3; do rc=$(cat /tmp/fap_$r.rc); if [ "$rc" != "0" ]; then bad=$((bad+1)); echo "=== round $i proc $r rc $rc" >> $out; head -20 /tmp/fap_$r.log >> $out; fi; done
done
echo "fresh_alloc_loop: $bad of $((N*4)) fresh processes lost a store ($(( $(date +%s) - t0 )) s)" | tee -a $out
