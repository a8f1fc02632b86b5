# usage (GPU box): bash scripts/fold_probe.sh   -- what bounds an intermediate pass of the 3_10 scan preconditioner?
# Three builds of libtsx side by side (the library as shipped; the pass's per-cell streams wrapped onto 2^16 cells = served by L2;
# the record index wrapped too = table gathers served by L1), the pass timed alone on the metric domain (scripts/pcsbench.py).
# The wrapped builds compute garbage: only `pass_ms` of their lines means anything.
set -e
cd $GRAFT_REPO_ROOT/tenstream_amd/csrc
for v in fold foldidx; do
  [ -f ../lib_$v/libtsx.so ] && continue   # built in the container and shipped with the snapshot
  mkdir -p ../lib_$v
  extra="-DTSX_PCS_FOLD=65536"; [ $v = foldidx ] && extra="$extra -DTSX_PCS_FOLD_IDX"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on $extra -c -o ../lib_$v/tsx_pcs.o tsx_pcs.hip &
done
wait
for v in fold foldidx; do
  [ -f ../lib_$v/libtsx.so ] && continue
  /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o ../lib_$v/libtsx.so ../lib_$v/tsx_pcs.o $(ls ../lib/obj/*.o | grep -v tsx_pcs.o) -ldl
done
cd $GRAFT_REPO_ROOT
for v in "" _fold _foldidx; do
  echo "== lib$v"
  TSX_PROBE_LIB=tenstream_amd/lib$v/libtsx.so CFGS="4,16,32" python3 - <<'PY' 2>&1 | grep -v amdgpu
import os, runpy, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import tenstream_amd._lib as L
L.LIB_PATH = os.path.join(os.environ["GRAFT_REPO_ROOT"], os.environ["TSX_PROBE_LIB"])
sys.argv = ["pcsbench.py"]
try:
    runpy.run_path(os.path.join(os.environ["GRAFT_REPO_ROOT"], "scripts", "pcsbench.py"), run_name="__main__")
except Exception as e:   # the wrapped builds may fail the solve: the pass timing is printed by then or not at all
    print("failed:", e)
PY
done
