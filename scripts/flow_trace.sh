# usage: in the container  bash scripts/flow_trace.sh build   (hipcc, ~1 min);  on the GPU box  bash scripts/flow_trace.sh [xm ym]
# Where a work item of the flow kernel (tsx_k_pcs_flow) spends its time: a build with -DTSX_FLOW_TRACE leaves wall-clock stamps
# (100 MHz) per item -- ticket known, neighbours' progress words seen, barrier, phase-1 loads + local scan done, first / second
# scan barrier, body done, stores drained, barrier, published -- read back after ONE application of M^-1.
cd ${GRAFT_REPO_ROOT:-/root/repo}
if [ "$1" = build ]; then
  cd tenstream_amd/csrc
  mkdir -p ../lib_trace
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -DTSX_FLOW_TRACE -c -o ../lib_trace/tsx_pcs_flow.o tsx_pcs_flow.hip &&
    /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o ../lib_trace/libtsx.so ../lib_trace/tsx_pcs_flow.o $(ls ../lib/obj/*.o | grep -v tsx_pcs_flow.o) -ldl && rm ../lib_trace/tsx_pcs_flow.o
  exit $?
fi
TSX_LIB=$PWD/tenstream_amd/lib_trace/libtsx.so python3 scripts/flow_trace.py "$@" 2>&1 | grep -v amdgpu
