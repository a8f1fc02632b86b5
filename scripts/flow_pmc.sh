# usage (GPU box): bash scripts/flow_pmc.sh [bench.py size arguments]   -- SQ / TA / TCP / TCC counters of the flow kernel (tsx_k_pcs_flow),
# one rocprofv3 --pmc run per counter set (no tracing options beside it), whole-device sums per working launch
cd /tmp; export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r05
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" \
           "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "VALUBusy MemUnitBusy MemUnitStalled L2CacheHit" "GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_SALU SQ_INST_CYCLES_VMEM"; do
  i=$((i+1)); rm -rf /tmp/fp$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/fp$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --skip-no-sharing --skip-extra-legs --steps 2 --warmup 1 --kernel-reps 4 "$@" > /tmp/fp$i.log 2>&1 || { echo "set $i failed: $set"; tail -3 /tmp/fp$i.log; continue; }
  f=$(ls /tmp/fp$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -z "$f" ] && { echo "set $i: no csv ($set)"; tail -3 /tmp/fp$i.log; continue; }
  python3 - "$f" <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(list)
name = None
for r in csv.DictReader(open(sys.argv[1])):
    if "tsx_k_pcs_flow" in r["Kernel_Name"]:
        name = re.search(r"tsx_k_pcs_flow<[^>]*>", r["Kernel_Name"]).group(0)
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    v.sort(); big = [x for x in v if x >= 0.5 * v[-1]] or v
    print(f"{name}  {k:34s} launches {len(big):4d}  mean {sum(big) / len(big):16.1f}")
PY
done 2>&1
