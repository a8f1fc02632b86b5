# usage (GPU box): bash scripts/pass_pmc.sh   -- what an intermediate pass of the 3_10 scan preconditioner waits for:
# SQ / TA / TCP / TCC counters of tsx_k_pcs_rb<..., MODE 0, RQ 2> on the metric domain (scripts/pcsbench.py, the pass timed alone
# 50 x), one rocprofv3 --pmc run per counter set (no tracing options beside it).
cd /tmp; export TMPDIR=/tmp
export CFGS="4,16,32"
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/pass_pmc
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "VALUBusy MemUnitBusy MemUnitStalled L2CacheHit" "GRBM_GUI_ACTIVE GRBM_COUNT TCP_TA_TCP_STATE_READ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1)); rm -rf /tmp/pp$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pp$i -- python3 $GRAFT_REPO_ROOT/scripts/pcsbench.py > /tmp/pp$i.log 2>&1 || { echo "set $i failed: $set"; tail -3 /tmp/pp$i.log; continue; }
  f=$(ls /tmp/pp$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -z "$f" ] && { echo "set $i: no csv ($set)"; tail -3 /tmp/pp$i.log; continue; }
  python3 - "$f" <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(r"tsx_k_pcs_rb<4, 16, 32, true, 0, true, 2", r["Kernel_Name"]):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    v.sort(); big = [x for x in v if x >= 0.5 * v[-1]] or v
    print(f"{k:34s} launches {len(big):4d}  mean {sum(big) / len(big):16.1f}")
PY
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/pass_pmc/summary.txt
