#!/bin/bash
# A/B of the per-block record layout of the scan passes on the all-distinct field (TSX_PC_ENTRY_MAJOR), one box
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -m gpu -q -x -k "shared_block or distinct or near" 2>&1 | tail -4
for em in 0 1; do
  TSX_PC_ENTRY_MAJOR=$em python bench.py --field heterogeneous --no-cpu-baseline --skip-no-sharing --skip-extra-legs --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('entry_major=$em', round(d['value']/1e6,1), 'Mcells/s', round(d['ms_per_step'],2), 'ms its', c['iterations'], 'pass_ms', d['roofline']['ms_per_launch'], 'iter_ms', c['iter_ms'], 'copy', c['copy_GBps_measured'], 'read', c['read_GBps_measured'], c['bandwidth_probe']['copy_variant'], c['bandwidth_probe']['read_variant'])"
done
