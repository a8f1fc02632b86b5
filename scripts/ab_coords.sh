#!/bin/bash
# sharing keyed on the LUT coordinates before interpolation (TSX_DEDUP_COORDS): parity subset, then config 4 and the coefficient set-up time
python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_reference_cases.py tests/test_gpu_fullsize.py -m gpu -q -x -k "not 8_16-256" 2>&1 | grep -E "^E|passed|failed" | head
for dc in 0 1; do
  TSX_DEDUP_COORDS=$dc python bench_specint.py --streams 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']['calls']; print('coords=$dc one instance', round(d['value'],2), 'g-points/s; cold', round(c[0]['gpoints_per_s'],2), 'balance', c[-1]['energy_balance_max'], 'toa', c[-1]['toa_net_down_Wm2'], 'its', c[-1]['iterations_min_med_max'])"
  TSX_DEDUP_COORDS=$dc python bench.py --no-cpu-baseline --skip-no-sharing --skip-extra-legs --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('   bench coords=$dc', round(d['value']/1e6,1), 'M its', c['iterations'], 'coeff_setup_ms', round(c['coeff_setup_ms'],2), 'distinct', c['coeff_dedup']['distinct_blocks'])"
done
