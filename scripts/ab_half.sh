#!/bin/bash
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "half_step or stop_rule or reason or retry or accept_incomplete or warm" 2>&1 | tail -5
for he in 0 1; do
  TSX_HALF_EXIT=$he python bench.py --no-cpu-baseline --skip-no-sharing --skip-extra-legs --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('half_exit=$he', round(d['value']/1e6,1), 'Mcells/s', round(d['ms_per_step'],2), 'ms its', c['iterations'], 'rel', c['rel_residual'], 'tight', c['tight_run'], 'warm', c['warm_start'])"
  for extra in "--nx 128 --ny 128" "--solver 8_16" "--field heterogeneous"; do
  TSX_HALF_EXIT=$he python bench.py $extra --no-cpu-baseline --skip-no-sharing --skip-extra-legs --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('   $extra half_exit=$he', round(d['value']/1e6,1), 'Mcells/s', round(d['ms_per_step'],2), 'ms its', c['iterations'], 'rel', c['rel_residual'])"
  done
done
