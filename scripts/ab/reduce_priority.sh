#!/bin/bash
# bench step with the reduce streams at low (1, default) / normal (0) priority
for m in 1 0 1 0; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config5 --no-prover-shaped --no-sweeps --reduce-priority $m | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']
print('reduce_priority=$m', 'ms_per_step', d['ms_per_step'], 'value', d['value'], e['msm_phase_ms'], 'ntt', e['ntt_ms'])"
done
