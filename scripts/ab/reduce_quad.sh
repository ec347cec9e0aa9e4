#!/bin/bash
# bench step for several msm_reduce_quad stage masks (bit 0 combine, 1 row/column sums, 2 bit planes, 3 plane sum)
for m in 14 12 10 8 6 0 15 14; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config5 --no-prover-shaped --no-sweeps --reduce-quad $m | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']
print('quad=$m', 'ms_per_step', d['ms_per_step'], 'value', d['value'], e['msm_phase_ms'], 'ntt', e['ntt_ms'])"
done
