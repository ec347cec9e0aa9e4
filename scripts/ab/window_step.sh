#!/bin/bash
# bench step at n = 2^20 for each compiled window width
for w in 20 19 17 22 20 19; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config5 --no-prover-shaped --no-sweeps --msm-window $w | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']
print('window=$w', 'ms_per_step', d['ms_per_step'], 'value', d['value'], e['msm_phase_ms'], 'ntt', e['ntt_ms'])"
done
