#!/bin/bash
# bench step for several msm_acc_waves (lane segments per SIMD lane of the accumulation)
for m in 0 3 5 6 8 0; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config5 --no-prover-shaped --no-sweeps --acc-waves $m | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']
print('acc_waves=$m', 'ms_per_step', d['ms_per_step'], 'value', d['value'], e['msm_phase_ms'], 'ntt', e['ntt_ms'])"
done
