#!/bin/bash
# A/B of the accumulation's limb format: option msm_limbs29 (bench.py --limbs29 0|1)
for v in 1 0 1 0; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config5 --no-prover-shaped --no-sweeps --limbs29 $v | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']
print('limbs29=$v', 'ms_per_step', d['ms_per_step'], 'value', d['value'], e['msm_phase_ms'], 'ntt', e['ntt_ms'], 'acc launch', d['roofline']['avg_launch_ms'])"
done
