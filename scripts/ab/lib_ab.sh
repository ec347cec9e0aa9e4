#!/bin/bash
# scripts/ab/lib_ab.sh <libA.so> <libB.so> [rounds]: the default bench step with two builds of the same ABI, alternating on ONE box
# (BBG_LIB_PATH; boxes of the pool differ by ~5 %, so only same-box pairs are comparable).  Experiments only.
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for lib in $A $B; do
    BBG_LIB_PATH=$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config5 --no-prover-shaped --no-sweeps | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']
print('$(basename $lib)', 'ms_per_step', d['ms_per_step'], 'value', d['value'], e['msm_phase_ms'], 'ntt', e['ntt_ms'], 'acc launch', d['roofline']['avg_launch_ms'])"
  done
done
