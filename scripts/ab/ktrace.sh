#!/bin/bash
# kernel trace + VALU counters of the default bench step: bash scripts/ab/ktrace.sh [extra bench flags]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 10 --warmup 2 --blocks 1 --no-cpu-baseline --no-config5 --no-prover-shaped --no-sweeps $*"
rm -rf /tmp/kt && rocprofv3 --kernel-trace -d /tmp/kt -o bench -- $BENCH > /tmp/kt.log 2>&1
python $ROOT/scripts/rocpd_summary.py kernels $(find /tmp/kt -name "*_results.db" | head -1) | head -24 | cut -c1-180
rm -rf /tmp/pm && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY -d /tmp/pm -o bench -- $BENCH > /tmp/pm.log 2>&1
python $ROOT/scripts/rocpd_summary.py pmc $(find /tmp/pm -name "*_results.db" | head -1) | head -30 | cut -c1-200
