#!/bin/bash
# one period of the bench step's kernel timeline (start relative to a k_sortA_count, duration, queue, kernel)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktl && rocprofv3 --kernel-trace -d /tmp/ktl -o bench -- python $ROOT/bench.py --steps 10 --warmup 2 --blocks 1 --no-cpu-baseline --no-config5 --no-prover-shaped --no-sweeps > /tmp/ktl.log 2>&1
python $ROOT/scripts/rocpd_timeline.py $(find /tmp/ktl -name "*_results.db" | head -1) k_sortA_count 8 26
