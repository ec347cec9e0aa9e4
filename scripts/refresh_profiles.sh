#!/bin/bash
# Regenerates the rocprofv3 summaries committed under profiles/ (run on the GPU box through gpurun; writes gpurun_out/).
#   gpurun --timeout 900 -- 'bash scripts/refresh_profiles.sh v3'
# Pass 1: --kernel-trace of the default bench command.  Passes 2-4: PMC counters, one group per run, never combined with
# other trace domains (MI355X_MICROARCH.md HBM section; gpurun refuses combined runs).
set -u
TAG=${1:-v1}
ROUND=${2:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 10 --warmup 2 --blocks 1 --no-cpu-baseline --no-config5 --no-prover-shaped --no-sweeps"
rm -rf /tmp/prof_kt && rocprofv3 --kernel-trace -d /tmp/prof_kt -o bench -- $BENCH > /tmp/bench_kt.log 2>&1
{
  echo "# rocprofv3 --kernel-trace -- python bench.py --steps 10 --warmup 2 --blocks 1 --no-cpu-baseline --no-config5 --no-prover-shaped --no-sweeps   (MI355X, $ROUND, build $TAG)"
  echo "# durations include overlap: the MSM reduce kernels (k_combine_lanes .. k_final_sum) run on an auxiliary stream beside the next step"
  echo "# bench line of this run:"
  grep "^{\"metric\"" /tmp/bench_kt.log | tail -1
  echo
  python $ROOT/scripts/rocpd_summary.py kernels $(find /tmp/prof_kt -name "*_results.db" | head -1)
} > $OUT/${ROUND}_kernel_stats_$TAG.txt
PMCBENCH="python $ROOT/bench.py --steps 3 --warmup 1 --blocks 1 --no-cpu-baseline --no-config5 --no-prover-shaped --no-sweeps"
DBS=""
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY"; do
  d=/tmp/prof_pmc_$(echo $grp | cut -d' ' -f1)
  rm -rf $d && rocprofv3 --pmc $grp -d $d -o bench -- $PMCBENCH > $d.log 2>&1
  DBS="$DBS $(find $d -name '*_results.db' | head -1)"
done
{
  echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (separate passes) -- python bench.py --steps 3 --warmup 1 --blocks 1 --no-cpu-baseline --no-config5 --no-prover-shaped --no-sweeps (build $TAG)"
  echo "# FETCH_SIZE/WRITE_SIZE in KiB per dispatch as reported (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced streams 2x on gfx950; uncalibrated for 64-B gathers)"
  python $ROOT/scripts/rocpd_summary.py pmc $DBS
} > $OUT/${ROUND}_pmc_$TAG.txt
# the resident prover rounds (extra.prover_shaped): kernel trace of the whole default bench command minus config 5
rm -rf /tmp/prof_ps && rocprofv3 --kernel-trace -d /tmp/prof_ps -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --blocks 1 --no-cpu-baseline --no-config5 --no-sweeps > /tmp/bench_ps.log 2>&1
{
  echo "# rocprofv3 --kernel-trace -- python bench.py --steps 2 --warmup 1 --blocks 1 --no-cpu-baseline --no-config5   (MI355X, $ROUND, build $TAG)"
  echo "# dominated by extra.prover_shaped: 6 passes of the TurboPLONK prover sequence at n = 2^20 on the resident prover rounds (bbg_prover_*)"
  grep "^{\"metric\"" /tmp/bench_ps.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('# prover_shaped:', json.dumps(d['extra']['prover_shaped']))"
  echo
  python $ROOT/scripts/rocpd_summary.py kernels $(find /tmp/prof_ps -name "*_results.db" | head -1)
} > $OUT/${ROUND}_prover_kernel_stats_$TAG.txt
head -25 $OUT/${ROUND}_kernel_stats_$TAG.txt | cut -c1-200
grep -c . $OUT/${ROUND}_pmc_$TAG.txt
