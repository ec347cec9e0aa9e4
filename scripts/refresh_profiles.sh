#!/bin/bash
# Regenerates the rocprofv3 summaries committed under profiles/ (run on the GPU box through gpurun; writes gpurun_out/).
#   gpurun --timeout 1500 -- 'bash scripts/refresh_profiles.sh v1 r04 <git describe>'
# Every file is STAMPED with the sha256 of the libbbg.so that ran (and the source revision passed as $3): bench.py compares the stamp with
# the library it loads and reports "profile_matches_build".
#   pass 0  CONTROL: the same bench command un-profiled, in the same session (the driver-style numbers the profiled passes are held against)
#   pass 1  --kernel-trace of the bench command (reduce phase on the auxiliary streams: durations include overlap)
#   pass 1b --kernel-trace with msm_async_reduce = 0 (BBG_BENCH_INLINE_REDUCE=1): every kernel alone on the device -> non-overlapped durations
#   pass 2-4 PMC counters, one group per run, never combined with other trace domains (MI355X_MICROARCH.md HBM section; gpurun refuses combined runs)
#   pass 5  GRBM_GUI_ACTIVE per kernel (busy cycles / duration = the effective clock under the profiler)
#   pass 6  kernel trace + PMC of config 5's single-GPU legs: ONE 2^24 MSM and ONE 2^24 coset NTT (tests/tools/r04_config5_legs.py)
#   pass 7  kernel trace of the resident prover rounds (extra.prover_shaped)
set -u
TAG=${1:-v1}
ROUND=${2:-r06}
REV=${3:-unknown}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
LIBSHA=$(sha256sum $ROOT/aztec-2.0_amd/csrc/libbbg.so | cut -d' ' -f1)
stamp() { echo "# build: libbbg.so sha256 $LIBSHA  source $REV  ($ROUND $TAG, $(date -u +%FT%TZ), one MI355X)"; }
cd /tmp && export TMPDIR=/tmp
# The DEFAULT timed region (20 steps x 5 blocks after 3 warm-up steps), without the extras.  A short run measures a device that has not
# settled: 12 steps in all gave 1.667 ms per step profiled AND un-profiled (round 4, v1) against 1.488 for the default region in the same
# session -- the "16 % slower under the profiler" of round 3 was the length of the profiled command, not the profiler.
BENCH="python $ROOT/bench.py --steps 20 --warmup 3 --blocks 5 --no-cpu-baseline --no-config5 --no-prover-shaped --no-sweeps"
# ---- pass 0: control
$BENCH > /tmp/bench_ctl.log 2>&1
python $ROOT/bench.py --steps 10 --warmup 2 --blocks 1 --no-cpu-baseline --no-config5 --no-prover-shaped --no-sweeps > /tmp/bench_ctl20.log 2>&1
# ---- pass 1 / 1b: kernel traces
rm -rf /tmp/prof_kt && rocprofv3 --kernel-trace -d /tmp/prof_kt -o bench -- $BENCH > /tmp/bench_kt.log 2>&1
rm -rf /tmp/prof_kti && BBG_BENCH_INLINE_REDUCE=1 rocprofv3 --kernel-trace -d /tmp/prof_kti -o bench -- $BENCH > /tmp/bench_kti.log 2>&1
BBG_BENCH_INLINE_REDUCE=1 $BENCH > /tmp/bench_ctli.log 2>&1
line() { grep "^{\"metric\"" $1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); e=d['extra']; print('ms_per_step', d['ms_per_step'], ' value', d['value'], ' accumulate_avg_ms', d['roofline']['avg_launch_ms'], ' phases', e['msm_phase_ms'], ' ntt_ms', e['ntt_ms'])"; }
{
  stamp
  echo "# rocprofv3 --kernel-trace -- python bench.py --steps 20 --warmup 3 --blocks 5 --no-cpu-baseline --no-config5 --no-prover-shaped --no-sweeps"
  echo "# CONTROL, same session, NOT profiled (same command):         $(line /tmp/bench_ctl.log)"
  echo "# CONTROL, same session, NOT profiled (10 steps x 1 block, the round-3 profile command: a device that has not settled):  $(line /tmp/bench_ctl20.log)"
  echo "# the profiled run's own line:                                $(line /tmp/bench_kt.log)"
  echo "# durations below include overlap: the MSM reduce kernels (k_combine .. k_final_sum) run on an auxiliary stream beside the next step"
  echo
  python $ROOT/scripts/rocpd_summary.py kernels $(find /tmp/prof_kt -name "*_results.db" | head -1)
  echo
  echo "# ---- the same with the reduce phase IN LINE on the main stream (msm_async_reduce = 0): one kernel at a time, non-overlapped durations"
  echo "# CONTROL (in-line reduce), NOT profiled: $(line /tmp/bench_ctli.log)"
  echo "# profiled (in-line reduce):              $(line /tmp/bench_kti.log)"
  echo
  python $ROOT/scripts/rocpd_summary.py kernels $(find /tmp/prof_kti -name "*_results.db" | head -1)
} > $OUT/${ROUND}_kernel_stats_$TAG.txt
# ---- passes 2-5: counters
PMCBENCH="python $ROOT/bench.py --steps 3 --warmup 1 --blocks 1 --no-cpu-baseline --no-config5 --no-prover-shaped --no-sweeps"
DBS=""
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE"; do
  d=/tmp/prof_pmc_$(echo $grp | cut -d' ' -f1)
  KT=""; if [ "$grp" = "GRBM_GUI_ACTIVE" ]; then KT="--kernel-trace"; fi   # durations of the same dispatches for the clock estimate
  rm -rf $d && rocprofv3 --pmc $grp $KT -d $d -o bench -- $PMCBENCH > $d.log 2>&1
  DBS="$DBS $(find $d -name '*_results.db' | head -1)"
done
{
  stamp
  echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* / GRBM_GUI_ACTIVE (separate passes) -- python bench.py --steps 3 --warmup 1 --blocks 1 --no-cpu-baseline --no-config5 --no-prover-shaped --no-sweeps"
  echo "# FETCH_SIZE/WRITE_SIZE in KiB per dispatch as reported (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced streams 2x on gfx950; uncalibrated for 64-B gathers)"
  echo "# GRBM_GUI_ACTIVE = GPU busy cycles during the dispatch: / the kernel's duration in the GRBM pass's own trace = the effective clock under the profiler"
  python $ROOT/scripts/rocpd_summary.py pmc $DBS
  echo
  echo "# effective clock per kernel: GRBM_GUI_ACTIVE / duration of the same dispatches.  The counter is summed over the 8 XCDs: GHz column / 8 = clock of one XCD"
  echo "# (k_accumulate29: 16.3 / 8 = 2.03 GHz against 2.3 for the light kernels -- the mad-bound kernel runs power-limited below the 2.4 GHz the issue peaks are priced at)"
  python $ROOT/scripts/rocpd_summary.py clocks $(find /tmp/prof_pmc_GRBM_GUI_ACTIVE -name '*_results.db' | head -1)
} > $OUT/${ROUND}_pmc_$TAG.txt
# ---- pass 6: config 5's single-GPU legs at 2^24
LEGS="python $ROOT/tests/tools/r04_config5_legs.py"
rm -rf /tmp/prof_c5 && rocprofv3 --kernel-trace -d /tmp/prof_c5 -o c5 -- $LEGS > /tmp/c5_kt.log 2>&1
C5DBS=""
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  d=/tmp/prof_c5_$(echo $grp | cut -d' ' -f1)
  rm -rf $d && rocprofv3 --pmc $grp -d $d -o c5 -- $LEGS > $d.log 2>&1
  C5DBS="$C5DBS $(find $d -name '*_results.db' | head -1)"
done
$LEGS > /tmp/c5_ctl.log 2>&1
{
  stamp
  echo "# config 5's single-GPU legs: 3 x (one 2^24-point MSM, C = 22: 12.9 GiB of window tables) + 3 x (one 2^24 coset NTT) -- tests/tools/r04_config5_legs.py"
  echo "# CONTROL, not profiled: $(grep '^legs' /tmp/c5_ctl.log | tail -1)"
  echo "# profiled:              $(grep '^legs' /tmp/c5_kt.log | tail -1)"
  echo
  python $ROOT/scripts/rocpd_summary.py kernels $(find /tmp/prof_c5 -name "*_results.db" | head -1)
  echo
  python $ROOT/scripts/rocpd_summary.py pmc $C5DBS
} > $OUT/${ROUND}_config5_legs_$TAG.txt
# ---- pass 7: the resident prover rounds
rm -rf /tmp/prof_ps && rocprofv3 --kernel-trace -d /tmp/prof_ps -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --blocks 1 --no-cpu-baseline --no-config5 --no-sweeps > /tmp/bench_ps.log 2>&1
{
  stamp
  echo "# rocprofv3 --kernel-trace -- python bench.py --steps 2 --warmup 1 --blocks 1 --no-cpu-baseline --no-config5"
  echo "# dominated by extra.prover_shaped: 6 passes of the TurboPLONK prover sequence at n = 2^20 on the resident prover rounds (bbg_prover_*)"
  grep "^{\"metric\"" /tmp/bench_ps.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('# prover_shaped:', json.dumps(d['extra']['prover_shaped']))"
  echo
  python $ROOT/scripts/rocpd_summary.py kernels $(find /tmp/prof_ps -name "*_results.db" | head -1)
} > $OUT/${ROUND}_prover_kernel_stats_$TAG.txt
head -30 $OUT/${ROUND}_kernel_stats_$TAG.txt | cut -c1-220
grep -c . $OUT/${ROUND}_pmc_$TAG.txt $OUT/${ROUND}_config5_legs_$TAG.txt
