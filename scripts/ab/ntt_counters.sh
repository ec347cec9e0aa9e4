#!/bin/bash
# scripts/ab/ntt_counters.sh <tag> [<tag> ...]: SQ wave / LDS counters of the NTT pass kernels of A/B builds build_ab/libbbg_<tag>.so
# (separate --pmc passes, --kernel-trace only; rocprofv3 output condensed by scripts/rocpd_summary.py).  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for tag in "$@"; do
  echo "## build $tag : python tests/tools/r05_ntt_time.py 20 (k_ntt_pass29<10>: column pass ROW=false, row pass ROW=true)"
  DBS=""
  for grp in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
    d=/tmp/ntt_pmc_${tag}_$(echo $grp | cut -d' ' -f1)
    rm -rf $d
    BBG_LIB_PATH=$ROOT/build_ab/libbbg_$tag.so rocprofv3 --pmc $grp --kernel-trace -d $d -o ntt -- python $ROOT/tests/tools/r05_ntt_time.py 20 > $d.log 2>&1
    DBS="$DBS $(find $d -name '*_results.db' | head -1)"
  done
  python $ROOT/scripts/rocpd_summary.py pmc $DBS | grep -E "kernel|ntt_pass"
done
