#!/bin/bash
# The soak record committed as profiles/r06_soak.txt (run on the GPU box through gpurun; writes gpurun_out/r06_soak.txt):
#   gpurun --timeout 1500 -- 'bash scripts/soak_run.sh'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_soak.txt
mkdir -p $ROOT/gpurun_out
SHA=$(sha256sum $ROOT/aztec-2.0_amd/csrc/libbbg.so | cut -d' ' -f1)
{
  echo "# soak of the shipped build (libbbg.so sha256 $SHA), one MI355X, $(date -u +%FT%TZ)"
  echo "## tests/tools/soak_msm.py 6000   (random sizes / offsets / window widths incl. the small path (8) / sort paths / scalar mixes incl. all-equal and P, -P pairs, against the oracle)"
  python $ROOT/tests/tools/soak_msm.py 6000 2>&1 | tail -1
  echo "## tests/tools/soak_ntt.py 6000   (random sizes up to 2^17 / ops / generator sizes / constants / pass plans (ntt_max_logr8 6 .. 11) / pass kernels (32-bit two-plane, one-plane, 29-bit with the constant-operand product), coset_fft_extend, against the oracle)"
  python $ROOT/tests/tools/soak_ntt.py 6000 2>&1 | tail -1
  echo "## tests/tools/soak_resident.py 1200   (resident proofs, five prover types, 2^9 .. 2^14 gates, sessions created and destroyed: every proof through the reference verifier; the division by Z*_H inside the coset iFFT's load, 29-bit linear combinations / evaluations)"
  python $ROOT/tests/tools/soak_resident.py 1200 2>&1 | tail -1
  echo "## tests/tools/soak_msm_batch.py 1200"
  python $ROOT/tests/tools/soak_msm_batch.py 1200 2>&1 | tail -1
} > $OUT
cat $OUT
