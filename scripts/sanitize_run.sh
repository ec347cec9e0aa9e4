#!/bin/bash
# Runs the sanitizer configuration built by scripts/sanitize_build.sh on the GPU box and writes gpurun_out/r06_sanitizers.txt:
#   gpurun --timeout 1500 -- 'bash scripts/sanitize_run.sh'
# protect_shadow_gap=0: the HIP runtime maps device memory into the range ASan's shadow gap covers.  Leak checking is on for the C-ABI driver
# (suppressing the runtime's own exit-time allocations), off under Python.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_sanitizers.txt
S=$ROOT/build_san
mkdir -p $ROOT/gpurun_out
cat > /tmp/lsan.supp <<EOS
leak:libamdhip64
leak:libhsa-runtime64
leak:libamd_comgr
leak:librocprofiler
leak:libhiprtc
EOS
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0
{
  echo "# ASan + UBSan over the host side (scripts/sanitize_build.sh / sanitize_run.sh), $(date -u +%F), one MI355X"
  echo "## (1) san_capi_driver: libbbg_san.so (clang ASan+UBSan on the host pass of every csrc/*.hip) -- SRS refcounts, arena regrowth, batches, prover key replacement, memory trim, 4-context device group"
  ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=1 LSAN_OPTIONS=suppressions=/tmp/lsan.supp:print_suppressions=0 $S/san_capi_driver 2>&1 | tee /tmp/san1.log | tail -40
  echo "exit status: ${PIPESTATUS[0]}"
  echo "## (2) shim_check_san 12: g++ ASan+UBSan over shim/bbg_barretenberg_shim.cpp + shim_check.cpp (table cache, transient tables, wrap vs __real_)"
  ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0 $S/shim_check_san 12 2>&1 | tee /tmp/san2.log | tail -25
  echo "exit status: ${PIPESTATUS[0]}"
  echo "## (3) the same with the 4-context device group behind pippenger_unsafe (BBG_SHIM_DEVICES=0,0,0,0)"
  ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0 BBG_SHIM_DEVICES=0,0,0,0 BBG_SHIM_MULTI_MIN_POINTS=1000 $S/shim_check_san 12 2>&1 | tee /tmp/san3.log | tail -12
  echo "exit status: ${PIPESTATUS[0]}"
  echo "## (4) the wrapped construct_proof() build with shim TUs + driver instrumented (libbbprover_wrap_san.so), under LD_PRELOAD=libasan: key cache, replay, all five prover types at 2^9; r5: the seven wrapped rounds, device-error fall-backs, key re-upload"
  ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) BBG_PROVER_WRAP_SO=$S/libbbprover_wrap_san.so \
    python $ROOT/tests/tools/san_wrap_check.py 2>&1 | tee /tmp/san4.log | tail -25
  echo "exit status: ${PIPESTATUS[0]}"
  echo "## findings in the complete outputs of the four legs (UBSan does not stop at a finding; ASan does)"
  for k in 1 2 3 4; do
    echo "leg $k: $(grep -c 'runtime error:' /tmp/san$k.log) UBSan reports, $(grep -c 'ERROR: AddressSanitizer' /tmp/san$k.log) ASan reports, $(grep -c 'ERROR: LeakSanitizer' /tmp/san$k.log) LeakSanitizer reports"
    grep 'runtime error:' /tmp/san$k.log | sort | uniq -c | head -10
  done
} > $OUT 2>&1
cat $OUT
