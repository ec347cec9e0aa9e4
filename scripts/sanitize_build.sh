#!/bin/bash
# Builds the sanitizer configuration (ASan + UBSan over the HOST code; reference precedent: barretenberg CMakeLists.txt:5-8, MEMORY_CHECKS)
# into build_san/ (git-ignored; travels to the GPU box with gpurun).  Run here (hipcc cross-compiles), then scripts/sanitize_run.sh there.
#   (1) libbbg_san.so      csrc/*.hip with -fsanitize=address,undefined on the host pass (-fno-gpu-sanitize: device code as shipped)
#   (2) san_capi_driver    tests/tools/san_capi_driver.cpp, same compiler and runtime
#   (3) shim_check_san     g++ ASan + UBSan over shim/bbg_barretenberg_shim.cpp + shim_check.cpp (+ the reference's TUs where they lie)
#   (4) libbbprover_wrap_san.so  the wrapped prover build with the two shim TUs and the driver instrumented (g++), for the key-cache test
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build_san
CS=$ROOT/aztec-2.0_amd/csrc
mkdir -p $OUT/obj
HIPCC=/opt/rocm/bin/hipcc
SAN="-fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -g"
OBJS=""
for f in ntt msm msm_tiny msm_w13 msm_w16 msm_w17 msm_w19 msm_w20 msm_w22 poly quotient prover multi bbg_capi; do
  ( $HIPCC --offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wall -Wno-unused-function -I/opt/rocm/include $SAN -c $CS/$f.hip -o $OUT/obj/$f.o ) &
  OBJS="$OBJS $OUT/obj/$f.o"
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC $SAN -o $OUT/libbbg_san.so $OBJS
$HIPCC --offload-arch=gfx950 -O1 -std=c++17 $SAN -o $OUT/san_capi_driver $ROOT/tests/tools/san_capi_driver.cpp -L$OUT -lbbg_san -Wl,-rpath,'$ORIGIN'
REF=${REF:-/root/reference}
B=$REF/barretenberg/src/aztec
if [ -d $B ]; then
  GS="-fsanitize=address,undefined -fno-omit-frame-pointer -g"
  CXXFLAGS="-std=gnu++20 -O1 -march=haswell -madx -fopenmp -fconstexpr-ops-limit=100000000 -Wno-deprecated -DNDEBUG -I$B"
  REFSRC="$B/ecc/curves/bn254/scalar_multiplication/scalar_multiplication.cpp $B/ecc/curves/bn254/scalar_multiplication/runtime_states.cpp \
          $B/ecc/curves/bn254/scalar_multiplication/process_buckets.cpp $B/polynomials/polynomial_arithmetic.cpp $B/polynomials/evaluation_domain.cpp \
          $B/numeric/random/engine.cpp $B/env/logstr.cpp $B/crypto/keccak/keccak.cpp $B/crypto/keccak/keccakf1600.cpp"
  # the reference's own TUs are compiled as shipped (-O3, no instrumentation: their unaligned casts are the reference's business); the shim
  # and the checker are instrumented
  mkdir -p $OUT/obj/ref
  i=0
  REFOBJ=""
  for s in $REFSRC; do
    i=$((i+1))
    ( g++ -std=gnu++20 -O3 -march=haswell -madx -fopenmp -fconstexpr-ops-limit=100000000 -Wno-deprecated -DNDEBUG -I$B -c $s -o $OUT/obj/ref/r$i.o ) &
    REFOBJ="$REFOBJ $OUT/obj/ref/r$i.o"
  done
  wait
  g++ $CXXFLAGS $GS -c $ROOT/shim/bbg_barretenberg_shim.cpp -o $OUT/obj/shim_san.o
  g++ $CXXFLAGS $GS -c $ROOT/shim/shim_check.cpp -o $OUT/obj/shim_check_san.o
  g++ $GS -fopenmp -o $OUT/shim_check_san $OUT/obj/shim_check_san.o $OUT/obj/shim_san.o $REFOBJ $(cat $ROOT/shim/wrap_flags.txt) \
      -L$CS -lbbg -Wl,-rpath,'$ORIGIN/../aztec-2.0_amd/csrc' -Wl,-rpath-link,/opt/rocm/lib
  # the wrapped prover build: shim TUs + the driver instrumented, the reference's prover objects as built for the oracle
  PF="-std=gnu++20 -O1 -march=haswell -madx -fopenmp -fconstexpr-ops-limit=100000000 -Wno-deprecated -DNDEBUG -fPIC -I$B"
  ( g++ $PF $GS -c $ROOT/shim/bbg_prover_wrap.cpp -o $OUT/obj/wrap_san.o ) &
  ( g++ $PF $GS -c $ROOT/shim/bbg_barretenberg_shim.cpp -o $OUT/obj/shim_pic_san.o ) &
  ( g++ $PF $GS -c $ROOT/oracle/ref_prover_driver.cpp -o $OUT/obj/driver_san.o ) &
  wait
  make -s -C $ROOT/oracle prover >/dev/null
  PROVEROBJ=$(ls $ROOT/oracle/_ref/obj/*.o | grep -v "ref_prover_driver\|bbg_shim\|bbg_prover_wrap")
  g++ -shared -fopenmp $GS -o $OUT/libbbprover_wrap_san.so $OUT/obj/driver_san.o $OUT/obj/shim_pic_san.o $OUT/obj/wrap_san.o $PROVEROBJ \
      $(cat $ROOT/shim/wrap_flags.txt) $(cat $ROOT/shim/wrap_flags_prover.txt) -L$CS -lbbg -Wl,-rpath,'$ORIGIN/../aztec-2.0_amd/csrc' -Wl,-rpath-link,/opt/rocm/lib
fi
ls -la $OUT | grep -v obj
