#!/bin/bash
# scripts/ab/msm_variant.sh <tag> <extra hipcc flags...>: an A/B build of the MSM translation units (msm.hip + every msm_wNN.hip) with extra flags,
# linked with the shipped objects into build_ab/libbbg_<tag>.so (same ABI; select with BBG_LIB_PATH).  Experiments only, never shipped.
set -e
cd "$(dirname "$0")/../../aztec-2.0_amd/csrc"
tag=$1; shift
mkdir -p ../../build_ab
objs=""
for tu in msm msm_tiny msm_w13 msm_w16 msm_w17 msm_w19 msm_w20 msm_w22; do
  /opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I/opt/rocm/include -c $tu.hip -o ../../build_ab/${tu}_$tag.o &
  objs="$objs ../../build_ab/${tu}_$tag.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ab/libbbg_$tag.so $objs ntt.o poly.o quotient.o prover.o multi.o bbg_capi.o
echo built $tag
