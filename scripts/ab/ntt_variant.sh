#!/bin/bash
# scripts/ab/ntt_variant.sh <tag> <extra hipcc flags...>: an A/B build of ntt.hip alone (BBG_NTT_FAST_AB: k_ntt_pass29 at radix 2^10 only: n = 2^20),
# (FULL=1 in the environment: every instantiation, minutes instead of seconds)
# linked with the shipped objects into build_ab/libbbg_<tag>.so (same ABI; select with BBG_LIB_PATH).  Experiments only, never shipped.
set -e
cd "$(dirname "$0")/../../aztec-2.0_amd/csrc"
tag=$1; shift
mkdir -p ../../build_ab
FAST=-DBBG_NTT_FAST_AB; [ -n "${FULL:-}" ] && FAST=
/opt/rocm/bin/hipcc $FAST "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I/opt/rocm/include -c ntt.hip -o ../../build_ab/ntt_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ab/libbbg_$tag.so ../../build_ab/ntt_$tag.o msm.o msm_tiny.o msm_w13.o msm_w16.o msm_w17.o msm_w19.o msm_w20.o msm_w22.o poly.o quotient.o prover.o multi.o bbg_capi.o
echo built $tag
