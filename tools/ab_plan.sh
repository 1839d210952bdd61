#!/bin/bash
# The single-block ladder of the flat plan (maxsim_abi.hip: flat_plan) re-measured in 16-token units: for every query count of the
# ridge the shipped choice against every shape that holds the batch in one block -- pair x 8 / x 10 units, 4 waves x 8 / x 10,
# 8 waves x 8 -- on random unit rows and on a zero-filled shard (same traffic, nothing toggling: structure without the power cap).
# Needs the measurement build: make -C colpali_amd/csrc ab.   bash tools/ab_plan.sh > gpurun_out/ab_plan.log 2>&1
export COLPALI_AMD_LIB=${COLPALI_AMD_LIB:-tools/_ab/libmaxsim_ab.so}
set -u
export AB_DOCS=${AB_DOCS:-65536}
run() { AB_TAG="$1" python tools/ab_variant.py "$2" 2>&1 | grep -v amdgpu.ids; }
for zero in 0 1; do
  export AB_ZERO=$zero
  echo "== AB_ZERO=$zero"
  AB_REF=$([ $zero = 0 ] && echo write || echo "") run "shipped plan" "5,8,9,10,11,12,14,16,17,18,20,24,32,33,36,40"
  export AB_REF=$([ $zero = 0 ] && echo check || echo "")
  MSIM_BATCH_NW=2 MSIM_BATCH_MAXU=10 run "pair x 10 units" "9,10"
  MSIM_BATCH_NW=4 MSIM_BATCH_MAXU=8  run "4 waves x 8 units" "5,8,9,10,11,12,14,16"
  MSIM_BATCH_NW=4 MSIM_BATCH_MAXU=10 run "4 waves x 10 units" "17,18,20"
  MSIM_BATCH_NW=8 MSIM_BATCH_MAXU=8  run "8 waves x 8 units" "9,10,12,16,17,18,20,24,32"
  MSIM_BATCH_NW=8 MSIM_BATCH_MAXU=10 run "8 waves x 10 units" "33,36,40"
done
