#!/bin/bash
# Round 5: the lead DESIGN.md section 8 left open -- "more workgroups per CU where a launch needs few units per wave": the pair form
# instantiated for at most FIVE units per wave (168 registers: SIX pairs per CU instead of four), measurement build only
# (MSIM_BATCH_NW=2 MSIM_BATCH_MAXU=5, with the 2- and the 3-chunk ring), against the shipped plan for every batch it can hold (<= 10 units):
# 4 x 40 tokens, 5 x 32, and through ab_variant's uniform 32-token queries 5 queries (10 units; 3 and 4 queries are K1s territory); random unit rows and a zero-filled shard.
#   make -C colpali_amd/csrc ab;  bash tools/ab_six_pairs.sh > gpurun_out/ab_six_pairs.log 2>&1
export COLPALI_AMD_LIB=${COLPALI_AMD_LIB:-tools/_ab/libmaxsim_ab.so}
set -u
export AB_DOCS=${AB_DOCS:-65536}
run() { AB_TAG="$1" python tools/ab_variant.py "$2" 2>&1 | grep -v amdgpu.ids; }
for zero in 0 1; do
  export AB_ZERO=$zero
  echo "== AB_ZERO=$zero"
  for rep in 1 2; do
    AB_REF=$([ $zero = 0 ] && [ $rep = 1 ] && echo write || echo check) run "shipped plan" "5"
    export AB_REF=check
    MSIM_BATCH_NW=2 MSIM_BATCH_MAXU=5 MSIM_BATCH_RING5=2 run "pair x 5 units, six pairs per CU, 2-chunk ring" "5"
    MSIM_BATCH_NW=2 MSIM_BATCH_MAXU=5 MSIM_BATCH_RING5=3 run "pair x 5 units, 3-chunk ring" "5"
    MSIM_BATCH_NW=2 MSIM_BATCH_MAXU=8 run "pair x 8 units (the shipped pair form)" "5"
  done
done
