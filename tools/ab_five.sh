#!/bin/bash
# round 4: the five-unit form of K1b's 4-wave shape (168 registers: THREE workgroups per CU) against the shipped plan at 9-10 queries x 32
# tokens, random rows and zero-filled shard.  Measurement build: make -C colpali_amd/csrc ab.
export COLPALI_AMD_LIB=${COLPALI_AMD_LIB:-tools/_ab/libmaxsim_ab.so}
set -u
export AB_DOCS=${AB_DOCS:-65536}
run() { AB_TAG="$1" python tools/ab_variant.py "$2" 2>&1 | grep -v amdgpu.ids; }
for zero in 0 1; do
  export AB_ZERO=$zero
  echo "== AB_ZERO=$zero"
  AB_REF=$([ $zero = 0 ] && echo write || echo "") run "shipped plan" "5,8,9,10,12"
  export AB_REF=$([ $zero = 0 ] && echo check || echo "")
  MSIM_BATCH_MAXU=5 MSIM_BATCH_RING5=3 run "4 waves x 5 units, 3 WG/CU, ring 3" "9,10"
  MSIM_BATCH_MAXU=5 MSIM_BATCH_RING5=2 run "4 waves x 5 units, 3 WG/CU, ring 2" "9,10"
  # not measured yet (the round's GPU minutes ran out): the same idea for the PAIR form, six pairs per CU -- 5 queries x 32 = 10 units
  MSIM_BATCH_NW=2 MSIM_BATCH_MAXU=5 MSIM_BATCH_RING5=2 run "pair x 5 units, 6 pairs/CU, ring 2" "5"
  MSIM_BATCH_NW=2 MSIM_BATCH_MAXU=5 MSIM_BATCH_RING5=3 run "pair x 5 units, 6 pairs/CU, ring 3" "5"
done
