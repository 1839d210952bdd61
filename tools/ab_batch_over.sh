#!/bin/bash
# K1b 4-wave / pair forms: more, smaller document ranges than resident workgroups (MSIM_BATCH_OVER = 1 | 2 | 4 | 8), interleaved twice inside
# ONE gpurun; bitwise check against OVER=1.  Forced 4-wave form at 24..64 queries last (two query blocks: not oversubscribed).
# the MSIM_* knobs exist in the measurement build only: `make -C colpali_amd/csrc ab` first
export COLPALI_AMD_LIB=${COLPALI_AMD_LIB:-tools/_ab/libmaxsim_ab.so}
set -u
export AB_DOCS=${AB_DOCS:-65536}
run() { AB_TAG="$1" python tools/ab_variant.py "$2" 2>&1 | grep -v amdgpu.ids; }
MSIM_BATCH_OVER=1 AB_REF=write run "over 1" "5,6,8,9,12,16"
export AB_REF=check
for r in 1 2; do
  for o in ${AB_OVERS:-4 8 2 1}; do MSIM_BATCH_OVER=$o run "over $o" "5,6,8,9,12,16"; done
done
