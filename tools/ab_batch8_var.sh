#!/bin/bash
# K1b8 measurement variants (MSIM_B8_VAR, maxsim_batch8.hip) against K1b inside one gpurun.  Variants 2 / 4 / 6 / 7 are KNOCK-OUTS
# (no LDS-DMA issue in the slab body / no chunk barrier): wrong scores on purpose, they price a component.
# the MSIM_* knobs exist in the measurement build only: `make -C colpali_amd/csrc ab` first
export COLPALI_AMD_LIB=${COLPALI_AMD_LIB:-tools/_ab/libmaxsim_ab.so}
set -u
export AB_DOCS=${AB_DOCS:-65536}
SIZES=${AB_SIZES:-32,256}
MSIM_BATCH8=0 AB_REF=write AB_TAG="K1b" python tools/ab_variant.py "$SIZES" 2>&1 | grep -v amdgpu.ids
for v in 0 1 2 4 6 7; do
  MSIM_BATCH8=1 MSIM_B8_VAR=$v AB_REF=check AB_TAG="K1b8 var $v" python tools/ab_variant.py "$SIZES" 2>&1 | grep -v amdgpu.ids
done
MSIM_BATCH8=0 AB_ZERO=1 AB_TAG="K1b, zero corpus" python tools/ab_variant.py "$SIZES" 2>&1 | grep -v amdgpu.ids
for v in 0 1 7; do
  MSIM_BATCH8=1 MSIM_B8_VAR=$v AB_ZERO=1 AB_TAG="K1b8 var $v, zero corpus" python tools/ab_variant.py "$SIZES" 2>&1 | grep -v amdgpu.ids
done
