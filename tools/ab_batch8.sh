#!/bin/bash
# K1b (two 256-register waves per SIMD, 4 tiles each) against K1b8 (one 512-register wave per SIMD, 8 tiles) inside ONE gpurun,
# interleaved twice; the first run writes the reference scores, every later run is compared with them bit for bit.
# the MSIM_* knobs exist in the measurement build only: `make -C colpali_amd/csrc ab` first
export COLPALI_AMD_LIB=${COLPALI_AMD_LIB:-tools/_ab/libmaxsim_ab.so}
set -u
export AB_DOCS=${AB_DOCS:-65536}
SIZES=${AB_SIZES:-9,10,12,14,16,20,24,32,64,256,1000}
first=1
for r in 1 2; do
  if [ $first = 1 ]; then mode=write; first=0; else mode=check; fi
  MSIM_BATCH8=0 AB_REF=$mode AB_TAG="K1b" python tools/ab_variant.py "$SIZES" 2>&1 | grep -v amdgpu.ids
  MSIM_BATCH8=1 MSIM_BATCH8_MIN=${AB_B8_MIN:-9} AB_REF=check AB_TAG="K1b8 (from ${AB_B8_MIN:-9} tiles)" python tools/ab_variant.py "$SIZES" 2>&1 | grep -v amdgpu.ids
done
