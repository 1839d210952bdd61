#!/bin/bash
# Five token tiles per wave (maxsim_batch.hip MAXT = 5, still two waves per SIMD) against the four-tile plan, inside ONE gpurun, interleaved
# twice; MSIM_BATCH_T5 bit 0: 9..10 queries on the pair form, bit 1: 17..20 on the 4-wave form, bit 2: 33..40 in one pass of the 8-wave
# form, bit 3: blocks of 40 above.  The first run writes the reference scores; every later run is compared with them bit for bit.
# the MSIM_* knobs exist in the measurement build only: `make -C colpali_amd/csrc ab` first
export COLPALI_AMD_LIB=${COLPALI_AMD_LIB:-tools/_ab/libmaxsim_ab.so}
export MSIM_BATCH8=0      # K1b, not K1b8 (the default of the measurement build since; it was 1 when the first table was taken)
export AB_DOCS=${AB_DOCS:-65536}
SIZES=${AB_SIZES:-9,10,17,18,20,33,36,40,64,80,256,1000}
first=1
for r in 1 2; do
  if [ $first = 1 ]; then mode=write; first=0; else mode=check; fi
  MSIM_BATCH_T5=0 AB_REF=$mode AB_TAG="four tiles per wave" python tools/ab_variant.py "$SIZES" 2>&1 | grep -v amdgpu.ids
  MSIM_BATCH_T5=15 AB_REF=check AB_TAG="five tiles per wave" python tools/ab_variant.py "$SIZES" 2>&1 | grep -v amdgpu.ids
done
