#!/bin/bash
# nt (no L2 / MALL allocation) on K1b's document stream when one query block reads the corpus, interleaved inside one gpurun
# the MSIM_* knobs exist in the measurement build only: `make -C colpali_amd/csrc ab` first
export COLPALI_AMD_LIB=${COLPALI_AMD_LIB:-tools/_ab/libmaxsim_ab.so}
export AB_DOCS=65536
for round in 1 2; do
  MSIM_BATCH_NT=0 AB_TAG="K1b default policy" python tools/ab_variant.py 5,8,12,16,24,32 2>&1 | grep -v amdgpu.ids
  AB_TAG="K1b nt when single block" python tools/ab_variant.py 5,8,12,16,24,32 2>&1 | grep -v amdgpu.ids
done
