#!/bin/bash
# K1b (8-wave form) scheduling experiments behind MSIM_BATCH_TUNE, interleaved twice inside ONE gpurun; bitwise check against tune 0.
set -u
export AB_DOCS=${AB_DOCS:-65536}
SIZES=${AB_SIZES:-32,64,256}
run() { AB_TAG="$1" python tools/ab_variant.py "$SIZES" 2>&1 | grep -v amdgpu.ids; }
MSIM_BATCH_TUNE=0 AB_REF=write run "tune 0"
for r in 1 2; do
  for t in ${AB_TUNES:-1 2 3 0}; do MSIM_BATCH_TUNE=$t AB_REF=check run "tune $t"; done
done
