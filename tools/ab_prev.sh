#!/bin/bash
# Current build against the previous one (tools/_ab/libmaxsim_prev.so, loaded through COLPALI_AMD_LIB) inside ONE gpurun, interleaved
# twice; the previous build writes the reference scores, the current one must reproduce them bit for bit.
# Usage on the GPU box:  AB_SIZES=8,12,16,32,64 bash tools/ab_prev.sh > gpurun_out/ab_prev.log 2>&1
set -u
export AB_DOCS=${AB_DOCS:-65536}
SIZES=${AB_SIZES:-8,12,16,24,32,64,256}
run() { AB_TAG="$1" python tools/ab_variant.py "$SIZES" 2>&1 | grep -v amdgpu.ids; }
COLPALI_AMD_LIB=tools/_ab/libmaxsim_prev.so AB_REF=write run "previous build"
export AB_REF=check
for r in 1 2; do
  run "this build"
  if [ -n "${AB_THIS_ENV:-}" ]; then env $AB_THIS_ENV bash -c "AB_TAG='this build, $AB_THIS_ENV' python tools/ab_variant.py $SIZES 2>&1 | grep -v amdgpu.ids"; fi
  COLPALI_AMD_LIB=tools/_ab/libmaxsim_prev.so run "previous build"
done
