#!/bin/bash
# K1s body A/B at 1..4 queries (16 GiB shard), interleaved inside one gpurun.
# the MSIM_* knobs exist in the measurement build only: `make -C colpali_amd/csrc ab` first
export COLPALI_AMD_LIB=${COLPALI_AMD_LIB:-tools/_ab/libmaxsim_ab.so}
export AB_DOCS=65536
for round in 1 2; do
  AB_TAG="K1s default" python tools/ab_variant.py 1,2,3,4 2>&1 | grep -v amdgpu.ids
  MSIM_STREAM_TILEMAJOR=1 AB_TAG="K1s<4> tile-major" python tools/ab_variant.py 4 2>&1 | grep -v amdgpu.ids
  MSIM_STREAM_RING=4 AB_TAG="K1s ring 4" python tools/ab_variant.py 3,4 2>&1 | grep -v amdgpu.ids
done
