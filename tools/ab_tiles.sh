#!/bin/bash
# 16x16x32 tiling (this build) against the previous build (32x32x16 tiles; tools/_ab/libmaxsim_prev.so, built from the parent commit)
# inside ONE gpurun, interleaved, 16 GiB shard.
export AB_DOCS=65536
for round in 1 2; do
  COLPALI_AMD_LIB=$PWD/tools/_ab/libmaxsim_prev.so AB_TAG="32x32x16 tiles (prev build)" python tools/ab_variant.py 1,4,8,12,16,32,64 2>&1 | grep -v amdgpu.ids
  AB_TAG="16x16x32 tiles" python tools/ab_variant.py 1,4,8,12,16,32,64 2>&1 | grep -v amdgpu.ids
done
