#!/bin/bash
# Several builds of the library (AB_LIBS="tag=path ..."; the first writes the reference scores) inside ONE gpurun, interleaved twice.
set -u
export AB_DOCS=${AB_DOCS:-65536}
SIZES=${AB_SIZES:-32,64,256}
first=1
for r in 1 2; do
  for kv in $AB_LIBS; do
    tag=${kv%%=*}; lib=${kv#*=}
    if [ $first = 1 ]; then mode=write; first=0; else mode=check; fi
    COLPALI_AMD_LIB=$lib AB_REF=$mode AB_TAG="$tag" python tools/ab_variant.py "$SIZES" 2>&1 | grep -v amdgpu.ids
  done
done
