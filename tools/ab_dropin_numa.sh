#!/bin/bash
# the drop-in call with and without the NUMA-local host side (COLPALI_AMD_NUMA=0|1), several PROCESSES each: placement is per process
cd "$(dirname "$0")/.."
for i in 1 2 3 4 5; do
  for numa in 0 1; do
    COLPALI_AMD_NUMA=$numa python tools/ab_dropin_knobs.py 2>&1 | grep -v amdgpu.ids
  done
done
