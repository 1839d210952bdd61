#!/bin/bash
# Ridge-regime sweep (5..32 queries of 32 tokens, 16 GiB shard): the shipped dispatch against forced stream-sharing widths
# (MSIM_BATCH_NW) and the DVFS check (same binaries on a zero-filled corpus: same HBM traffic, nothing toggling in the matrix pipe).
# The log committed as profiles/r02_logs/ab_ridge.log was taken with the round-2 development knobs of that commit (32x32x16
# tiles; K1s up to 8 tiles, two-pass K1b as "default"); this script reproduces the sweep on the current build.
# Usage on the GPU box:  bash tools/ab_ridge.sh > gpurun_out/ab_ridge.log 2>&1
# the MSIM_* knobs exist in the measurement build only: `make -C colpali_amd/csrc ab` first
export COLPALI_AMD_LIB=${COLPALI_AMD_LIB:-tools/_ab/libmaxsim_ab.so}
set -u
export AB_DOCS=${AB_DOCS:-65536}
run() { AB_TAG="$1" python tools/ab_variant.py "$2" 2>&1 | grep -v amdgpu.ids; }
AB_REF=write run "default" "4,5,6,7,8,9,10,12,14,16,24,32,64"
export AB_REF=check
MSIM_BATCH_NW=4 run "K1b forced NW=4" "5,6,7,8,24,32"
MSIM_BATCH_NW=8 run "K1b forced NW=8" "8,12,16"
MSIM_BATCH_NW=2 run "K1b forced NW=2 (<= 8 tiles)" "5,6,7,8"
export AB_REF=
AB_ZERO=1 run "default, zero corpus" "1,4,8,12,16,32,64"
