#!/bin/bash
# Ridge-regime A/B (5..16 queries of 32 tokens, 16 GiB shard): K1s / K1b defaults against the pair form (NW=2) and one-pass bodies.
# Usage on the GPU box:  bash tools/ab_ridge.sh > gpurun_out/ab_ridge.log 2>&1
set -u
export AB_DOCS=${AB_DOCS:-65536}
run() { AB_TAG="$1" python tools/ab_variant.py "$2" 2>&1 | grep -v amdgpu.ids; }
AB_REF=write run "default" "4,5,6,7,8,9,10,12,14,16,24,32,64"
export AB_REF=check
MSIM_STREAM_MAX_TILES=4 run "K1b NW4 (tiles>4)" "5,6,7,8"
for v in 2,4,1 2,4,0 2,3,1; do
  MSIM_STREAM_MAX_TILES=4 MSIM_BATCH_EXP=$v run "pair NW,RING,ONEPASS=$v" "5,6,7,8"
done
for v in 4,3,1 8,3,1; do
  MSIM_STREAM_MAX_TILES=4 MSIM_BATCH_EXP=$v run "K1b NW,RING,ONEPASS=$v" "8,9,10,12,14,16,24,32,64"
done
# DVFS check: the same binaries on a zero-filled corpus (same HBM traffic, no operand toggling in the matrix pipe)
export AB_REF=
AB_ZERO=1 run "default, zero corpus" "1,4,8,12,16,32,64"
AB_ZERO=1 MSIM_STREAM_MAX_TILES=4 MSIM_BATCH_EXP=2,4,1 run "pair 2,4,1, zero corpus" "5,6,7,8"
AB_ZERO=1 MSIM_BATCH_EXP=8,3,1 run "K1b 8,3,1, zero corpus" "32,64"
