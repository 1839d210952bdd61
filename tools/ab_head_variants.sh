#!/bin/bash
# K3 variants A/B inside ONE gpurun (boxes differ by several per cent): MSIM_HEAD_VARIANT bit 0 = flag-synchronised weight ring,
# bit 1 = hand-pipelined operand fetch, bit 2 (value 4, the default) = swapped MFMA roles + per-row epilogue; two interleaved rounds.
# the MSIM_* knobs exist in the measurement build only: `make -C colpali_amd/csrc ab` first
export COLPALI_AMD_LIB=${COLPALI_AMD_LIB:-tools/_ab/libmaxsim_ab.so}
for round in 1 2; do
  for v in ${AB_HEAD_VARIANTS:-0 4}; do
    echo "--- round $round MSIM_HEAD_VARIANT=$v"
    MSIM_HEAD_VARIANT=$v timeout 120 python tools/ab_head.py 2>&1 | grep K3
  done
done
