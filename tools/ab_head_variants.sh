#!/bin/bash
# K3 variants A/B inside ONE gpurun (boxes differ by several per cent): MSIM_HEAD_VARIANT bit 0 = flag-synchronised weight ring,
# bit 1 = hand-pipelined operand fetch; two interleaved rounds.
for round in 1 2; do
  for v in 0 1 2 3; do
    echo "--- round $round MSIM_HEAD_VARIANT=$v"
    MSIM_HEAD_VARIANT=$v timeout 120 python tools/ab_head.py 2>&1 | grep K3
  done
done
