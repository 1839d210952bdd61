#!/bin/bash
# TEMPORARY: decompose K3's time (bit 1: no barriers, bit 2: no W loads, bit 4: no MFMA / ds_read)
for e in 0 1 2 3 4 5 6 7; do
  echo "== MSIM_HEAD_EXP=$e"
  MSIM_HEAD_EXP=$e python tools/ab_head.py 2>&1 | grep "K3 fused"
done
