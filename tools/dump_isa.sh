#!/bin/bash
# Regenerate the gfx950 assembly of the kernels DESIGN.md quotes instruction counts for (not committed: ~90k generated lines) and
# print the counts: MFMAs, LDS-DMA loads, operand fetches, accumulator copies, compiler-inserted vmcnt waits, wait states.
#   bash tools/dump_isa.sh            -> tools/_ab/isa/{k1b,k3}.s + a summary
set -eu
cd "$(dirname "$0")/.."
for k in k1b k3; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -Icolpali_amd/csrc -S --cuda-device-only \
      -o tools/_ab/isa/$k.s tools/_ab/isa/$k.hip
  echo "== $k: $(wc -l < tools/_ab/isa/$k.s) lines"
  for pat in v_mfma "buffer_load.*lds" ds_read_b128 v_accvgpr "s_waitcnt vmcnt" s_nop v_max3 s_barrier; do
    printf "  %-22s %6d\n" "$pat" "$(grep -c -E "$pat" tools/_ab/isa/$k.s || true)"
  done
done
