#!/bin/bash
# one process per knob setting, the round-5 defaults first and last (drift of the box)
cd "$(dirname "$0")/.."
run() { env "$@" python tools/ab_dropin_knobs.py 2>&1 | grep -v amdgpu.ids; }
run COLPALI_AMD_EDGE_CHUNK_MB=0
run COLPALI_AMD_EDGE_CHUNK_MB=8
run COLPALI_AMD_EDGE_CHUNK_MB=4
run COLPALI_AMD_EDGE_CHUNK_MB=8 COLPALI_AMD_COPY_THREADS=12
run COLPALI_AMD_EDGE_CHUNK_MB=8 COLPALI_AMD_COPY_THREADS=6
run COLPALI_AMD_EDGE_CHUNK_MB=8 COLPALI_AMD_STAGING_MB=32
run COLPALI_AMD_EDGE_CHUNK_MB=8 COLPALI_AMD_STAGING_MB=128
run COLPALI_AMD_EDGE_CHUNK_MB=0
