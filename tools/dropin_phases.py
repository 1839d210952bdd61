#!/usr/bin/env python
"""The bench line's drop-in leg on its own (bench_legs/dropin.py: BASELINE configs 2 / 3 from host lists, 31 calls, phase stamps), one
process per knob setting -- COLPALI_AMD_ASYNC_GATHER=0 is the pre-round-6 order (every gather blocks the calling thread; queries
packed and buffers allocated before the first gather)."""
import json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd
from bench_legs.dropin import dropin_numbers

r = dropin_numbers(amd)
tag = os.environ.get("AB_TAG", "default")
for key in ("config2_colpali_1000x1030", "config3_colqwen2_1000x267-779"):
    b = r[key]["breakdown"]
    print(f"{tag:22s} {key:32s} median {r[key]['ms']:6.2f} ms  p95 {b['p95_ms']:6.2f}  min {b['min_ms']:6.2f} | front {b['query_checks_ms']:.2f} + checks {b['checks_ms']:.2f} "
          f"| loop {b['gather_and_h2d_issue_loop_ms']:.2f} | tail {b['gpu_tail_ms_last_h2d_kernel_d2h']:.2f} | H2D floor {b['h2d_floor_ms']:.2f} ms -> {b['frac_of_h2d_roof']:.3f} of the roof  "
          f"err {r[key]['max_rel_err_vs_reference_fp32_on_this_gpu']:.1e}", flush=True)
