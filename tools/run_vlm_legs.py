#!/usr/bin/env python
"""bench.py's two "VLM in the loop" legs by themselves (BASELINE configs 2 / 3 with a random-init model of the named geometry)."""
import json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import bench, colpali_amd as amd

dev = torch.device("cuda:0")
for fam in (sys.argv[1:] or ["colpali", "colqwen2"]):
    print(fam, json.dumps(bench.vlm_in_the_loop_numbers(amd, dev, fam), indent=1), flush=True)
