#!/usr/bin/env python
"""bench.py's BASELINE config 5 leg (`loss_step_config5`) by itself."""
import json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import bench, colpali_amd as amd

print(json.dumps(bench.loss_step_numbers(amd, torch.device("cuda:0")), indent=1), flush=True)
