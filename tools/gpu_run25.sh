#!/bin/bash
python -m pytest tests/test_gpu_abi.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -30
