#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_topk.py -x -q -m gpu 2>&1 | tail -3
BENCH_SHARE_GPU=1 BENCH_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --docs 20000 --steps 3 --warmup 1 --regimes "" --no-cpu-baseline 2>&1 | tail -2 | cut -c1-600
