set -u
mkdir -p gpurun_out/r3g
export COLPALI_AMD_LIB=tools/_ab/libmaxsim_ab.so
( MSIM_HEAD_PAIR=1 timeout 300 python -m pytest tests/test_gpu_head.py -x -q -m gpu 2>&1 | tail -4
  for round in 1 2 3; do
    echo "--- round $round default (loader two chunks ahead, rings 3 + 3, whole-row nt stores)"
    timeout 120 python tools/ab_head.py 2>&1 | grep K3
    echo "--- round $round MSIM_HEAD_PAIR=1 (chunks requested two at a time, rings 4 + 2)"
    MSIM_HEAD_PAIR=1 timeout 120 python tools/ab_head.py 2>&1 | grep K3
  done ) > gpurun_out/r3g/ab_head_pair.log 2>&1
cat gpurun_out/r3g/ab_head_pair.log
