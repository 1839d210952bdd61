set -u
mkdir -p gpurun_out/r3h
export COLPALI_AMD_LIB=tools/_ab/libmaxsim_ab.so
( for round in 1 2; do
    for cfg in "0 1" "4 1" "8 1" "16 1" "32 1" "8 2" "32 2"; do
      set -- $cfg
      echo "--- round $round MSIM_HEAD_STAGGER=$1 MSIM_HEAD_STAGGER_SLEEP=$2"
      MSIM_HEAD_STAGGER=$1 MSIM_HEAD_STAGGER_SLEEP=$2 timeout 120 python tools/ab_head.py 2>&1 | grep K3
    done
  done ) > gpurun_out/r3h/ab_head_stagger.log 2>&1
cat gpurun_out/r3h/ab_head_stagger.log
