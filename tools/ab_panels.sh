#!/bin/bash
# dim-320 panel kernels: this build (16x16x32 tiles) against tools/_ab/libmaxsim_prev.so (32x32x16 tiles) inside one gpurun
for round in 1 2; do
  echo "--- prev build (32x32x16 tiles)"; COLPALI_AMD_LIB=$PWD/tools/_ab/libmaxsim_prev.so python tools/ab_generic.py 2>&1 | grep "dim= 320"
  echo "--- this build (16x16x32 tiles)"; python tools/ab_generic.py 2>&1 | grep "dim= 320"
done
