set -u
mkdir -p gpurun_out/r3c
export AB_DOCS=65536
AB=tools/_ab/libmaxsim_ab.so
( SIZES=17,18,20,24
  AB_REF=write AB_TAG="K1b default" python tools/ab_variant.py "$SIZES"
  MSIM_BATCH_NW=8 AB_REF=check AB_TAG="K1b forced NW=8" python tools/ab_variant.py "$SIZES"
  COLPALI_AMD_LIB=$AB MSIM_BATCH8=1 MSIM_BATCH8_MIN=17 MSIM_B8_VAR=0 AB_REF=check AB_TAG="K1b8 var0" python tools/ab_variant.py "$SIZES"
  COLPALI_AMD_LIB=$AB MSIM_BATCH8=1 MSIM_BATCH8_MIN=17 MSIM_B8_VAR=1 AB_REF=check AB_TAG="K1b8 var1 (deferred folds)" python tools/ab_variant.py "$SIZES"
  SIZES=9,10,12,14,16
  AB_REF=write AB_TAG="K1b default" python tools/ab_variant.py "$SIZES"
  COLPALI_AMD_LIB=$AB MSIM_BATCH8=1 MSIM_BATCH8_MIN=9 MSIM_B8_VAR=1 AB_REF=check AB_TAG="K1b8 pair, var1" python tools/ab_variant.py "$SIZES"
  SIZES=32,64,1000
  AB_REF=write AB_TAG="K1b default" python tools/ab_variant.py "$SIZES"
  COLPALI_AMD_LIB=$AB MSIM_BATCH8=1 MSIM_BATCH8_MIN=21 MSIM_B8_VAR=1 AB_REF=check AB_TAG="K1b8 var1" python tools/ab_variant.py "$SIZES"
) 2>&1 | grep -v amdgpu.ids > gpurun_out/r3c/ab_batch8_final.log
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_head.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r3c/pytest_new.log
timeout 300 python tools/power_sample.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3c/power_sample.log
cat gpurun_out/r3c/pytest_new.log gpurun_out/r3c/ab_batch8_final.log gpurun_out/r3c/power_sample.log
