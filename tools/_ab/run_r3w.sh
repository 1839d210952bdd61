set -u
mkdir -p gpurun_out/r3w
# NOTE: in the measurement build K1b8 takes every size from 21 tiles up unless MSIM_BATCH8=0 -- an A/B of K1b plans needs it switched off
export COLPALI_AMD_LIB=tools/_ab/libmaxsim_ab.so AB_DOCS=65536 MSIM_BATCH8=0
SIZES=33,34,35,36,40,64,80,256,1000
( for r in 1 2; do
    MSIM_BATCH_T5=0 AB_REF=$([ $r = 1 ] && echo write || echo check) AB_TAG="four tiles per wave" python tools/ab_variant.py $SIZES
    MSIM_BATCH_T5=7 AB_REF=check AB_TAG="36..40 in one pass x 5 tiles (shipped plan)" python tools/ab_variant.py $SIZES
    MSIM_BATCH_T5=7 MSIM_BATCH_T5_LO=33 AB_REF=check AB_TAG="33..40 in one pass x 5 tiles" python tools/ab_variant.py $SIZES
    MSIM_BATCH_T5=15 AB_REF=check AB_TAG="+ blocks of 40 above" python tools/ab_variant.py $SIZES
  done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r3w/ab_t5_above32.log
cat gpurun_out/r3w/ab_t5_above32.log
