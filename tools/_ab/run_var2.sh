set -u
export AB_DOCS=65536
SIZES=256
MSIM_BATCH8=0 AB_REF=write AB_TAG="K1b" python tools/ab_variant.py "$SIZES" 2>&1 | grep -v amdgpu.ids
for v in 0 1 8 9 3; do
  MSIM_BATCH8=1 MSIM_B8_VAR=$v AB_REF=check AB_TAG="K1b8 var $v" python tools/ab_variant.py "$SIZES" 2>&1 | grep -v amdgpu.ids
done
