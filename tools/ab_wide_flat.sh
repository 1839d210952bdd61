#!/bin/bash
# width 320: the dispatch's choice (shipped library), then the measurement build forced to the box form and to the flat form
cd "$(dirname "$0")/.."
python tools/ab_wide_flat.py
COLPALI_AMD_LIB=$PWD/tools/_ab/libmaxsim_ab.so MSIM_PANELS_FLAT=0 python tools/ab_wide_flat.py
COLPALI_AMD_LIB=$PWD/tools/_ab/libmaxsim_ab.so MSIM_PANELS_FLAT=1 python tools/ab_wide_flat.py
