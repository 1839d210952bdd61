#!/bin/bash
# The host-core scorer (colpali_amd/csrc/maxsim_host.cpp: persistent thread pool, concurrent callers) under AddressSanitizer +
# UBSan and under ThreadSanitizer: four threads call msim_fwd_host concurrently, 20 times each, results compared bit for bit with a
# single-threaded call.  CPU only.   bash tools/host_sanitize/run.sh
set -eu
cd "$(dirname "$0")"
SRC="harness.cpp ../../colpali_amd/csrc/maxsim_host.cpp"
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fPIC -o /tmp/msim_host_asan $SRC -lpthread
/tmp/msim_host_asan
# (no ifunc clones under TSAN: a resolver runs before the sanitizer's runtime is up)
g++ -O1 -g -std=c++17 -fsanitize=thread -DMSIM_HOST_NO_CLONES -mavx2 -mfma -fPIC -o /tmp/msim_host_tsan $SRC -lpthread
TSAN_OPTIONS="halt_on_error=1" /tmp/msim_host_tsan
echo "host scorer: ASAN/UBSAN and TSAN clean"
