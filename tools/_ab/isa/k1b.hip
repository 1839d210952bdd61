#include "../../../colpali_amd/csrc/maxsim_stream.hip"
#include "../../../colpali_amd/csrc/maxsim_batch.hip"
// <F16, NW, RING, AUX, MAXU>: the 8-wave form (several query blocks), the 4-wave and pair forms (one block, nt), the ten-unit form
template __global__ void msim::maxsim_batch_kernel<false, 8, 3, 0, 8>(const uint16_t *, const uint16_t *, const int32_t *, const uint8_t *, float *, msim::BatchArgs);
template __global__ void msim::maxsim_batch_kernel<false, 4, 3, 2, 8>(const uint16_t *, const uint16_t *, const int32_t *, const uint8_t *, float *, msim::BatchArgs);
template __global__ void msim::maxsim_batch_kernel<false, 2, 4, 2, 8>(const uint16_t *, const uint16_t *, const int32_t *, const uint8_t *, float *, msim::BatchArgs);
template __global__ void msim::maxsim_batch_kernel<false, 8, 3, 0, 10>(const uint16_t *, const uint16_t *, const int32_t *, const uint8_t *, float *, msim::BatchArgs);
// K1s <NU, RING, F16, AUX, IL>: the headline (4 queries x 32 tokens = 8 units) and one query
template __global__ void msim::maxsim_stream_kernel<8, 2, false, 2, true>(const uint16_t *, const uint16_t *, const int32_t *, const uint8_t *, float *, msim::StreamArgs);
template __global__ void msim::maxsim_stream_kernel<2, 4, false, 2, true>(const uint16_t *, const uint16_t *, const int32_t *, const uint8_t *, float *, msim::StreamArgs);
