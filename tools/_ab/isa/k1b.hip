#include "../../../colpali_amd/csrc/maxsim_batch.hip"
template __global__ void msim::maxsim_batch_kernel<1, false, 8, 3, 0>(const uint16_t *, const uint16_t *, const int32_t *, const uint8_t *, float *, msim::BatchArgs);
template __global__ void msim::maxsim_batch_kernel<1, false, 4, 3, 2>(const uint16_t *, const uint16_t *, const int32_t *, const uint8_t *, float *, msim::BatchArgs);
template __global__ void msim::maxsim_batch_kernel<1, false, 2, 4, 2>(const uint16_t *, const uint16_t *, const int32_t *, const uint8_t *, float *, msim::BatchArgs);
template __global__ void msim::maxsim_batch_kernel<2, false, 8, 3, 0>(const uint16_t *, const uint16_t *, const int32_t *, const uint8_t *, float *, msim::BatchArgs);
