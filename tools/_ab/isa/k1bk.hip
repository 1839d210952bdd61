#include "../../../colpali_amd/csrc/maxsim_batch_packed.hip"
// K1bK <F16, AUX, MAXU>: the packed eight-wave form (several short documents per chunk), eight and ten units per wave
template __global__ void msim::maxsim_batch_packed_kernel<false, 0, 8>(const uint16_t *, const uint16_t *, const int32_t *, const uint8_t *, float *, msim::BatchArgs);
template __global__ void msim::maxsim_batch_packed_kernel<false, 0, 10>(const uint16_t *, const uint16_t *, const int32_t *, const uint8_t *, float *, msim::BatchArgs);
