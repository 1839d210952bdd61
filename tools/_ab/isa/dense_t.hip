#include "../../../colpali_amd/csrc/maxsim_batch_t.hip"
#include "../../../colpali_amd/csrc/maxsim_dense_t.hip"
// the dense hard-max backward of the transposed shape (round 6) and the forward that leaves its routing
template __global__ void msim::maxsim_batch_t_kernel<false, 2, 4, true>(const uint16_t *, const uint16_t *, float *, int32_t *, uint8_t *, msim::BatchTArgs);
template __global__ void msim::maxsim_batch_t_kernel<false, 2, 4, false>(const uint16_t *, const uint16_t *, float *, int32_t *, uint8_t *, msim::BatchTArgs);
template __global__ void msim::dense_t_bwd_long_kernel<false, 1>(const uint16_t *, const uint8_t *, const float *, msim::GScale, uint16_t *, msim::DenseTArgs);
template __global__ void msim::dense_t_bwd_long_kernel<false, 2>(const uint16_t *, const uint8_t *, const float *, msim::GScale, uint16_t *, msim::DenseTArgs);
template __global__ void msim::dense_t_bwd_short_kernel<false, 2, 2>(const uint16_t *, const uint8_t *, const float *, msim::GScale, float *, msim::DenseTArgs);
template __global__ void msim::dense_t_bwd_short_kernel<false, 1, 4>(const uint16_t *, const uint8_t *, const float *, msim::GScale, float *, msim::DenseTArgs);
