#include "../../../colpali_amd/csrc/embed_head.hip"
template __global__ void msim::embed_head_kernel<false, false, false, false, true, false, false, true>(const uint16_t *, const uint16_t *, const uint16_t *, const int32_t *, uint16_t *, msim::HeadArgs);
