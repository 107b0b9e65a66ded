// Host-visible declarations of kernels_shard.hip.
#pragma once
#include "engine.hpp"

namespace cm {

struct CopySeg { const uint32_t* src; uint32_t* dst; uint64_t words; };
// dst[i] = src[i] for every segment (device-to-device), one launch; synchronises `st` before returning
void copy_segments(const std::vector<CopySeg>& segs, hipStream_t st);
// d_out[i] = sum_k d_in[k * stride + i], k < n_copies; modular = sums of M31 values, else plain u32 (counts)
void sum_copies(const uint32_t* d_in, uint32_t n_copies, uint64_t stride, uint64_t words, uint32_t* d_out, bool modular, hipStream_t st);

}  // namespace cm
