// Host-visible declarations of kernels_shard.hip.
#pragma once
#include "engine.hpp"

namespace cm {

struct CopySeg { const uint32_t* src; uint32_t* dst; uint64_t words; };
// dst[i] = src[i] for every segment (device-to-device), one launch, enqueued on `st` (no host synchronisation)
void copy_segments(const std::vector<CopySeg>& segs, hipStream_t st);
// d_out[i] = sum_k d_in[k * stride + i], k < n_copies; modular = sums of M31 values, else plain u32 (counts)
void sum_copies(const uint32_t* d_in, uint32_t n_copies, uint64_t stride, uint64_t words, uint32_t* d_out, bool modular, hipStream_t st);

// ---- previous-row halo of a row-sharded column (round 6) ----
// On bit-reversed storage the previous trace row of EVERY row of a rank's range lies in ONE other rank's range: the even local
// positions (first half of the domain: natural index i -> i - 1) in the range whose bit-reversed rank number is one LOWER, the odd ones
// (second half: i -> i + 1) in the range one HIGHER.  A rank therefore sends the even half of its slice to one neighbour and the odd
// half to the other, instead of every rank gathering the whole column.
// dst[i] = src[2 i + parity], i < half_len  (packs one half of a slice for the exchange)
void pack_parity(const uint32_t* d_src, uint32_t* d_dst, uint32_t half_len, uint32_t parity, hipStream_t st);
// out[q] = value of the column at the previous row of global position row0 + q, q < len: looked up in the half slices received from
// the two neighbours (`even_half` = even local positions of rank `even_src`'s slice, `odd_half` likewise); n = log of the domain,
// trace_log = n - 1, log_ranks = log2 N.  A lookup that falls outside the expected neighbour sets *d_err.
void halo_build(const uint32_t* d_even_half, const uint32_t* d_odd_half, uint32_t even_src, uint32_t odd_src, uint32_t row0, uint32_t len,
                uint32_t n, uint32_t trace_log, uint32_t log_ranks, uint32_t* d_out, uint32_t* d_err, hipStream_t st);

}  // namespace cm
