// Host-visible declarations for kernels_fri.hip.
#pragma once
#include "engine.hpp"

namespace cm {

struct TwiddleView {
  uint32_t R;
  const uint32_t *xtw, *ixtw, *ytw, *iytw;
};
inline TwiddleView view(const Twiddles& t) { return TwiddleView{t.R, t.xtw, t.ixtw, t.ytw, t.iytw}; }

// One ColumnSampleBatch of a size group (Stwo core::pcs::quotients).
struct QuotientBatch {
  uint32_t begin, end;       // entry range in col_index / coef_c
  uint32_t point[8];         // sample point: x = (point[0..4]), y = (point[4..8]) as QM31 words
  uint32_t sum_a[4], sum_b[4];  // sum over the batch of alpha^i * a_i and alpha^i * b_i
  uint32_t batch_coeff[4];   // random_coeff ^ (#columns in the batch)
};
struct QuotientArgs {
  TwiddleView tw;
  uint32_t log_size;
  const uint32_t* const* cols;     // LDE columns of the size group (device array)
  const uint32_t* col_index;       // per entry: column index in `cols`
  const uint32_t* coef_c;          // per entry: alpha^i * c_i (4 u32)
  const QuotientBatch* batches;    // device array
  uint32_t n_batches;
  uint32_t* const* out;            // 4 coordinate columns (device array)
};
void launch_quotients(const QuotientArgs& a, double n_cols, hipStream_t st);
void fold_circle_into_line(uint32_t* const dst[4], const uint32_t* const src[4], uint32_t log_n, const Twiddles& tw,
                           const QM31& alpha, bool accumulate, hipStream_t st);
void fold_line(uint32_t* const out[4], const uint32_t* const src[4], uint32_t log_n, const Twiddles& tw, const QM31& alpha,
               hipStream_t st);

}  // namespace cm
