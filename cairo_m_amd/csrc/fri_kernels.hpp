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
// Per-batch job of k_quotient_coeffs: fills coef_c[begin, end), sum_a, sum_b and batch_coeff of *qb from the sampled
// values (already in HBM after the OODS kernels) and the random coefficient (a kernel argument)
struct QuotientCoefJob {
  QuotientBatch* qb;          // begin / end / point set by the host
  uint32_t* coef_c;           // the group's coef_c array (4 words per entry)
  const uint32_t* sample_idx; // the group's per-entry index of the sampled value (QM31 = 4 words at 4 * idx)
};
void quotient_coeffs(const QuotientCoefJob* d_jobs, uint32_t n_jobs, const uint32_t* d_samples, const QM31& coeff, hipStream_t st);
struct QuotientArgs {
  TwiddleView tw;
  uint32_t log_size;
  const uint32_t* const* cols;     // LDE columns of the size group (device array)
  const uint32_t* col_index;       // per entry: column index in `cols`
  const uint32_t* const* entry_cols = nullptr;   // optional, per entry: cols[col_index[e]] resolved by the host (one scalar load
                                                 // per column instead of two dependent ones)
  const uint32_t* coef_c;          // per entry: alpha^i * c_i (4 u32)
  const QuotientBatch* batches;    // device array
  uint32_t n_batches;
  uint32_t* const* out;            // 4 coordinate columns (device array)
  // row-sharded launch (intra-proof multi-GPU): this rank computes rows [row0, row0 + n_rows) of the 2^log_size domain;
  // `cols` / `out` then point at the rank's row SLICES (index 0 = row0).  n_rows = 0: the whole domain.
  uint32_t row0 = 0, n_rows = 0;
  // (round 6) the largest size group of a single-GPU proof: the FRI first-layer tree's LEAF layer (hash of the row's four
  // coordinate words, Stwo MerkleOps::commit_on_layer with no children) is written by the quotient kernel itself, 32 B per row
  // — the row's value is hashed in the registers it was computed in, next to a kernel that otherwise waits on HBM.  null = no.
  uint32_t* leaf_hashes = nullptr;
};
void launch_quotients(const QuotientArgs& a, double n_cols, hipStream_t st);
bool quotient_leaf_serves(const QuotientArgs& a);   // may `leaf_hashes` be set for this launch?
void fold_circle_into_line(uint32_t* const dst[4], const uint32_t* const src[4], uint32_t log_n, const Twiddles& tw,
                           const QM31& alpha, bool accumulate, hipStream_t st, const uint32_t* d_alpha = nullptr);
// d_alpha != null: the folding challenge is read from device memory (4 u32) instead of `alpha`
void fold_line(uint32_t* const out[4], const uint32_t* const src[4], uint32_t log_n, const Twiddles& tw, const QM31& alpha,
               hipStream_t st, const uint32_t* d_alpha = nullptr);
// Row-range forms (sharded FRI: a rank folds its own slice; folds are pair-local, so rows [2 i0, 2 (i0 + n_out)) of the source
// give rows [i0, i0 + n_out) of the result): dst / out / src point at arrays that hold ONLY that range.
void fold_circle_into_line_rows(uint32_t* const dst[4], const uint32_t* const src[4], uint32_t log_n, const Twiddles& tw,
                                const QM31& alpha, bool accumulate, uint32_t i0, uint32_t n_out, hipStream_t st,
                                const uint32_t* d_alpha = nullptr);
void fold_line_rows(uint32_t* const out[4], const uint32_t* const src[4], uint32_t log_n, const Twiddles& tw, const QM31& alpha,
                    uint32_t i0, uint32_t n_out, hipStream_t st, const uint32_t* d_alpha = nullptr);

// Tail of the FRI commit phase in ONE launch (one 1024-thread block): for every remaining layer
// (2^top_log ... 2^(last_log+1) values) fold the pending circle quotients in, build the layer's Merkle
// tree, do the transcript step (mix_root, draw the folding challenge) and fold to the next layer.  These
// layers are pure launch/dependency latency as separate kernels (~6 launches each).
constexpr uint32_t FRI_TAIL_MAX_LOG = 13;
constexpr uint32_t FRI_TAIL_DEFAULT_LOG = 10;  // env CM_FRI_TAIL_LOG overrides (tuning)
struct FriTailLayer {
  uint32_t* cols[4];                        // line evaluation of the layer (4 coordinates x 2^log)
  const uint32_t* circle[4];                // quotient columns of log + 1 to fold in first, or null
  uint32_t* merkle[FRI_TAIL_MAX_LOG + 1];   // Merkle layers of the layer's tree: merkle[k] has 2^k nodes
};
struct FriTailArgs {
  TwiddleView tw;
  uint32_t top_log, last_log;               // layers top_log .. last_log+1 are committed; layers[last_log].cols is the output
  uint32_t first_index;                     // index of layer top_log in alphas (4 u32 each) / roots (8 u32 each)
  uint32_t *chan, *alphas, *roots;          // device transcript state, challenges (alphas[0..4) = circle alpha), root log
  FriTailLayer layers[FRI_TAIL_MAX_LOG + 1];  // indexed by log
};
void fri_tail(const FriTailArgs& a, hipStream_t st);
// fold_line of layer log_n into layer log_n - 1 fused with fold_circle_into_line of the quotient columns of log_n
void fold_line_and_circle(uint32_t* const out[4], const uint32_t* const src[4], const uint32_t* const circle[4], uint32_t log_n,
                          const Twiddles& tw, hipStream_t st, const uint32_t* d_alpha, const uint32_t* d_alpha_circle);
// (row0, n_rows: outputs [row0, row0 + n_rows) of the layer into arrays — values, sources, leaf hashes — that hold just that row range, the
// sharded prover's slices; n_rows = 0: the whole layer)
bool fold_circle_leaf(uint32_t* const out[4], const uint32_t* const circle[4], uint32_t log_n, const Twiddles& tw, hipStream_t st,
                      const uint32_t* d_alpha_circle, uint32_t* d_leaf_hashes, uint32_t row0 = 0, uint32_t n_rows = 0);
bool fold_line_leaf(uint32_t* const out[4], const uint32_t* const src[4], const uint32_t* const* circle, uint32_t log_n,
                    const Twiddles& tw, hipStream_t st, const uint32_t* d_alpha, const uint32_t* d_alpha_circle, uint32_t* d_leaf_hashes,
                    uint32_t row0 = 0, uint32_t n_rows = 0);

}  // namespace cm
