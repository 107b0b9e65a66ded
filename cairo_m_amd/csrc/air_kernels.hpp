// Host-visible declarations of the per-component AIR kernels (kernels_air.hip).
#pragma once
#include "engine.hpp"

namespace cm {

struct DevRelations;
struct HistPtrs;

struct ConstraintArgs {
  const uint32_t* const* tr;    // tree-1 LDE columns of the component (device array)
  const uint32_t* const* it;    // tree-2 LDE columns of the component
  const uint32_t* const* pp;    // tree-0 LDE columns by PreprocId
  const DevRelations* rels;
  const uint32_t* coeff;        // random-coefficient powers of this component's constraints (4 u32 each)
  uint32_t* const* acc;         // 4 accumulator columns of 2^(log_size+1)
  uint32_t log_size;            // trace log size
  int n_base;                   // number of add_constraint constraints
  uint32_t cumsum_shift[4];
  uint32_t denom_inv[2];        // 1 / coset_vanishing on the two cosets of the evaluation domain
  // Row range of the evaluation domain to cover (n_rows = 0: all 2^(log_size + 1) rows).  The sharded prover evaluates a
  // component that is split over the ranks on ITS rows only: tr / it then point `row0` words in front of the rank's slices (the
  // kernels index columns by the global row), the last four interaction columns (the cumulative sum, read at the previous
  // row as well) and pp are full columns.
  uint32_t row0 = 0, n_rows = 0;
  // (round 6, sharded prover) the four cumulative-sum columns of a split component at the PREVIOUS row of every row of the rank's range
  // (pointers `row0` words in front of the halo slices, like tr / it): the range's own neighbours instead of four whole columns.
  // null: the last four `it` columns are whole and are read at prev_row.
  const uint32_t* const* it_prev = nullptr;
};

void launch_opcode_trace(int cid, const void* bundles, uint32_t n, const void* acc, uint32_t log_size, uint32_t* const* d_cols,
                         hipStream_t st);
void launch_memory_trace(const void* init, uint32_t ni, const void* fin, uint32_t nf, uint32_t root_i, uint32_t root_f,
                         uint32_t log_size, uint32_t* const* d_cols, hipStream_t st);
void launch_merkle_trace(const void* init, uint32_t ni, const void* fin, uint32_t nf, uint32_t root_i, uint32_t root_f,
                         uint32_t log_size, uint32_t* const* d_cols, hipStream_t st);
void launch_clock_update_trace(const void* rows, uint32_t n, uint32_t log_size, uint32_t* const* d_cols, hipStream_t st);
void launch_poseidon2_trace(const void* init, uint32_t ni, const void* fin, uint32_t nf, uint32_t log_size,
                            uint32_t* const* d_cols, hipStream_t st);
void launch_hist(int cid, const uint32_t* const* d_cols, uint32_t log_size, const HistPtrs& h, hipStream_t st);
// both in one launch (the histogram from the row still in registers): the prover's path for the large components
void launch_opcode_trace_hist(int cid, const void* bundles, uint32_t n, const void* acc, uint32_t log_size, uint32_t* const* d_cols,
                              const HistPtrs& h, hipStream_t st);
void launch_logup(int cid, const uint32_t* const* d_cols, const uint32_t* const* d_pp, uint32_t log_size,
                  const DevRelations* d_rels, uint32_t* const* d_out, hipStream_t st);
void launch_constraints(int cid, const ConstraintArgs& a, hipStream_t st);
// Every small component of a phase in ONE launch (blockIdx.y = job): see k_logup_small in kernels_air.inc
struct SmallLogupJob { const uint32_t* const* cols; uint32_t* const* out; uint32_t log_size; int cid; };
constexpr uint32_t SMALL_COMPONENT_MAX_LOG = 8;
void launch_logup_small(const SmallLogupJob* d_jobs, uint32_t n_jobs, uint32_t max_log, const uint32_t* const* d_pp,
                        const DevRelations* d_rels, hipStream_t st);
// trace rows + histogram of the small opcode components (<= 256 rows each), one block per job
struct SmallTraceJob { const void* bundles; uint32_t n; uint32_t* const* cols; uint32_t log_size; int cid; };
void launch_trace_hist_small(const SmallTraceJob* d_jobs, uint32_t n_jobs, const void* acc, const HistPtrs& h, hipStream_t st);
void launch_constraints_small(const ConstraintArgs* d_jobs, const int* d_cids, uint32_t n_jobs, uint32_t max_log, hipStream_t st);
// LogupTraceGenerator::finalize_last for every component at once
struct LogupTailJob {
  uint32_t* col[4];     // last 4 interaction columns (trace domain, bit-reversed circle order)
  uint32_t log_size;
  uint32_t tmp_off, btot_off;  // filled by logup_finalize_all
  uint32_t id, pad;            // index of the job in the caller's list (claimed sums / shifts)
};
void logup_finalize_all(const std::vector<LogupTailJob>& jobs, uint32_t* d_sums, hipStream_t st, std::vector<DevBuf>* keep = nullptr);
void launch_preproc(int pp_id, uint32_t log_size, uint32_t* d_col, hipStream_t st);
void launch_preproc_all(uint32_t* const cols[], hipStream_t st);   // every preprocessed column (PREPROC_LOG sizes) in one launch
// d_acc[c][i] += sum_s slots[((s * 4 + c) << log_n) + i]   (c < 4, s < n_slots)
void sum_slots(uint32_t* const* d_acc, const uint32_t* d_slots, uint32_t n_slots, uint32_t log_n, hipStream_t st);
void add_columns(uint32_t* const* d_dst, const uint32_t* const* d_src, uint32_t ncols, uint32_t n, hipStream_t st);
// d_dst[c][i] += sum_k (i < 2^log[k] ? src[k][c][i] : 0): every smaller accumulator in one launch
struct AddColumnsSrc { uint32_t n; uint32_t log[28]; const uint32_t* const* src[28]; };
void add_columns_multi(uint32_t* const* d_dst, const AddColumnsSrc& s, uint32_t ncols, hipStream_t st);

}  // namespace cm
