/*
 * cairom_hip.h — C ABI of libcairom_hip.so, the MI355X (gfx950) proving backend for Cairo-M.
 *
 * Drop-in boundary for ONE path of kkrt-labs/cairo-m: `prove_cairo_m::<Blake2sMerkleChannel>`
 * (crates/prover/src/prover.rs:23-147) and the Stwo backend-trait surface it drives
 * (ColumnOps / PolyOps / MerkleOps / GrindOps / QuotientOps / FriOps / AccumulationOps), plus the
 * Cairo-M component operations that are NOT reachable through those traits in the reference
 * (trace generation, LogUp interaction trace, constraint-quotient evaluation).
 *
 * Conventions
 *   - every function returns int32_t status: 0 = OK, non-zero = error; the message is available
 *     through cm_last_error() (thread-local).
 *   - cm_handle is an opaque 64-bit handle to library-owned device memory (a column = u32[n] of
 *     canonical M31 values; evaluations are stored in Stwo's BitReversedOrder).  Handles are freed
 *     by the caller with the matching cm_*_free.
 *   - QM31 / SecureField values cross the boundary as uint32_t[4] = to_m31_array()
 *     (reference layout: crates/prover/src/public_data.rs:146-149).
 *   - all functions take a cm_stream_t (0 = the library's default stream) and are safe to call
 *     concurrently from several host threads on different streams (the reference calls backend ops
 *     from rayon workers).
 *   - plain pointers and sizes only; no torch / C++ types.
 */
#ifndef CAIROM_HIP_H
#define CAIROM_HIP_H
#include <stddef.h>
#include <stdint.h>
/* Revision of this header.  4: cm_comm.struct_size at offset 0 (breaking for cm_comm users).  5: cm_shard_plan* take the PCS
 * config and column-array capacities; cm_set_device_tail, cm_tail_list.  6: cm_runner_segment grows by initial_heap /
 * n_initial_heap at its END (a revision-5 caller must be recompiled: the library reads the two fields);
 * cm_host_segment_set_initial_heap. */
#define CM_ABI_REVISION 6

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t cm_handle;
typedef uint64_t cm_stream_t;

/* ---- runtime ------------------------------------------------------------------------------- */
int32_t cm_init(int32_t device);
int32_t cm_shutdown(void);
/* Returns the calling host thread's cached device blocks to the driver (the per-thread pool otherwise keeps the high-water
 * mark of the largest segment proved: a 2^26-row segment leaves ~116 GiB cached) — including what the thread's last proof parked
 * for its successor (the FRI phase and quotient columns of the deferred teardown: round 6; cm_shutdown and the pool's own
 * out-of-memory retry release them too). */
int32_t cm_pool_trim(void);
/* Free / total bytes of the library device's HBM (hipMemGetInfo): segment sizing for the 288 GB of an MI355X. */
int32_t cm_device_mem_info(uint64_t* free_bytes, uint64_t* total_bytes);
int32_t cm_last_error(char* buf, size_t buf_len);
int32_t cm_stream_create(cm_stream_t* out);
int32_t cm_stream_destroy(cm_stream_t s);
int32_t cm_stream_sync(cm_stream_t s);
/* CPU placement of proving threads.  A proof is ~450 kernel launches and ~10 host round trips; on a two-socket host a
 * thread on the socket away from the GPU pays the inter-socket fabric on each of them (0.1-0.5 ms per proof).
 *   mode 1 (default; env CM_CPU_AFFINITY=1): SCOPED — during cm_prove_device / cm_prove_segment / cm_prove_sharded the
 *          calling thread's affinity is narrowed to the GPU's sysfs local_cpulist (intersected with the mask it
 *          already had) and the caller's own mask is restored before the call returns, so nothing leaks into the
 *          host application, into threads it creates later or into child processes.  The library's own
 *          cm_prove_many worker threads stay on the GPU's node.
 *   mode 2 (env CM_CPU_AFFINITY=2): STICKY — a thread that calls cm_prove* is narrowed the same way ONCE and stays there: a
 *          dedicated proving thread saves the three affinity system calls of every proof (30-50 us on the hosts this was
 *          measured on; bench.py's worker threads use it and say so in their line).
 *   mode 0 (env CM_CPU_AFFINITY=0 or CM_NO_CPU_AFFINITY=1): the library never calls sched_setaffinity.
 * cm_get_cpu_affinity returns the mode in force. */
int32_t cm_set_cpu_affinity(int32_t mode);
int32_t cm_get_cpu_affinity(void);
/* Framing switches for the Stwo-side conventions that no in-tree reference vector settles (Stwo @ ab57a1c is an empty
 * submodule of the reference, .gitmodules:1-3; crates/prover/tests/prover.rs only asserts verify(..).is_ok()).  spec =
 * comma-separated name=value pairs, "" / "default" = all defaults; process-wide; env CM_FRAMING gives the initial value:
 *   mix_u64       = raw (default: one raw compression F(digest, [lo, hi, 0..]) — what SimdBackend::grind searches over,
 *                   prover.rs:90 / verifier.rs:55-58)  |  u32s (Blake2s256(digest || lo || hi) = mix_u32s(&[lo, hi]))
 *   hash_node     = raw (default: compression chain from the zero state, t = f = 0)  |  rfc (RFC 7693 Blake2s-256 of
 *                   left || right || le32(values))                       — Blake2sMerkleHasher::hash_node / commit_on_layer
 *   sample_batch  = insertion (default)  |  sorted (by point)           — ColumnSampleBatch::new_vec
 *   pcs_mix       = bql (default: pow_bits, log_blowup, n_queries, log_last_layer)  |  blq   — PcsConfig::mix_into, prover.rs:36
 * A reference-produced transcript (integration/prover-hip/tests/golden_dump.rs -> tests/golden/ref_*.json, compared step by
 * step by tests/test_ref_golden.py) tells which value is right; prover and verifier must run under the same setting.
 * cm_get_framing writes the setting in force ("mix_u64=raw,hash_node=raw,...") and returns its length (-1: the environment
 * variable CM_FRAMING is malformed — a hard error for every entry point until cm_set_framing replaces it).
 * cm_set_framing is refused (status 1) while a proof or a verification is running: one proof = one framing. */
int32_t cm_set_framing(const char* spec);
int32_t cm_get_framing(char* buf, size_t buf_len);
/* Transcript log: on = every proof records one entry per Fiat-Shamir call the reference prover makes (Stwo Channel::{mix_u32s,
 * mix_felts, mix_u64, draw_felt, draw_felts, draw_random_bytes} + MerkleChannel::mix_root, in the order of prover.rs:36-131)
 * with the channel digest after the call.  cm_proof_transcript returns them as JSON (buffer owned by the proof):
 * [{"op": "mix_u64", "digest": "<64 hex>", "n_words": 2, "words": [first <= 16 words mixed / drawn]}, ...]. */
int32_t cm_set_transcript_log(int32_t on);

/* ---- Column<T> / ColumnOps (Stwo core::backend::{Column, ColumnOps}; used through
 *      `ComponentTrace::to_evals`, crates/prover/src/components/mod.rs:168-177) ----------------- */
int32_t cm_col_alloc(uint64_t n_u32, cm_handle* out);
int32_t cm_col_free(cm_handle h);
int32_t cm_col_h2d(cm_handle h, const uint32_t* src, uint64_t n_u32, cm_stream_t s);
int32_t cm_col_d2h(cm_handle h, uint32_t* dst, uint64_t n_u32, cm_stream_t s);
/* Column::at / Column::set (element ranges, offsets in u32 words) and Column::clone (device to device) */
int32_t cm_col_read(cm_handle h, uint64_t offset_u32, uint32_t* dst, uint64_t n_u32, cm_stream_t s);
int32_t cm_col_write(cm_handle h, uint64_t offset_u32, const uint32_t* src, uint64_t n_u32, cm_stream_t s);
int32_t cm_col_copy(cm_handle dst, cm_handle src, uint64_t n_u32, cm_stream_t s);
/* ColumnOps::bit_reverse_column, in place, on n_cols columns of 2^log_n */
int32_t cm_bit_reverse(const cm_handle* cols, uint32_t n_cols, uint32_t log_n, cm_stream_t s);

/* ---- PolyOps (Stwo core::poly::circle::PolyOps) --------------------------------------------- */
/* PolyOps::precompute_twiddles for CanonicCoset(log_size).circle_domain().half_coset
 * (reference: crates/prover/src/prover.rs:56-60). */
int32_t cm_twiddles_precompute(uint32_t log_size, cm_handle* tw_out);
int32_t cm_twiddles_free(cm_handle tw);
/* PolyOps::interpolate_columns (reference: tree_builder.extend_evals, prover.rs:72, 81, 101):
 * bit-reversed evaluations on CanonicCoset(log_n).circle_domain() -> coefficients, in place. */
int32_t cm_interpolate(const cm_handle* cols, uint32_t n_cols, uint32_t log_n, cm_handle tw, cm_stream_t s);
/* PolyOps::evaluate / evaluate_polynomials (reference: tree_builder.commit, prover.rs:73, 82, 102):
 * coefficients of 2^log_n -> bit-reversed evaluations on CanonicCoset(log_out).circle_domain(). */
int32_t cm_evaluate(const cm_handle* coeffs, uint32_t n_cols, uint32_t log_n, uint32_t log_out, cm_handle tw,
                    const cm_handle* out, cm_stream_t s);
/* tree_builder.extend_evals at log_blowup_factor 1 (prover.rs:71-73, 80-82, 100-102): interpolate_columns followed by
 * evaluate on CanonicCoset(log_n + 1) in one call — evals (2^log_n each, bit-reversed) -> coeffs (2^log_n) and lde (2^(log_n+1)).
 * Results equal cm_interpolate + cm_evaluate word for word; for 2^18..2^21 rows the last inverse pass and the first forward
 * pass are one sweep over HBM.  evals[i] may equal coeffs[i] (in place). */
int32_t cm_interpolate_extend(const cm_handle* evals, const cm_handle* coeffs, const cm_handle* lde, uint32_t n_cols, uint32_t log_n,
                              cm_handle tw, cm_stream_t s);
/* PolyOps::eval_at_point for n_cols polynomials at one QM31 circle point (x[4], y[4]);
 * out = n_cols * 4 u32 (host). */
int32_t cm_eval_at_point(const cm_handle* coeffs, uint32_t n_cols, uint32_t log_n, const uint32_t pt_xy[8],
                         uint32_t* out, cm_stream_t s);

/* ---- MerkleOps<Blake2sMerkleHasher>::commit_on_layer (reached from tree_builder.commit) ----- */
/* out_hashes: handle of 8 * 2^log_size u32.  prev_layer = 0 for the largest layer. */
int32_t cm_merkle_commit_layer(uint32_t log_size, cm_handle prev_layer, const cm_handle* cols, uint32_t n_cols,
                               cm_handle out_hashes, cm_stream_t s);
/* Whole mixed-degree tree (Stwo MerkleProver::commit): columns in commitment order with their
 * log sizes; writes the 32-byte root. */
int32_t cm_merkle_commit(const cm_handle* cols, const uint32_t* col_logs, uint32_t n_cols, uint8_t root[32],
                         cm_stream_t s);

/* ---- GrindOps<Blake2sChannel>::grind (reference: prover.rs:90) ------------------------------- */
int32_t cm_grind(const uint8_t digest[32], uint32_t pow_bits, uint64_t* nonce_out, cm_stream_t s);

/* ---- FieldOps::batch_inverse ---------------------------------------------------------------- */
int32_t cm_batch_inverse_m31(cm_handle in, cm_handle out, uint64_t n, cm_stream_t s);
int32_t cm_batch_inverse_qm31(const cm_handle in[4], const cm_handle out[4], uint64_t n, cm_stream_t s);

/* ---- FriOps (Stwo core::fri::FriOps), SecureColumnByCoords = 4 coordinate handles ------------ */
/* dst (line evaluation, 2^(log_n-1)) = dst * alpha^2 + fold(src circle evaluation of 2^log_n) */
int32_t cm_fri_fold_circle_into_line(const cm_handle dst[4], const cm_handle src[4], const uint32_t alpha[4],
                                     uint32_t log_n, cm_handle tw, cm_stream_t s);
/* out (2^(log_n-1)) = fold_line(in (2^log_n), alpha) */
int32_t cm_fri_fold_line(const cm_handle in[4], const uint32_t alpha[4], uint32_t log_n, cm_handle tw,
                         const cm_handle out[4], cm_stream_t s);
/* A FRI layer and the leaf layer of its commitment in one pass (what FriProver::commit_inner_layers does per layer: fold, then
 * MerkleProver::commit over the folded SecureColumn — prover.rs:131 -> Stwo fri.rs).  Sources, each with its challenge:
 *   in (4 handles, 2^log_n line evaluations) + alpha              -> out = fold_line(in, alpha)
 *   circle (4 handles, 2^log_n circle evaluations) + alpha_circle -> FriOps::fold_circle_into_line of the quotient columns of
 *   that size: out = out * alpha_circle^2 + fold_circle(circle, alpha_circle); alone (in == NULL, alpha == NULL) it folds into a
 *   blank layer: out = fold_circle(circle, alpha_circle).
 * out: 4 handles of 2^(log_n-1).  leaf_hashes: 8 words per row of `out` = MerkleOps::commit_on_layer(log_n - 1, None, out). */
int32_t cm_fri_fold_line_leaves(const cm_handle* in, const cm_handle* circle, const uint32_t* alpha, const uint32_t* alpha_circle,
                                uint32_t log_n, cm_handle tw, const cm_handle out[4], cm_handle leaf_hashes, cm_stream_t s);

/* ---- QuotientOps::accumulate_quotients ------------------------------------------------------ */
/* One call per distinct LDE log size.  cols: the n_cols committed LDE columns of that size.
 * Sample batches (Stwo ColumnSampleBatch): batch b has point pts[b] (x[4],y[4]) and the entries
 * batch_cols[batch_off[b] .. batch_off[b+1]) = (column index, sampled value[4]).
 * out: 4 coordinate columns of 2^log_size. */
typedef struct {
  uint32_t n_batches;
  const uint32_t* points;       /* n_batches * 8 */
  const uint32_t* batch_off;    /* n_batches + 1 */
  const uint32_t* col_index;    /* total entries */
  const uint32_t* values;       /* total entries * 4 */
} cm_sample_batches;
int32_t cm_accumulate_quotients(uint32_t log_size, const cm_handle* cols, uint32_t n_cols,
                                const cm_sample_batches* batches, const uint32_t random_coeff[4],
                                const cm_handle out[4], cm_handle tw, cm_stream_t s);

/* ---- Cairo-M prover input (mirror of crates/prover/src/adapter/mod.rs:27-83 ProverInput) ----- */
#define CM_N_OPCODE_COMPONENTS 26
#define CM_N_COMPONENTS 34

/* ExecutionBundle (crates/prover/src/adapter/memory.rs:95-124) flattened the way
 * PackedExecutionBundle does (crates/prover/src/utils/execution_bundle.rs:12-31). */
typedef struct {
  uint32_t pc, fp, clock, inst_prev_clock;
  uint32_t inst[6];      /* instruction words incl. opcode, zero padded */
  uint32_t span_start;   /* AccessSpan.start into data_accesses */
  uint32_t span_len;     /* AccessSpan.len */
} cm_bundle;
/* DataAccess (adapter/memory.rs:57-67) */
typedef struct {
  uint32_t address, prev_clock, prev_value, value;
} cm_data_access;
/* one boundary memory cell: (addr, value[4], clock, multiplicity) (adapter/memory.rs:186-193) */
typedef struct {
  uint32_t address, value[4], clock, multiplicity;
} cm_memory_cell;
/* clock_update_data entry (adapter/memory.rs:192) */
typedef struct {
  uint32_t address, prev_clock, value[4];
} cm_clock_update;
/* NodeData (adapter/merkle.rs:92-104) */
typedef struct {
  uint32_t index, depth, left_value, right_value, parent_value, left_mult, right_mult, parent_mult;
} cm_merkle_node;

typedef struct {
  /* Instructions (adapter/mod.rs:69-83) */
  uint32_t initial_pc, initial_fp, final_pc, final_fp;
  /* bundles grouped by opcode COMPONENT, in the macro order of
   * crates/prover/src/components/opcodes/mod.rs:223-268 */
  const cm_bundle* bundles[CM_N_OPCODE_COMPONENTS];
  uint64_t n_bundles[CM_N_OPCODE_COMPONENTS];
  const cm_data_access* data_accesses;
  uint64_t n_data_accesses;
  /* Memory (adapter/memory.rs:186-193): rows are given in the order the memory component must
   * emit them (the reference iterates HashMaps; the caller fixes the order — SURVEY F3). */
  const cm_memory_cell* initial_memory;
  uint64_t n_initial_memory;
  const cm_memory_cell* final_memory;
  uint64_t n_final_memory;
  const cm_clock_update* clock_updates;
  uint64_t n_clock_updates;
  /* MerkleTrees (adapter/mod.rs:47-58) */
  const cm_merkle_node* initial_tree;
  uint64_t n_initial_tree;
  const cm_merkle_node* final_tree;
  uint64_t n_final_tree;
  uint32_t initial_root, final_root;
  /* PublicAddressRanges (crates/common/src/program.rs:111-122): [start, end) */
  uint32_t program_range[2], input_range[2], output_range[2];
} cm_prover_input;

/* ---- host-side input pipeline (no GPU needed) ------------------------------------------------------
 * Synthetic VM + adapter: runs a CASM program (instruction = 1..6 M31 words, opcode first; encodings of
 * crates/common/src/instruction.rs:314-577) from `entry_pc` with the runner's calling convention
 * (crates/runner/src/vm/mod.rs:249-281), cuts segments every max_steps like vm/mod.rs:158-240, and
 * converts segment `segment_index` with `import_from_runner_output`
 * (crates/prover/src/adapter/mod.rs:251-266).  Memory rows are emitted in ascending address order. */
typedef struct cm_host_input cm_host_input;
int32_t cm_vm_run(const uint32_t* instr_words, const uint32_t* instr_lens, uint32_t n_instr, uint32_t entry_pc,
                  const uint32_t* args, uint32_t n_args, uint32_t n_returns, uint64_t max_steps,
                  uint32_t segment_index, cm_host_input** out, uint32_t* n_segments_out);
/* The hand-assembled fibonacci_loop of SURVEY §8d: 10*n + 12 steps. */
int32_t cm_synth_fibonacci(uint32_t n, uint64_t max_steps, uint32_t segment_index, cm_host_input** out);
const cm_prover_input* cm_host_input_view(const cm_host_input* h);
uint64_t cm_host_input_steps(const cm_host_input* h);
int32_t cm_host_input_free(cm_host_input* h);
/* Test hook: replay `Memory::push` (crates/prover/src/adapter/memory.rs:470-535) over a script of accesses. */
int32_t cm_adapter_memory_script(const uint32_t* preload, uint32_t n_preload, const uint32_t* script, uint32_t n,
                                 uint32_t* results, uint32_t* n_clock_updates, uint32_t* clock_updates_out, uint32_t cu_cap,
                                 const uint32_t* query_addrs, uint32_t n_query, uint32_t* state_out);
/* Test hook: build_partial_merkle_tree (crates/prover/src/adapter/merkle.rs:183-295) over n cells (addr, v0..v3); nodes_out
 * = up to cap cm_merkle_node records, *n_nodes = their total count, *root = the root (0 for an empty memory). */
int32_t cm_adapter_partial_tree(const uint32_t* cells, uint32_t n, int32_t initial, const uint32_t ranges[6], uint32_t* nodes_out,
                                uint64_t cap, uint64_t* n_nodes, uint32_t* root);
/* Poseidon2-M31 t=16 permutation in place (reference KAT: crates/prover/tests/poseidon2.rs:14-34). */
int32_t cm_poseidon2_permute(uint32_t state[16]);

/* PcsConfig (crates/prover/src/prover_config.rs:13-20) */
typedef struct {
  uint32_t pow_bits;
  uint32_t log_blowup_factor;
  uint32_t log_last_layer_degree_bound;
  uint32_t n_queries;
} cm_pcs_config;

/* Proof object: opaque; serialised with cm_proof_json (serde layout of `Proof<Blake2sHash>`,
 * crates/prover/src/lib.rs:61-73 + main.rs:86-91). */
typedef struct cm_proof cm_proof;

/* prove_cairo_m::<Blake2sMerkleChannel> (crates/prover/src/prover.rs:23-147).
 * config == NULL selects REGULAR_96_BITS. */
int32_t cm_prove_segment(const cm_prover_input* input, const cm_pcs_config* config, cm_proof** out);
int32_t cm_proof_free(cm_proof* p);
/* Same path with the input already resident in HBM (what bench.py times): upload once, prove many. */
typedef struct cm_device_input cm_device_input;
int32_t cm_input_upload(const cm_prover_input* input, cm_device_input** out);
int32_t cm_input_free(cm_device_input* h);
int32_t cm_prove_device(const cm_device_input* input, const cm_pcs_config* config, cm_proof** out);
/* verify_cairo_m (crates/prover/src/verifier.rs:17-95 + Stwo verify): host code, works without a GPU.
 * 0 = the proof is accepted; otherwise status 11 and cm_last_error() names the failed check
 * (InvalidLogupSum, OodsNotMatching, Merkle(...), Fri(...), ProofOfWork, ...).  `words` = cm_proof_words format.
 * `expected` is the VERIFIER's PcsConfig (verify_cairo_m's `pcs_config: Option<PcsConfig>` argument, verifier.rs:19; NULL =
 * REGULAR_96_BITS {16, 1, 0, 80}): it is what goes into the transcript and what bounds the PoW / query count; a proof
 * made under any other config is rejected with InvalidStructure(config) — the prover never chooses the security level. */
int32_t cm_verify_proof(const cm_proof* p, const cm_pcs_config* expected);
int32_t cm_verify_proof_words(const uint32_t* words, uint64_t n_words, const cm_pcs_config* expected);
/* ---- intra-proof sharding (BASELINE configs[3]): `world` ranks (one process per GPU) prove ONE segment.
 * Whole components are assigned to ranks (cm_shard_plan: longest-processing-time bin packing by columns x rows) for trace
 * generation, LogUp, IFFT / LDE, constraint evaluation and OODS sampling; an all-to-all turns the committed LDE columns
 * into row-range ownership, so every rank hashes the Merkle subtree of its rows (sub-roots are all-gathered, the top
 * log2(world) levels are hashed by everyone) and accumulates the DEEP quotients of its rows; partial composition
 * accumulators are reduced across ranks.  FRI is committed by row range above 2^16 rows (folds are pair-local; a layer's tree
 * is the rank's subtree + a 32-byte all-gather) and finished on every rank below; the transforms of the preprocessed and the
 * composition tree are computed by every rank (replicated).  Every rank returns the same proof, bit-identical to cm_prove_device's.
 * The library does no communication itself: the host side (torch.distributed / RCCL in bench.py, anything else
 * elsewhere) provides two blocking collectives over two DEVICE staging buffers it owns.  Layouts are rank-major and
 * contiguous: all_to_all_v sends send_words[d] words to rank d from send_buf (blocks in rank order) and receives
 * recv_words[s] words from rank s into recv_buf; all_gather sends send_buf[0, words_per_rank) and receives world blocks. */
typedef struct cm_comm {
  uint32_t struct_size;              /* sizeof(cm_comm) as the CALLER compiled it: the fields behind all_gather are optional and are
                                      * only read when this size covers them (a binding that stops at all_gather works); smaller
                                      * than that = status 1.  ABI note: this field was inserted at offset 0 in header revision 4
                                      * (CM_ABI_REVISION), which moved every other field — a caller compiled against a revision-3
                                      * header is NOT compatible and is rejected with status 1 (its `rank` is read as a size) */
  uint32_t rank, world;              /* world: a power of two, 1..8 */
  void* ctx;
  uint32_t* send_buf;
  uint32_t* recv_buf;
  uint64_t buf_words;                /* capacity of each staging buffer (cm_shard_plan reports what a proof needs) */
  int32_t (*all_to_all_v)(void* ctx, const uint64_t* send_words, const uint64_t* recv_words);
  int32_t (*all_gather)(void* ctx, uint64_t words_per_rank);
  /* Optional (zero / NULL = the blocking form above).  CM_COMM_STREAM_ORDERED: the callbacks only ENQUEUE the exchange on the
   * stream given through set_stream and return at once; the prover then neither drains its stream before a collective nor waits
   * after it — everything stays ordered on that stream.  The prover calls set_stream with its own stream when a proof starts and
   * MAY call it again between two collectives (round 6: the claimed sums' all-gather follows the LogUp tail onto a side stream and
   * the communicator is handed back to the main stream right behind it): a callback always uses the stream of the latest call. */
  uint32_t flags;
  int32_t (*set_stream)(void* ctx, cm_stream_t stream);
  /* Optional.  Called when THIS rank fails inside cm_prove_sharded, before the error is returned: the other ranks are about to
   * block in their next collective, and only the communicator can release them (the in-library RCCL communicator calls
   * ncclCommAbort).  NULL: the host relies on its communicator's own timeout. */
  void (*abort)(void* ctx);
} cm_comm;
#define CM_COMM_STREAM_ORDERED 1u
/* host code (no GPU): which rank owns which component, and the staging capacity (words) a sharded proof of `input` needs UNDER
 * `config` (NULL = REGULAR_96_BITS) — the plan and the bound are the ones cm_prove_sharded runs with that config (components are
 * split only at log_blowup_factor 1; the exchanges grow with the blowup).  Revision 5: the `config` argument is new. */
int32_t cm_shard_plan(const cm_prover_input* input, const cm_pcs_config* config, uint32_t world, int32_t owner[CM_N_COMPONENTS],
                      uint64_t* staging_words);
/* owner[c] = -1: component c is SPLIT over all ranks — a large opcode component (more than an eighth of a rank's fair share of
 * the cells, at least 2^12 rows) is generated, looked up and constrained by ROW RANGE on every rank, and its columns are
 * transformed by the ranks the plan gives them to one by one (the four cumulative-sum columns of its LogUp stay together).
 * cm_shard_plan_columns reports that column-level plan: the owner of every column of trees 1 / 2 in commitment order and the
 * cells every rank transforms (load_cells[r], r < world).  *n_trace_cols / *n_interaction_cols: the CAPACITY of the array on
 * entry, the column count on return; an array that is too small is status 1 (nothing is written past it).  Pass NULL arrays to
 * learn the counts. */
int32_t cm_shard_plan_columns(const cm_prover_input* input, const cm_pcs_config* config, uint32_t world, int32_t* trace_col_owner,
                              uint32_t* n_trace_cols, int32_t* interaction_col_owner, uint32_t* n_interaction_cols,
                              uint64_t load_cells[8]);
int32_t cm_prove_sharded(const cm_device_input* input, const cm_pcs_config* config, const cm_comm* comm, cm_proof** out);
/* In-library cm_comm on RCCL (xGMI inside a node), stream-ordered: no host synchronisation around an exchange, nothing but
 * the library in the data path.  Rank 0 makes the 128-byte id (cm_rccl_unique_id) and the launcher hands it to every rank;
 * every rank creates its communicator with the staging capacity cm_shard_plan reports and passes cm_rccl_comm_view() to
 * cm_prove_sharded.  librccl.so is loaded on first use (dlopen; a copy the process already mapped is reused). */
typedef struct cm_rccl_comm cm_rccl_comm;
int32_t cm_rccl_unique_id(uint8_t id_out[128]);
int32_t cm_rccl_comm_create(const uint8_t id[128], uint32_t rank, uint32_t world, uint64_t staging_words, cm_rccl_comm** out);
const cm_comm* cm_rccl_comm_view(const cm_rccl_comm* c);
int32_t cm_rccl_comm_destroy(cm_rccl_comm* c);
/* Segment pipeline: prove n independent segments with up to `inflight` (1..8) proofs in flight on the
 * GPU (persistent worker threads inside the library, one stream set / device pool each).  outs[i] = proof of
 * inputs[i]; on error the first failure is returned and the proofs already built stay in outs (free them). */
int32_t cm_prove_many(const cm_device_input* const* inputs, uint32_t n, const cm_pcs_config* config, uint32_t inflight,
                      cm_proof** outs);
/* Streaming ingest: cm_prove_many from HOST inputs — the entry point of the reference takes a host
 * `&mut ProverInput` (crates/prover/src/prover.rs:23-29).  The calling thread uploads input i + 1 (cm_prove_many_host) or runs
 * the device adapter on runner segment i + 1 (cm_prove_many_segments = import_from_runner_output, adapter/mod.rs:97-193) on
 * its own stream while up to `inflight` (1..8) library threads prove the inputs before it: the PCIe copies hide under the
 * proofs.  At most inflight + 1 inputs are resident in HBM at any time (inflight + 2 for cm_prove_many_segments, whose adapter —
 * uploads, sorts and five host round trips per segment — runs on a second library thread as well).  outs[i] = proof of item i (bit-identical to
 * cm_prove_segment / cm_adapt_segment_device + cm_prove_device of the same item); error contract of cm_prove_many. */
int32_t cm_prove_many_host(const cm_prover_input* const* inputs, uint32_t n, const cm_pcs_config* config, uint32_t inflight,
                           cm_proof** outs);
/* Preprocessed-tree cache: tree 0 commits constant tables (crates/prover/src/preprocessed/mod.rs:75-82;
 * verifier.rs:38 notes its root is a known constant).  Off by default (every proof recomputes it, as prover.rs:70-73
 * does); on = each host thread keeps the committed tree (coefficients, LDE, Merkle layers) between proofs of one
 * PCS config.  Env CM_PREPROCESSED_CACHE=1 sets the initial value.  Proof bytes are identical either way. */
int32_t cm_set_preprocessed_cache(int32_t on);
/* Twiddle tables: the reference recomputes them in every prove_cairo_m (prover.rs:56-60), and so does every proof here by
 * default (pool memory, on a side stream next to trace generation).  on = keep one table per domain size for the whole
 * process (env CM_TWIDDLE_CACHE=1 sets the initial value).  Off in every quoted number.  Proof bytes are identical. */
int32_t cm_set_twiddle_cache(int32_t on);
/* Tail of a proof — everything stwo `prove` (prover.rs:131) does behind the last FRI fold: the last layer's polynomial, the
 * proof of work, the query draws and the decommitment of every tree.  1 (default; env CM_DEVICE_TAIL=0 sets the initial value
 * to 0) = four launches enqueued behind the last fold, witnesses written to pinned host memory in proof order, the host
 * replays the transcript steps afterwards and refuses the proof on a mismatch (cairo_m_amd/csrc/tail_device.hpp);
 * 0 = the host drives each step (round trip per step, symbolic decommitment walk on the host).  Proof bytes are identical. */
int32_t cm_set_device_tail(int32_t on);
/* (revision 6) Measurement switches, flipped inside one process so that two forms can be timed alternately on the same box
 * (tools/ab_switch.py: paired blocks of lone proofs); every form produces the same proof bytes.  Each has an environment
 * variable of the same name in capitals behind CM_ for its initial value.  key (default):
 *   "oods_poll" (1)          the sampled values are watched arriving in pinned host memory; 0 = event / stream synchronisation
 *   "oods_host_write" (1)    (with oods_poll) the kernel that reduces them writes the pinned words; 0 = copy commands, watched
 *   "oods_split" (780)       per mille of the sampled values evaluated, copied and hashed first (1000 = one chunk)
 *   "stage_copy_kernel" (1)  small host -> device uploads are a kernel reading the pinned staging ring; 0 = hipMemcpyAsync (the SDMA
 *                            engine above a few KB)
 *   "stage_lazy_events" (1)  the staging ring's event of the thread's main stream is recorded when the ring wraps; 0 = per upload
 *   "defer_teardown" (1)     the pool blocks of a proof's FRI phase / quotient plan are given back by the calling thread's NEXT
 *                            proof while it waits for tree 1 (or when the thread ends); 0 = before cm_prove* returns
 *   "tail_flags" (1)         the host follows the device-side tail by two header words the kernels set behind their pinned writes;
 *                            0 = two events recorded between the tail's launches
 *   "flag_join" / "flag_fork" (1)  fork regions joined / forked by flag words polled by one-wave kernels; 0 = HIP events
 *   "commit_prep_early" (1)  the interaction tree's launch plan is prepared under the LogUp kernels; 0 = behind them
 *   "trace_hist_fuse" (1)    trace cells and lookup histogram of a large opcode component in one launch; 0 = two
 *   "logup_defer" (1)        the LogUp tail (claimed sums, prefix scans) on a side stream next to the first transforms; 0 = in front
 *   "fri_top_fuse" (1)       fold + transcript step inside the tree-top launch of the FRI layers <= 2^16; 0 = three launches
 * and the policy choices of earlier rounds, numeric where the environment variable was: "fork_main" (1), "merkle_npw" (-1 = by
 * layer size, 0 = k_merkle_layer, 1..8 chunks per wave), "fork_width" (0 = all side streams), "pp_side" (1), "tree0_prio" (-1 high
 * priority stream, 0 fork side stream, 1 low), "tree1_first" (1), "logup_width" (4), "quot_rows" (2), "fri_fold_leaf" (1),
 * "fft_fused" (1), "commit_pipe" (1), "fft_chunk_mb" (0 = off), "pace" (-1 = by load), "pace_early" (1).
 * Round 6: "cons_wide_first" (1: a wide component of few rows — poseidon2's 443 columns — is the first launch of the constraint
 * region), "cons_plan" (side-stream plan of that region as eight octal digits, default 01237456), "logup_small_stream" (-1 = stream
 * `logup_width`; 0..7 picks the side stream of the batched small LogUp launch), "quot_leaf" (1: the DEEP-quotient kernel of the largest
 * size group also writes the leaf hashes of the FRI first-layer tree), "shard_fri_stream" (1: cm_prove_sharded keeps its row-sharded
 * FRI layers on the stream — tree top and transcript step on the device, one host replay per proof), "shard_halo" (1: the previous-row
 * neighbours of a split component's cumulative-sum columns come from the two neighbouring row ranges instead of an all-gather), "tree0_guest"
 * (0; 1 = the preprocessed columns are transformed inside the trace tree's size-group launches: measured neutral), "tw_batch" (8), "fft_half_occ" (0; bit 0 / 1 =
 * one 2^14-tile / four 2^12-tile transform blocks per CU instead of two / eight: measured slower, kept for A/B), "shard_tree_stream" (1:
 * cm_prove_sharded keeps the transcript steps behind its four commitment trees on the stream — tree top on the device, the single-GPU
 * prover's step kernels, host replay at the two waits that are left before FRI; 0 = a host round trip per root), "shard_fri_stop_log"
 * (16: FRI layers of at most 2^v rows are gathered and finished on every rank; 99 = FRI not sharded; also CM_SHARD_FRI_STOP_LOG).
 * Test hook: "tail_grind_cap" (0 = off; v > 0 stops the device tail's proof-of-work search after 2^(v-1) nonces, so that the
 * host-driven fallback behind a missed nonce — probability e^-16 in production — can be exercised; same proof bytes).
 * status 1 for an unknown key or a value outside the key's range.  Flip a switch only while no proof is running in the process: a
 * proof reads some of them more than once (a fork region decides at its fork AND at every side stream how it synchronises). */
int32_t cm_set_tuning(const char* key, int32_t value);
/* The node lists the device-side tail gathers by (host code, no GPU: the mirror the prover checks the device against; the CPU
 * tests compare it with a restatement of MerkleProver::decommit).  positions: the sorted, de-duplicated query positions on the
 * domain of 2^log_domain points; qmask: bit l = the first FRI tree carries columns of 2^l rows; list: 0 = U[k] (queried nodes
 * of the layer of 2^(log_domain - k) nodes), 1 = W[k] (their unqueried siblings), 2 = F[k] (hashes of the layer below that the
 * first FRI tree's walk asks for at that layer).  Writes min(*n_out, cap) entries; *n_out = the list's length. */
int32_t cm_tail_list(const uint32_t* positions, uint32_t n_positions, uint32_t log_domain, uint32_t qmask, uint32_t list, uint32_t k,
                     uint32_t* out, uint32_t cap, uint32_t* n_out);
/* Proof object from its flat word stream (host code, no GPU): re-serialise with cm_proof_json / verify with cm_verify_proof. */
int32_t cm_proof_from_words(const uint32_t* words, uint64_t n_words, cm_proof** out);
/* Flat u32 serialisation of the proof (format: cairo_m_amd/csrc/proof.hpp), used by the parity tests. */
int32_t cm_proof_words(const cm_proof* p, const uint32_t** words_out, uint64_t* n_out);
/* JSON text of the proof; *len_out = length without the terminating NUL; the buffer is owned by
 * the proof object. */
int32_t cm_proof_json(const cm_proof* p, const char** json_out, size_t* len_out);
int32_t cm_proof_transcript(const cm_proof* p, const char** json_out, size_t* len_out);
/* The four commitment roots (trees 0..3), 32 bytes each. */
int32_t cm_proof_commitments(const cm_proof* p, uint8_t roots[4][32]);
/* ---- device-side adapter -----------------------------------------------------------------
 * One runner segment in the runner's own terms: the VM trace, the memory access log and the memory at
 * segment start (crates/runner/src/vm/mod.rs:306-375, crates/common/src/execution.rs:28-66).
 * cm_adapt_segment_device = import_from_runner_output (crates/prover/src/adapter/mod.rs:97-193) with the
 * per-step work (previous-access tracking, clock updates, opcode bucketing) on the GPU; the result is the
 * device-resident ProverInput cm_prove_device consumes.  Same row order as the host adapter (cm_vm_run). */
typedef struct {
  const uint32_t* trace;           /* (pc, fp) pairs: n_trace = steps + 1 entries */
  uint64_t n_trace;
  const uint32_t* memory_trace;    /* (address, v0, v1, v2, v3): 5 words per logged access, in execution order */
  uint64_t n_memory_trace;
  const uint32_t* initial_memory;  /* 4 words per cell: addresses 0 .. n_initial_memory-1 at segment start */
  uint64_t n_initial_memory;
  uint32_t program_range[2], input_range[2], output_range[2];
  /* (revision 6) the HEAP at segment start, 4 words per cell: index i = the cell at MAX_ADDRESS - i = 2^28 - 1 - i — the second
   * half of the reference's Segment::initial_memory (crates/runner/src/vm/mod.rs:205-221: `heap[i]` maps to MAX_ADDRESS - i;
   * the allocator of `new T[n]` grows it downwards).  Empty for a program's first segment; NULL / 0 when there is none. */
  const uint32_t* initial_heap;
  uint64_t n_initial_heap;
} cm_runner_segment;
int32_t cm_adapt_segment_device(const cm_runner_segment* seg, cm_device_input** out);
/* streaming ingest from runner segments: see cm_prove_many_host */
int32_t cm_prove_many_segments(const cm_runner_segment* const* segments, uint32_t n, const cm_pcs_config* config, uint32_t inflight,
                               cm_proof** outs);
/* Copy a device-resident ProverInput back (tests: device adapter vs host adapter). */
int32_t cm_device_input_download(const cm_device_input* in, cm_host_input** out);
/* The synthetic VM's raw output for one segment (what cm_vm_run feeds to the host adapter). */
typedef struct cm_host_segment cm_host_segment;
int32_t cm_vm_segment(const uint32_t* instr_words, const uint32_t* instr_lens, uint32_t n_instr, uint32_t entry_pc,
                      const uint32_t* args, uint32_t n_args, uint32_t n_returns, uint64_t max_steps,
                      uint32_t segment_index, cm_host_segment** out, uint32_t* n_segments_out);
int32_t cm_synth_fibonacci_segment(uint32_t n, uint64_t max_steps, uint32_t segment_index, cm_host_segment** out);
const cm_runner_segment* cm_host_segment_view(const cm_host_segment* h);
/* Runner artifact wire formats (crates/common/src/execution.rs:28-66, crates/prover/src/adapter/io.rs:38-80):
 * trace = (fp, pc) little-endian u32 pairs; memory trace = [u32 program_length header] + (address, v0..v3)
 * records.  out == NULL only reports *len.  The reference does not serialise the initial memory and the public
 * ranges (adapter/mod.rs:215-237): cm_segment_from_artifacts takes them from the caller
 * (ranges = program, input, output [start, end)). */
int32_t cm_segment_serialize_trace(const cm_runner_segment* s, uint8_t* out, uint64_t cap, uint64_t* len);
int32_t cm_segment_serialize_memory_trace(const cm_runner_segment* s, int32_t with_header, uint8_t* out, uint64_t cap, uint64_t* len);
int32_t cm_segment_from_artifacts(const uint8_t* trace, uint64_t trace_len, const uint8_t* mem, uint64_t mem_len,
                                  int32_t mem_has_header, const uint32_t* initial_memory, uint64_t n_initial_memory,
                                  const uint32_t ranges[6], cm_host_segment** out);
/* (revision 6) the heap cells of a segment built from artifacts (copied; see cm_runner_segment.initial_heap) */
int32_t cm_host_segment_set_initial_heap(cm_host_segment* h, const uint32_t* initial_heap, uint64_t n_initial_heap);
int32_t cm_host_segment_free(cm_host_segment* h);
/* ---- per-component AIR ops --------------------------------------------------------------------
 * What the reference does per component through Rust generics that name SimdBackend — and therefore cannot be
 * reached through Stwo's Backend traits — on caller-owned device columns (cm_handle = u32[2^log_size]):
 *   cm_trace_write            = <component>::Claim::write_trace (e.g. opcodes/store_fp_imm.rs:147-296, memory.rs:93-195,
 *                               merkle.rs:92-201, clock_update.rs:77-166, poseidon2.rs:172-325)
 *   cm_histogram              = range_check_N / bitwise Claim::write_trace multiplicities
 *                               (preprocessed/range_check/range_check_macro.rs:62-112, preprocessed/bitwise.rs:72-157)
 *   cm_preprocessed_column    = PreProcessedTrace::gen_trace columns (preprocessed/mod.rs:36-38, bitwise.rs:283-319)
 *   cm_interaction_write      = <component>::InteractionClaim::write_interaction_trace + LogupTraceGenerator::finalize_last
 *   cm_constraints_accumulate = FrameworkComponent::evaluate_constraint_quotients_on_domain of <component>::Eval
 * Component ids: 0..25 = opcode components in the order of define_opcodes! (opcodes/mod.rs:223-268), then memory 26,
 * merkle 27, clock_update 28, poseidon2 29, range_check_8/16/20 30..32, bitwise 33.  The whole-segment prover runs the
 * same kernels. */
#define CM_N_RELATIONS 8
#define CM_MAX_RELATION_SIZE 16
#define CM_N_PREPROCESSED 7
/* Relations::draw (components/mod.rs:311-323): z and the powers alpha^0.. of every relation, QM31 as 4 words */
typedef struct {
  uint32_t z[CM_N_RELATIONS][4];
  uint32_t alpha_pow[CM_N_RELATIONS][CM_MAX_RELATION_SIZE][4];
} cm_relations;
int32_t cm_component_info(int32_t component, uint32_t* n_trace_cols, uint32_t* n_interaction_cols, uint32_t* n_constraints);
/* log size of the component's trace for this input: max(4, ceil_log2(rows)) — fixed 8/16/20/18 for the lookup tables */
int32_t cm_component_log_size(const cm_device_input* input, int32_t component, uint32_t* log_size);
/* components 0..29; cols = n_trace_cols columns of 2^log_size (padding rows included) */
int32_t cm_trace_write(const cm_device_input* input, int32_t component, const cm_handle* cols, cm_stream_t s);
/* adds the lookups of one opcode component's trace (padding rows included) to the four multiplicity columns
 * (2^8, 2^16, 2^20, 2^18 words, zeroed by the caller); an out-of-range value is status 1 */
int32_t cm_histogram(int32_t component, const cm_handle* trace_cols, uint32_t log_size, cm_handle rc8, cm_handle rc16,
                     cm_handle rc20, cm_handle bitwise, cm_stream_t s);
/* preprocessed column id 0..6 (bitwise op / a / b / result, range_check_8, _16, _20) on its trace domain */
int32_t cm_preprocessed_column(int32_t id, cm_handle col, cm_stream_t s);
/* trace_cols / preprocessed: trace-domain columns (preprocessed = all CM_N_PREPROCESSED columns);
 * out = n_interaction_cols columns; claimed_sum = the component's InteractionClaim */
int32_t cm_interaction_write(int32_t component, const cm_handle* trace_cols, const cm_handle* preprocessed, uint32_t log_size,
                             const cm_relations* relations, const cm_handle* out, uint32_t claimed_sum[4], cm_stream_t s);
/* *_lde: the component's columns evaluated on CanonicCoset(log_size + 1) (cm_interpolate + cm_evaluate), preprocessed_lde
 * likewise (column k on CanonicCoset(its log + 1)); coeff_powers = the random-coefficient powers of this component's
 * n_constraints constraints, 4 words each, in constraint order; acc += sum_k coeff_k * C_k / vanishing */
int32_t cm_constraints_accumulate(int32_t component, const cm_handle* trace_lde, const cm_handle* interaction_lde,
                                  const cm_handle* preprocessed_lde, uint32_t log_size, const cm_relations* relations,
                                  const uint32_t* coeff_powers, const uint32_t claimed_sum[4], const cm_handle acc[4],
                                  cm_stream_t s);
/* AccumulationOps::accumulate: dst[k][i] += src[k][i] (4 coordinate columns of n words); generate_secure_powers:
 * out[i] = felt^i for i < n (host array of 4 * n words).  Column::zeros = cm_col_alloc + cm_col_zero. */
int32_t cm_accumulate(const cm_handle dst[4], const cm_handle src[4], uint64_t n, cm_stream_t s);
int32_t cm_generate_secure_powers(const uint32_t felt[4], uint64_t n, uint32_t* out);
int32_t cm_col_zero(cm_handle h, uint64_t n_u32, cm_stream_t s);
/* FriOps::decompose: lambda = (sum of the first half - sum of the second half) / 2^log_n of a bit-reversed secure
 * evaluation; g = f - lambda on the first half, f + lambda on the second.  In place; lambda_out = 4 words. */
int32_t cm_fri_decompose(const cm_handle f[4], uint32_t log_n, uint32_t lambda_out[4], cm_stream_t s);
/* Optional HIP-event kernel timing on the launch stream (bench.py `roofline`). */
int32_t cm_kprof_enable(int32_t on);
int32_t cm_kprof_report(char* buf, size_t buf_len);
/* Restrict the timing to one kernel class (a key of cm_kprof_report); NULL or "" = every class. */
int32_t cm_kprof_filter(const char* name);
/* cells = sum over committed columns of trees 0,1,2 of 2^log_size. */
int32_t cm_proof_stats(const cm_proof* p, uint64_t* cells, uint64_t* steps, double* phase_ms, uint32_t n_phases);

#ifdef __cplusplus
}
#endif
#endif /* CAIROM_HIP_H */
