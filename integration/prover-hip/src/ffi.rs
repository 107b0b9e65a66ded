//! Raw bindings of include/cairom_hip.h — EVERY exported function (tests/test_rust_shim.py fails on a header function without
//! an extern twin of the same arity; tools/gen_ffi_rs.py prints the block from the header).  Layouts are `#[repr(C)]` mirrors
//! of the C structs; every function returns a status (0 = OK) and leaves a message for `cm_last_error`.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

pub type cm_handle = u64;
pub type cm_stream_t = u64;

pub const CM_N_OPCODE_COMPONENTS: usize = 26;

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct cm_bundle {
    pub pc: u32,
    pub fp: u32,
    pub clock: u32,
    pub inst_prev_clock: u32,
    pub inst: [u32; 6],
    pub span_start: u32,
    pub span_len: u32,
}
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct cm_data_access {
    pub address: u32,
    pub prev_clock: u32,
    pub prev_value: u32,
    pub value: u32,
}
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct cm_memory_cell {
    pub address: u32,
    pub value: [u32; 4],
    pub clock: u32,
    pub multiplicity: u32,
}
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct cm_clock_update {
    pub address: u32,
    pub prev_clock: u32,
    pub value: [u32; 4],
}
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct cm_merkle_node {
    pub index: u32,
    pub depth: u32,
    pub left_value: u32,
    pub right_value: u32,
    pub parent_value: u32,
    pub left_mult: u32,
    pub right_mult: u32,
    pub parent_mult: u32,
}
#[repr(C)]
pub struct cm_prover_input {
    pub initial_pc: u32,
    pub initial_fp: u32,
    pub final_pc: u32,
    pub final_fp: u32,
    pub bundles: [*const cm_bundle; CM_N_OPCODE_COMPONENTS],
    pub n_bundles: [u64; CM_N_OPCODE_COMPONENTS],
    pub data_accesses: *const cm_data_access,
    pub n_data_accesses: u64,
    pub initial_memory: *const cm_memory_cell,
    pub n_initial_memory: u64,
    pub final_memory: *const cm_memory_cell,
    pub n_final_memory: u64,
    pub clock_updates: *const cm_clock_update,
    pub n_clock_updates: u64,
    pub initial_tree: *const cm_merkle_node,
    pub n_initial_tree: u64,
    pub final_tree: *const cm_merkle_node,
    pub n_final_tree: u64,
    pub initial_root: u32,
    pub final_root: u32,
    pub program_range: [u32; 2],
    pub input_range: [u32; 2],
    pub output_range: [u32; 2],
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct cm_pcs_config {
    pub pow_bits: u32,
    pub log_blowup_factor: u32,
    pub log_last_layer_degree_bound: u32,
    pub n_queries: u32,
}
#[repr(C)]
pub struct cm_proof {
    _private: [u8; 0],
}
#[repr(C)]
pub struct cm_device_input {
    _private: [u8; 0],
}
#[repr(C)]
pub struct cm_host_input {
    _private: [u8; 0],
}
#[repr(C)]
pub struct cm_host_segment {
    _private: [u8; 0],
}
/// Stwo `ColumnSampleBatch`es of one size group, flattened (QuotientOps::accumulate_quotients)
#[repr(C)]
pub struct cm_sample_batches {
    pub n_batches: u32,
    pub points: *const u32,
    pub batch_off: *const u32,
    pub col_index: *const u32,
    pub values: *const u32,
}
pub const CM_N_COMPONENTS: usize = 34;
pub const CM_N_RELATIONS: usize = 8;
pub const CM_MAX_RELATION_SIZE: usize = 16;
pub const CM_N_PREPROCESSED: usize = 7;
/// `Relations::draw` (components/mod.rs:311-323): z and alpha^0.. of every relation, QM31 as 4 words
#[repr(C)]
#[derive(Clone, Copy)]
pub struct cm_relations {
    pub z: [[u32; 4]; CM_N_RELATIONS],
    pub alpha_pow: [[[u32; 4]; CM_MAX_RELATION_SIZE]; CM_N_RELATIONS],
}
/// One runner segment (crates/common/src/execution.rs:10-15) as plain arrays
#[repr(C)]
pub struct cm_runner_segment {
    pub trace: *const u32,
    pub n_trace: u64,
    pub memory_trace: *const u32,
    pub n_memory_trace: u64,
    pub initial_memory: *const u32,
    pub n_initial_memory: u64,
    pub program_range: [u32; 2],
    pub input_range: [u32; 2],
    pub output_range: [u32; 2],
    /// (revision 6) heap cells at segment start: index i = the cell at MAX_ADDRESS - i (runner/src/vm/mod.rs:205-221)
    pub initial_heap: *const u32,
    pub n_initial_heap: u64,
}
/// Collectives of the sharded prover (cm_prove_sharded): two blocking calls over two device staging buffers
#[repr(C)]
pub struct cm_comm {
    /// `size_of::<cm_comm>()`: the library only reads the optional fields this size covers
    pub struct_size: u32,
    pub rank: u32,
    pub world: u32,
    pub ctx: *mut c_void,
    pub send_buf: *mut u32,
    pub recv_buf: *mut u32,
    pub buf_words: u64,
    pub all_to_all_v: Option<unsafe extern "C" fn(ctx: *mut c_void, send_words: *const u64, recv_words: *const u64) -> i32>,
    pub all_gather: Option<unsafe extern "C" fn(ctx: *mut c_void, words_per_rank: u64) -> i32>,
    /// CM_COMM_STREAM_ORDERED (1): the callbacks enqueue on the stream given through `set_stream` and do not block
    pub flags: u32,
    pub set_stream: Option<unsafe extern "C" fn(ctx: *mut c_void, stream: cm_stream_t) -> i32>,
    /// called when this rank fails inside cm_prove_sharded, so that the communicator can release the peers (None: its own timeout)
    pub abort: Option<unsafe extern "C" fn(ctx: *mut c_void)>,
}
pub const CM_COMM_STREAM_ORDERED: u32 = 1;
#[repr(C)]
pub struct cm_rccl_comm {
    _private: [u8; 0],
}

unsafe extern "C" {
    pub fn cm_init(device: i32) -> i32;
    pub fn cm_shutdown() -> i32;
    pub fn cm_pool_trim() -> i32;
    pub fn cm_device_mem_info(free_bytes: *mut u64, total_bytes: *mut u64) -> i32;
    pub fn cm_last_error(buf: *mut c_char, buf_len: usize) -> i32;
    pub fn cm_stream_create(out: *mut cm_stream_t) -> i32;
    pub fn cm_stream_destroy(s: cm_stream_t) -> i32;
    pub fn cm_stream_sync(s: cm_stream_t) -> i32;
    pub fn cm_set_cpu_affinity(mode: i32) -> i32;
    pub fn cm_get_cpu_affinity() -> i32;
    pub fn cm_set_framing(spec: *const c_char) -> i32;
    pub fn cm_get_framing(buf: *mut c_char, buf_len: usize) -> i32;
    pub fn cm_set_transcript_log(on: i32) -> i32;
    pub fn cm_col_alloc(n_u32: u64, out: *mut cm_handle) -> i32;
    pub fn cm_col_free(h: cm_handle) -> i32;
    pub fn cm_col_h2d(h: cm_handle, src: *const u32, n_u32: u64, s: cm_stream_t) -> i32;
    pub fn cm_col_d2h(h: cm_handle, dst: *mut u32, n_u32: u64, s: cm_stream_t) -> i32;
    pub fn cm_col_read(h: cm_handle, offset_u32: u64, dst: *mut u32, n_u32: u64, s: cm_stream_t) -> i32;
    pub fn cm_col_write(h: cm_handle, offset_u32: u64, src: *const u32, n_u32: u64, s: cm_stream_t) -> i32;
    pub fn cm_col_copy(dst: cm_handle, src: cm_handle, n_u32: u64, s: cm_stream_t) -> i32;
    pub fn cm_bit_reverse(cols: *const cm_handle, n_cols: u32, log_n: u32, s: cm_stream_t) -> i32;
    pub fn cm_twiddles_precompute(log_size: u32, tw_out: *mut cm_handle) -> i32;
    pub fn cm_twiddles_free(tw: cm_handle) -> i32;
    pub fn cm_interpolate(cols: *const cm_handle, n_cols: u32, log_n: u32, tw: cm_handle, s: cm_stream_t) -> i32;
    pub fn cm_evaluate(coeffs: *const cm_handle, n_cols: u32, log_n: u32, log_out: u32, tw: cm_handle, out: *const cm_handle, s: cm_stream_t) -> i32;
    pub fn cm_interpolate_extend(evals: *const cm_handle, coeffs: *const cm_handle, lde: *const cm_handle, n_cols: u32, log_n: u32, tw: cm_handle, s: cm_stream_t) -> i32;
    pub fn cm_eval_at_point(coeffs: *const cm_handle, n_cols: u32, log_n: u32, pt_xy: *const u32, out: *mut u32, s: cm_stream_t) -> i32;
    pub fn cm_merkle_commit_layer(log_size: u32, prev_layer: cm_handle, cols: *const cm_handle, n_cols: u32, out_hashes: cm_handle, s: cm_stream_t) -> i32;
    pub fn cm_merkle_commit(cols: *const cm_handle, col_logs: *const u32, n_cols: u32, root: *mut u8, s: cm_stream_t) -> i32;
    pub fn cm_grind(digest: *const u8, pow_bits: u32, nonce_out: *mut u64, s: cm_stream_t) -> i32;
    pub fn cm_batch_inverse_m31(src: cm_handle, out: cm_handle, n: u64, s: cm_stream_t) -> i32;
    pub fn cm_batch_inverse_qm31(src: *const cm_handle, out: *const cm_handle, n: u64, s: cm_stream_t) -> i32;
    pub fn cm_fri_fold_circle_into_line(dst: *const cm_handle, src: *const cm_handle, alpha: *const u32, log_n: u32, tw: cm_handle, s: cm_stream_t) -> i32;
    pub fn cm_fri_fold_line(src: *const cm_handle, alpha: *const u32, log_n: u32, tw: cm_handle, out: *const cm_handle, s: cm_stream_t) -> i32;
    pub fn cm_fri_fold_line_leaves(src: *const cm_handle, circle: *const cm_handle, alpha: *const u32, alpha_circle: *const u32, log_n: u32, tw: cm_handle, out: *const cm_handle, leaf_hashes: cm_handle, s: cm_stream_t) -> i32;
    pub fn cm_accumulate_quotients(log_size: u32, cols: *const cm_handle, n_cols: u32, batches: *const cm_sample_batches, random_coeff: *const u32, out: *const cm_handle, tw: cm_handle, s: cm_stream_t) -> i32;
    pub fn cm_vm_run(instr_words: *const u32, instr_lens: *const u32, n_instr: u32, entry_pc: u32, args: *const u32, n_args: u32, n_returns: u32, max_steps: u64, segment_index: u32, out: *mut *mut cm_host_input, n_segments_out: *mut u32) -> i32;
    pub fn cm_synth_fibonacci(n: u32, max_steps: u64, segment_index: u32, out: *mut *mut cm_host_input) -> i32;
    pub fn cm_host_input_view(h: *const cm_host_input) -> *const cm_prover_input;
    pub fn cm_host_input_steps(h: *const cm_host_input) -> u64;
    pub fn cm_host_input_free(h: *mut cm_host_input) -> i32;
    pub fn cm_adapter_memory_script(preload: *const u32, n_preload: u32, script: *const u32, n: u32, results: *mut u32, n_clock_updates: *mut u32, clock_updates_out: *mut u32, cu_cap: u32, query_addrs: *const u32, n_query: u32, state_out: *mut u32) -> i32;
    pub fn cm_adapter_partial_tree(cells: *const u32, n: u32, initial: i32, ranges: *const u32, nodes_out: *mut u32, cap: u64, n_nodes: *mut u64, root: *mut u32) -> i32;
    pub fn cm_poseidon2_permute(state: *mut u32) -> i32;
    pub fn cm_prove_segment(input: *const cm_prover_input, config: *const cm_pcs_config, out: *mut *mut cm_proof) -> i32;
    pub fn cm_proof_free(p: *mut cm_proof) -> i32;
    pub fn cm_input_upload(input: *const cm_prover_input, out: *mut *mut cm_device_input) -> i32;
    pub fn cm_input_free(h: *mut cm_device_input) -> i32;
    pub fn cm_prove_device(input: *const cm_device_input, config: *const cm_pcs_config, out: *mut *mut cm_proof) -> i32;
    pub fn cm_verify_proof(p: *const cm_proof, expected: *const cm_pcs_config) -> i32;
    pub fn cm_verify_proof_words(words: *const u32, n_words: u64, expected: *const cm_pcs_config) -> i32;
    pub fn cm_shard_plan(input: *const cm_prover_input, config: *const cm_pcs_config, world: u32, owner: *mut i32, staging_words: *mut u64) -> i32;
    pub fn cm_shard_plan_columns(input: *const cm_prover_input, config: *const cm_pcs_config, world: u32, trace_col_owner: *mut i32, n_trace_cols: *mut u32, interaction_col_owner: *mut i32, n_interaction_cols: *mut u32, load_cells: *mut u64) -> i32;
    pub fn cm_prove_sharded(input: *const cm_device_input, config: *const cm_pcs_config, comm: *const cm_comm, out: *mut *mut cm_proof) -> i32;
    pub fn cm_rccl_unique_id(id_out: *mut u8) -> i32;
    pub fn cm_rccl_comm_create(id: *const u8, rank: u32, world: u32, staging_words: u64, out: *mut *mut cm_rccl_comm) -> i32;
    pub fn cm_rccl_comm_view(c: *const cm_rccl_comm) -> *const cm_comm;
    pub fn cm_rccl_comm_destroy(c: *mut cm_rccl_comm) -> i32;
    pub fn cm_prove_many(inputs: *const *const cm_device_input, n: u32, config: *const cm_pcs_config, inflight: u32, outs: *mut *mut cm_proof) -> i32;
    pub fn cm_prove_many_host(inputs: *const *const cm_prover_input, n: u32, config: *const cm_pcs_config, inflight: u32, outs: *mut *mut cm_proof) -> i32;
    pub fn cm_prove_many_segments(segments: *const *const cm_runner_segment, n: u32, config: *const cm_pcs_config, inflight: u32, outs: *mut *mut cm_proof) -> i32;
    pub fn cm_set_preprocessed_cache(on: i32) -> i32;
    pub fn cm_set_twiddle_cache(on: i32) -> i32;
    pub fn cm_set_device_tail(on: i32) -> i32;
    pub fn cm_set_tuning(key: *const c_char, value: i32) -> i32;
    pub fn cm_tail_list(positions: *const u32, n_positions: u32, log_domain: u32, qmask: u32, list: u32, k: u32, out: *mut u32, cap: u32, n_out: *mut u32) -> i32;
    pub fn cm_proof_from_words(words: *const u32, n_words: u64, out: *mut *mut cm_proof) -> i32;
    pub fn cm_proof_words(p: *const cm_proof, words_out: *mut *const u32, n_out: *mut u64) -> i32;
    pub fn cm_proof_json(p: *const cm_proof, json_out: *mut *const c_char, len_out: *mut usize) -> i32;
    pub fn cm_proof_transcript(p: *const cm_proof, json_out: *mut *const c_char, len_out: *mut usize) -> i32;
    pub fn cm_proof_commitments(p: *const cm_proof, roots: *mut [u8; 32]) -> i32;
    pub fn cm_adapt_segment_device(seg: *const cm_runner_segment, out: *mut *mut cm_device_input) -> i32;
    pub fn cm_device_input_download(src: *const cm_device_input, out: *mut *mut cm_host_input) -> i32;
    pub fn cm_vm_segment(instr_words: *const u32, instr_lens: *const u32, n_instr: u32, entry_pc: u32, args: *const u32, n_args: u32, n_returns: u32, max_steps: u64, segment_index: u32, out: *mut *mut cm_host_segment, n_segments_out: *mut u32) -> i32;
    pub fn cm_synth_fibonacci_segment(n: u32, max_steps: u64, segment_index: u32, out: *mut *mut cm_host_segment) -> i32;
    pub fn cm_host_segment_view(h: *const cm_host_segment) -> *const cm_runner_segment;
    pub fn cm_segment_serialize_trace(s: *const cm_runner_segment, out: *mut u8, cap: u64, len: *mut u64) -> i32;
    pub fn cm_segment_serialize_memory_trace(s: *const cm_runner_segment, with_header: i32, out: *mut u8, cap: u64, len: *mut u64) -> i32;
    pub fn cm_host_segment_set_initial_heap(h: *mut cm_host_segment, initial_heap: *const u32, n_initial_heap: u64) -> i32;
    pub fn cm_segment_from_artifacts(trace: *const u8, trace_len: u64, mem: *const u8, mem_len: u64, mem_has_header: i32, initial_memory: *const u32, n_initial_memory: u64, ranges: *const u32, out: *mut *mut cm_host_segment) -> i32;
    pub fn cm_host_segment_free(h: *mut cm_host_segment) -> i32;
    pub fn cm_component_info(component: i32, n_trace_cols: *mut u32, n_interaction_cols: *mut u32, n_constraints: *mut u32) -> i32;
    pub fn cm_component_log_size(input: *const cm_device_input, component: i32, log_size: *mut u32) -> i32;
    pub fn cm_trace_write(input: *const cm_device_input, component: i32, cols: *const cm_handle, s: cm_stream_t) -> i32;
    pub fn cm_histogram(component: i32, trace_cols: *const cm_handle, log_size: u32, rc8: cm_handle, rc16: cm_handle, rc20: cm_handle, bitwise: cm_handle, s: cm_stream_t) -> i32;
    pub fn cm_preprocessed_column(id: i32, col: cm_handle, s: cm_stream_t) -> i32;
    pub fn cm_interaction_write(component: i32, trace_cols: *const cm_handle, preprocessed: *const cm_handle, log_size: u32, relations: *const cm_relations, out: *const cm_handle, claimed_sum: *mut u32, s: cm_stream_t) -> i32;
    pub fn cm_constraints_accumulate(component: i32, trace_lde: *const cm_handle, interaction_lde: *const cm_handle, preprocessed_lde: *const cm_handle, log_size: u32, relations: *const cm_relations, coeff_powers: *const u32, claimed_sum: *const u32, acc: *const cm_handle, s: cm_stream_t) -> i32;
    pub fn cm_accumulate(dst: *const cm_handle, src: *const cm_handle, n: u64, s: cm_stream_t) -> i32;
    pub fn cm_generate_secure_powers(felt: *const u32, n: u64, out: *mut u32) -> i32;
    pub fn cm_col_zero(h: cm_handle, n_u32: u64, s: cm_stream_t) -> i32;
    pub fn cm_fri_decompose(f: *const cm_handle, log_n: u32, lambda_out: *mut u32, s: cm_stream_t) -> i32;
    pub fn cm_kprof_enable(on: i32) -> i32;
    pub fn cm_kprof_report(buf: *mut c_char, buf_len: usize) -> i32;
    pub fn cm_kprof_filter(name: *const c_char) -> i32;
    pub fn cm_proof_stats(p: *const cm_proof, cells: *mut u64, steps: *mut u64, phase_ms: *mut f64, n_phases: u32) -> i32;
}
