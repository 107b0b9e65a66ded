//! Raw bindings of include/cairom_hip.h (only what the whole-path twin needs).  Layouts are `#[repr(C)]` mirrors of the
//! C structs; every function returns a status (0 = OK) and leaves a message for `cm_last_error`.
#![allow(non_camel_case_types)]
use std::os::raw::c_char;

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

unsafe extern "C" {
    pub fn cm_init(device: i32) -> i32;
    pub fn cm_last_error(buf: *mut c_char, len: usize) -> i32;
    pub fn cm_prove_segment(input: *const cm_prover_input, cfg: *const cm_pcs_config, out: *mut *mut cm_proof) -> i32;
    pub fn cm_verify_proof(proof: *const cm_proof, expected: *const cm_pcs_config) -> i32;
    pub fn cm_proof_json(p: *const cm_proof, json: *mut *const c_char, len: *mut usize) -> i32;
    pub fn cm_proof_free(p: *mut cm_proof) -> i32;
}
