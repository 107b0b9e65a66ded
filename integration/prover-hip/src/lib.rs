//! `prove_cairo_m_hip` — the MI355X twin of `cairo_m_prover::prover::prove_cairo_m::<Blake2sMerkleChannel>`
//! (crates/prover/src/prover.rs:23-147): same arguments, same result type, same error type.  The whole Stwo path runs in
//! libcairom_hip.so (hand-written gfx950 kernels); this crate only flattens `ProverInput` into the C ABI
//! (include/cairom_hip.h `cm_prover_input`) and deserialises the returned `Proof<Blake2sMerkleHasher>` JSON.
//!
//! Shipped as source: the HIP repository's build image has no Rust toolchain, so this file has never been compiled there.
//! The two data-layout contracts it relies on are tested on the C side: the proof JSON has exactly the serde shape of
//! `Proof<H>` (tests/test_proof_json.py, schema extracted from this reference's structs) and `cm_prover_input` is what the
//! HIP prover is tested against (tests/test_gpu_prove.py).
pub mod ffi;

use std::ffi::CStr;
use std::sync::Once;

use cairo_m_prover::Proof;
use cairo_m_prover::adapter::{ExecutionBundle, ProverInput};
use cairo_m_prover::errors::{ProvingError, VerificationError};
use cairo_m_prover::prover_config::REGULAR_96_BITS;
use num_traits::Zero;
use stwo_prover::core::fields::m31::M31;
use stwo_prover::core::fields::qm31::QM31;
use stwo_prover::core::pcs::PcsConfig;
use stwo_prover::core::prover::{ProvingError as StwoProvingError, VerificationError as StwoVerificationError};
use stwo_prover::core::vcs::blake2_merkle::Blake2sMerkleHasher;

use ffi::*;

/// Opcode groups of `define_opcodes!` (crates/prover/src/components/opcodes/mod.rs:223-268), in macro order: component k
/// of the HIP library = group k; inside a group the bundles of the listed opcodes are concatenated in this order, exactly
/// as `opcodes::Claim::write_trace` does (opcodes/mod.rs:51-58).
const OPCODE_GROUPS: [&[u32]; CM_N_OPCODE_COMPONENTS] = {
    use cairo_m_common::instruction::*;
    [
        &[ASSERT_EQ_FP_IMM],
        &[CALL_ABS_IMM],
        &[JMP_ABS_IMM, JMP_REL_IMM],
        &[JNZ_FP_IMM],
        &[RET],
        &[STORE_IMM],
        &[STORE_ADD_FP_FP, STORE_SUB_FP_FP, STORE_MUL_FP_FP, STORE_DIV_FP_FP],
        &[STORE_ADD_FP_IMM, STORE_MUL_FP_IMM],
        &[STORE_DOUBLE_DEREF_FP, STORE_TO_DOUBLE_DEREF_FP_IMM],
        &[STORE_DOUBLE_DEREF_FP_FP, STORE_TO_DOUBLE_DEREF_FP_FP],
        &[STORE_FRAME_POINTER],
        &[U32_STORE_IMM],
        &[U32_STORE_ADD_FP_IMM],
        &[U32_STORE_MUL_FP_IMM],
        &[U32_STORE_DIV_REM_FP_IMM],
        &[U32_STORE_EQ_FP_FP],
        &[U32_STORE_EQ_FP_IMM],
        &[U32_STORE_LT_FP_IMM],
        &[U32_STORE_LT_FP_FP],
        &[U32_STORE_ADD_FP_FP],
        &[U32_STORE_SUB_FP_FP],
        &[U32_STORE_MUL_FP_FP],
        &[U32_STORE_DIV_REM_FP_FP],
        &[U32_STORE_AND_FP_FP, U32_STORE_OR_FP_FP, U32_STORE_XOR_FP_FP],
        &[U32_STORE_AND_FP_IMM, U32_STORE_OR_FP_IMM, U32_STORE_XOR_FP_IMM],
        &[STORE_LE_FP_IMM],
    ]
};

fn last_error() -> String {
    let mut buf = vec![0i8; 2048];
    unsafe {
        cm_last_error(buf.as_mut_ptr(), buf.len());
        CStr::from_ptr(buf.as_ptr()).to_string_lossy().into_owned()
    }
}

/// Selects the GPU once per process (one process per GPU; `CAIROM_HIP_DEVICE` or LOCAL_RANK picks the device).
fn ensure_init() {
    static INIT: Once = Once::new();
    INIT.call_once(|| {
        let dev = std::env::var("CAIROM_HIP_DEVICE")
            .or_else(|_| std::env::var("LOCAL_RANK"))
            .ok()
            .and_then(|s| s.parse().ok())
            .unwrap_or(0);
        let rc = unsafe { cm_init(dev) };
        assert!(rc == 0, "cm_init({dev}) failed: {}", last_error());
    });
}

fn bundle(b: &ExecutionBundle) -> cm_bundle {
    // what `Pack::pack` reads (crates/prover/src/utils/execution_bundle.rs:29-75)
    let words = b.instruction.instruction.to_smallvec();
    let mut inst = [0u32; 6];
    for (k, w) in words.iter().enumerate() {
        inst[k] = w.0;
    }
    cm_bundle {
        pc: b.registers.pc.0,
        fp: b.registers.fp.0,
        clock: b.clock.0,
        inst_prev_clock: b.instruction.prev_clock.0,
        inst,
        span_start: b.access_span.start,
        span_len: b.access_span.len as u32,
    }
}

fn cell(addr: &M31, (value, clock, mult): &(QM31, M31, M31)) -> cm_memory_cell {
    let v = value.to_m31_array();
    cm_memory_cell { address: addr.0, value: [v[0].0, v[1].0, v[2].0, v[3].0], clock: clock.0, multiplicity: mult.0 }
}

fn pcs(c: &PcsConfig) -> cm_pcs_config {
    cm_pcs_config {
        pow_bits: c.pow_bits,
        log_blowup_factor: c.fri_config.log_blowup_factor,
        log_last_layer_degree_bound: c.fri_config.log_last_layer_degree_bound,
        n_queries: c.fri_config.n_queries as u32,
    }
}

/// Owned flattening of a `ProverInput`; `view()` borrows it as the C struct.
struct Flat {
    bundles: Vec<Vec<cm_bundle>>,
    data_accesses: Vec<cm_data_access>,
    initial_memory: Vec<cm_memory_cell>,
    final_memory: Vec<cm_memory_cell>,
    clock_updates: Vec<cm_clock_update>,
    initial_tree: Vec<cm_merkle_node>,
    final_tree: Vec<cm_merkle_node>,
    regs: [u32; 4],
    roots: [u32; 2],
    ranges: [[u32; 2]; 3],
}

impl Flat {
    /// Consumes the bundles like `prove_cairo_m` does (opcodes/mod.rs:53-58 drains `states_by_opcodes`).
    fn new(input: &mut ProverInput) -> Self {
        let ins = &mut input.instructions;
        let bundles = OPCODE_GROUPS
            .iter()
            .map(|group| {
                let mut v = Vec::new();
                for opcode in group.iter() {
                    if let Some(states) = ins.states_by_opcodes.get_mut(opcode) {
                        v.extend(states.drain(..).map(|b| bundle(&b)));
                    }
                }
                v
            })
            .collect();
        let data_accesses = ins
            .data_accesses
            .iter()
            .map(|a| cm_data_access { address: a.address.0, prev_clock: a.prev_clock.0, prev_value: a.prev_value.0, value: a.value.0 })
            .collect();
        // Row order of the memory component: the reference iterates its HashMaps (memory.rs:104-133), i.e. an unspecified
        // order; the library commits the rows in the order given here.  Ascending addresses make the proof reproducible.
        let mut init: Vec<_> = input.memory.initial_memory.iter().collect();
        init.sort_by_key(|(a, _)| a.0);
        let mut fin: Vec<_> = input.memory.final_memory.iter().collect();
        fin.sort_by_key(|(a, _)| a.0);
        let node = |n: &cairo_m_prover::adapter::merkle::NodeData| {
            let a = n.to_m31_array();
            cm_merkle_node {
                index: a[0].0, depth: a[1].0, left_value: a[2].0, right_value: a[3].0, parent_value: a[4].0,
                left_mult: a[5].0, right_mult: a[6].0, parent_mult: a[7].0,
            }
        };
        let r = &input.public_address_ranges;
        Flat {
            bundles,
            data_accesses,
            initial_memory: init.into_iter().map(|(a, s)| cell(a, s)).collect(),
            final_memory: fin.into_iter().map(|(a, s)| cell(a, s)).collect(),
            clock_updates: input
                .memory
                .clock_update_data
                .iter()
                .map(|(addr, prev_clk, value)| {
                    let v = value.to_m31_array();
                    cm_clock_update { address: addr.0, prev_clock: prev_clk.0, value: [v[0].0, v[1].0, v[2].0, v[3].0] }
                })
                .collect(),
            initial_tree: input.merkle_trees.initial_tree.iter().map(node).collect(),
            final_tree: input.merkle_trees.final_tree.iter().map(node).collect(),
            regs: [ins.initial_registers.pc.0, ins.initial_registers.fp.0, ins.final_registers.pc.0, ins.final_registers.fp.0],
            roots: [
                input.merkle_trees.initial_root.unwrap_or_else(M31::zero).0,
                input.merkle_trees.final_root.unwrap_or_else(M31::zero).0,
            ],
            ranges: [[r.program.start, r.program.end], [r.input.start, r.input.end], [r.output.start, r.output.end]],
        }
    }

    fn view(&self) -> cm_prover_input {
        let mut bundles = [std::ptr::null(); CM_N_OPCODE_COMPONENTS];
        let mut n_bundles = [0u64; CM_N_OPCODE_COMPONENTS];
        for (k, v) in self.bundles.iter().enumerate() {
            bundles[k] = v.as_ptr();
            n_bundles[k] = v.len() as u64;
        }
        cm_prover_input {
            initial_pc: self.regs[0], initial_fp: self.regs[1], final_pc: self.regs[2], final_fp: self.regs[3],
            bundles, n_bundles,
            data_accesses: self.data_accesses.as_ptr(), n_data_accesses: self.data_accesses.len() as u64,
            initial_memory: self.initial_memory.as_ptr(), n_initial_memory: self.initial_memory.len() as u64,
            final_memory: self.final_memory.as_ptr(), n_final_memory: self.final_memory.len() as u64,
            clock_updates: self.clock_updates.as_ptr(), n_clock_updates: self.clock_updates.len() as u64,
            initial_tree: self.initial_tree.as_ptr(), n_initial_tree: self.initial_tree.len() as u64,
            final_tree: self.final_tree.as_ptr(), n_final_tree: self.final_tree.len() as u64,
            initial_root: self.roots[0], final_root: self.roots[1],
            program_range: self.ranges[0], input_range: self.ranges[1], output_range: self.ranges[2],
        }
    }
}

struct ProofHandle(*mut cm_proof);
impl Drop for ProofHandle {
    fn drop(&mut self) {
        unsafe { cm_proof_free(self.0) };
    }
}

/// Twin of `prove_cairo_m::<Blake2sMerkleChannel>` (crates/prover/src/prover.rs:23-29).  `input` is consumed in place the
/// same way (bundle vectors drained).  Status 10 of the library = the one Stwo error the reference surfaces
/// (`ConstraintsNotSatisfied`, errors.rs:14-18); any other failure is a bug or a device error and panics, like the
/// `unwrap`/`expect`s of the reference path do.
pub fn prove_cairo_m_hip(input: &mut ProverInput, pcs_config: Option<PcsConfig>) -> Result<Proof<Blake2sMerkleHasher>, ProvingError> {
    ensure_init();
    let cfg = pcs(&pcs_config.unwrap_or(REGULAR_96_BITS));
    let flat = Flat::new(input);
    let view = flat.view();
    let mut out: *mut cm_proof = std::ptr::null_mut();
    let rc = unsafe { cm_prove_segment(&view, &cfg, &mut out) };
    match rc {
        0 => {}
        10 => return Err(ProvingError::Stwo(StwoProvingError::ConstraintsNotSatisfied)),
        _ => panic!("libcairom_hip: status {rc}: {}", last_error()),
    }
    let handle = ProofHandle(out);
    let (mut ptr, mut len) = (std::ptr::null(), 0usize);
    let rc = unsafe { cm_proof_json(handle.0, &mut ptr, &mut len) };
    assert!(rc == 0, "cm_proof_json: {}", last_error());
    let json = unsafe { std::slice::from_raw_parts(ptr as *const u8, len) };
    // serde layout of `Proof<H>` (lib.rs:61-73), the text `sonic_rs::to_string(&proof)` would produce (main.rs:86-91)
    let proof: Proof<Blake2sMerkleHasher> = sonic_rs::from_slice(json).expect("libcairom_hip returned a malformed Proof JSON");
    Ok(proof)
}

/// `verify_cairo_m::<Blake2sMerkleChannel>` stays the reference's own function: the value returned above is an ordinary
/// `Proof<Blake2sMerkleHasher>`.  This helper is the library-side verifier (host code, no GPU) for callers that want the
/// check without Stwo: same acceptance conditions, error mapped onto the reference's enum.
pub fn verify_words_hip(proof: &ProofHandleRef, pcs_config: Option<PcsConfig>) -> Result<(), VerificationError> {
    let cfg = pcs(&pcs_config.unwrap_or(REGULAR_96_BITS));
    let rc = unsafe { cm_verify_proof(proof.0, &cfg) };
    match rc {
        0 => Ok(()),
        _ if last_error().contains("InvalidLogupSum") => Err(VerificationError::InvalidLogupSum),
        _ if last_error().contains("ProofOfWork") => Err(VerificationError::Stwo(StwoVerificationError::ProofOfWork)),
        _ if last_error().contains("OodsNotMatching") => Err(VerificationError::Stwo(StwoVerificationError::OodsNotMatching)),
        _ => Err(VerificationError::Stwo(StwoVerificationError::InvalidStructure(last_error()))),
    }
}
/// Borrowed library proof object (e.g. kept by a caller that proves many segments and verifies them later).
pub struct ProofHandleRef(pub *const cm_proof);
