//! `prove_cairo_m_hip` — the MI355X twin of `cairo_m_prover::prover::prove_cairo_m::<Blake2sMerkleChannel>`
//! (crates/prover/src/prover.rs:23-147): same arguments, same result type, same error type.  The whole Stwo path runs in
//! libcairom_hip.so (hand-written gfx950 kernels); this crate only flattens `ProverInput` into the C ABI
//! (include/cairom_hip.h `cm_prover_input`) and deserialises the returned `Proof<Blake2sMerkleHasher>` JSON.
//!
//! Shipped as source: the HIP repository's build image has no Rust toolchain, so this file has never been compiled there.
//! The two data-layout contracts it relies on are tested on the C side: the proof JSON has exactly the serde shape of
//! `Proof<H>` (tests/test_proof_json.py, schema extracted from this reference's structs) and `cm_prover_input` is what the
//! HIP prover is tested against (tests/test_gpu_prove.py).
pub mod backend;
pub mod ffi;
pub mod flat;

use std::ffi::CStr;
use std::sync::Once;

use cairo_m_prover::Proof;
use cairo_m_prover::adapter::ProverInput;
use cairo_m_prover::errors::{ProvingError, VerificationError};
use cairo_m_prover::prover_config::REGULAR_96_BITS;
use stwo_prover::core::pcs::PcsConfig;
use stwo_prover::core::prover::{ProvingError as StwoProvingError, VerificationError as StwoVerificationError};
use stwo_prover::core::vcs::blake2_merkle::Blake2sMerkleHasher;

use ffi::*;
use flat::{Flat, MemoryOrder};

/// Opcode groups of `define_opcodes!` (crates/prover/src/components/opcodes/mod.rs:223-268), in macro order: component k
/// of the HIP library = group k; inside a group the bundles of the listed opcodes are concatenated in this order, exactly
/// as `opcodes::Claim::write_trace` does (opcodes/mod.rs:51-58).
pub(crate) const OPCODE_GROUPS: [&[u32]; CM_N_OPCODE_COMPONENTS] = {
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

pub(crate) fn last_error() -> String {
    let mut buf = vec![0i8; 2048];
    unsafe {
        cm_last_error(buf.as_mut_ptr(), buf.len());
        CStr::from_ptr(buf.as_ptr()).to_string_lossy().into_owned()
    }
}

/// Selects the GPU once per process (one process per GPU; `CAIROM_HIP_DEVICE` or LOCAL_RANK picks the device).
pub(crate) fn ensure_init() {
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

fn pcs(c: &PcsConfig) -> cm_pcs_config {
    cm_pcs_config {
        pow_bits: c.pow_bits,
        log_blowup_factor: c.fri_config.log_blowup_factor,
        log_last_layer_degree_bound: c.fri_config.log_last_layer_degree_bound,
        n_queries: c.fri_config.n_queries as u32,
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
    // ascending addresses: the reference iterates a HashMap here (components/memory.rs:105-109), i.e. an unspecified order
    let flat = Flat::new(input, MemoryOrder::AscendingAddress);
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

/// Streaming form for a service that proves the continuation segments of a run (crates/runner/src/vm/mod.rs:184-240 cuts them,
/// crates/prover/tests/prover.rs:203-243 proves them one by one): `cm_prove_many_host` uploads segment k + 1 on the calling
/// thread while up to `inflight` library threads prove the segments before it, so the 6 ms PCIe copy of a 2^22-step segment hides
/// under the 10 ms proof of its predecessor.  Proofs come back in input order; the first failing segment is reported
/// (`ConstraintsNotSatisfied` for status 10) after the others have been proved.
pub fn prove_segments_hip(
    inputs: &mut [ProverInput],
    pcs_config: Option<PcsConfig>,
    inflight: u32,
) -> Result<Vec<Proof<Blake2sMerkleHasher>>, ProvingError> {
    ensure_init();
    let cfg = pcs(&pcs_config.unwrap_or(REGULAR_96_BITS));
    let flats: Vec<Flat> = inputs.iter_mut().map(|i| Flat::new(i, MemoryOrder::AscendingAddress)).collect();
    let views: Vec<_> = flats.iter().map(|f| f.view()).collect();
    let ptrs: Vec<*const cm_prover_input> = views.iter().map(|v| v as *const cm_prover_input).collect();
    let mut outs: Vec<*mut cm_proof> = vec![std::ptr::null_mut(); inputs.len()];
    let rc = unsafe { cm_prove_many_host(ptrs.as_ptr(), ptrs.len() as u32, &cfg, inflight, outs.as_mut_ptr()) };
    let handles: Vec<ProofHandle> = outs.into_iter().filter(|p| !p.is_null()).map(ProofHandle).collect();   // freed on drop
    match rc {
        0 => {}
        10 => return Err(ProvingError::Stwo(StwoProvingError::ConstraintsNotSatisfied)),
        _ => panic!("libcairom_hip: status {rc}: {}", last_error()),
    }
    handles
        .iter()
        .map(|h| {
            let (mut ptr, mut len) = (std::ptr::null(), 0usize);
            let rc = unsafe { cm_proof_json(h.0, &mut ptr, &mut len) };
            assert!(rc == 0, "cm_proof_json: {}", last_error());
            let json = unsafe { std::slice::from_raw_parts(ptr as *const u8, len) };
            Ok(sonic_rs::from_slice(json).expect("libcairom_hip returned a malformed Proof JSON"))
        })
        .collect()
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
