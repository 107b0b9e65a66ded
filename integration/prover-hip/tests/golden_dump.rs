//! Golden-vector harness: runs the REFERENCE prover (`cairo_m_prover::prover::prove_cairo_m`, Stwo `SimdBackend`) and writes
//! everything the HIP repository needs to pin its restatement of Stwo to it — one `ref_<case>.json` per case:
//!
//! ```text
//! { "name": ..., "source": "reference", "stwo_rev": "ab57a1c", "pcs_config": [pow_bits, log_blowup, log_last_layer, n_queries],
//!   "input":      the ProverInput flattened like cm_prover_input (src/flat.rs), memory rows IN THE ORDER THE REFERENCE'S
//!                 HashMaps ITERATE (components/memory.rs:105-109 — SURVEY F3),
//!   "transcript": one entry per Channel-trait call the prover made, in order:
//!                 {"op": "mix_u64" | "mix_u32s" | "mix_felts" | "mix_root" | "draw_felt" | "draw_felts" | "draw_random_bytes",
//!                  "digest": hex of the channel digest AFTER the call, "n_words": words mixed in / drawn, "words": first <= 16},
//!   "commitments": [4 hex roots], "interaction_pow": n, "verified": true,
//!   "proof": the `Proof<Blake2sMerkleHasher>` exactly as `sonic_rs::to_string(&proof)` writes it (main.rs:86-91) }
//! ```
//!
//! How the transcript is captured without touching Stwo: `prove_cairo_m` is generic over `MC: MerkleChannel`
//! (crates/prover/src/prover.rs:23-29), so the harness instantiates it with `LoggingMerkleChannel`, whose channel wraps
//! `Blake2sChannel` and records the digest after every trait call.  The three impls below are legal under the orphan rule
//! (a local type appears as the trait's type parameter).  The proof produced is byte-identical to the one
//! `Blake2sMerkleChannel` gives: the wrapper forwards every call unchanged.
//!
//! Cases: `unchanged_memory` is the hand-built input of crates/prover/tests/prover.rs:33-112 (no compiler, no runner).  Every
//! `tests/golden/cases/*.case.json` (made by tools/make_ref_cases.py of the HIP repository from ITS synthetic VM: trace,
//! memory log, initial memory, public ranges) goes through the reference's own `import_from_runner_output`, so those cases
//! also pin the HIP repository's adapter to the reference's.
//!
//! Run (any box with the reference workspace, Rust nightly-2025-04-06 and the stwo submodule checked out; no GPU needed —
//! nothing here calls into libcairom_hip.so, but the crate links it: point CAIROM_HIP_DIR at a built copy):
//!
//! ```text
//! CAIROM_GOLDEN_DIR=/path/to/hip-repo/tests/golden \
//!   cargo test -p cairo-m-prover-hip --release --test golden_dump -- --nocapture
//! ```
//! then, in the HIP repository: `python -m pytest tests/test_ref_golden.py` (CPU: oracle vs reference; `-m gpu`: HIP vs reference).
//!
//! NOT compiled in the HIP repository's build image (no Rust toolchain there): shipped as source.
use std::cell::RefCell;
use std::collections::HashMap;
use std::path::PathBuf;
use std::sync::Mutex;

use cairo_m_common::execution::Segment;
use cairo_m_common::state::MemoryEntry;
use cairo_m_common::{PublicAddressRanges, State};
use cairo_m_prover::adapter::memory::Memory;
use cairo_m_prover::adapter::merkle::{TreeType, build_partial_merkle_tree};
use cairo_m_prover::adapter::{HashInput, Instructions, MerkleTrees, ProverInput, import_from_runner_output};
use cairo_m_prover::poseidon2::Poseidon2Hash;
use cairo_m_prover::prover::prove_cairo_m;
use cairo_m_prover::prover_config::REGULAR_96_BITS;
use cairo_m_prover::verifier::verify_cairo_m;
use cairo_m_prover_hip::flat::{Flat, MemoryOrder};
use serde::Deserialize;
use stwo_prover::core::backend::BackendForChannel;
use stwo_prover::core::backend::simd::SimdBackend;
use stwo_prover::core::channel::{Blake2sChannel, Channel, MerkleChannel};
use stwo_prover::core::fields::m31::M31;
use stwo_prover::core::fields::qm31::{QM31, SecureField};
use stwo_prover::core::proof_of_work::GrindOps;
use stwo_prover::core::vcs::blake2_hash::Blake2sHash;
use stwo_prover::core::vcs::blake2_merkle::{Blake2sMerkleChannel, Blake2sMerkleHasher};

// ---- the logging channel ----------------------------------------------------------------------------------------------
struct Entry {
    op: &'static str,
    digest: [u8; 32],
    n_words: usize,
    words: Vec<u32>,
}
static TRANSCRIPT: Mutex<Vec<Entry>> = Mutex::new(Vec::new());

fn record(op: &'static str, ch: &Blake2sChannel, words: &[u32]) {
    let digest: [u8; 32] = ch.digest().0;
    TRANSCRIPT.lock().unwrap().push(Entry { op, digest, n_words: words.len(), words: words.iter().take(16).copied().collect() });
}
fn felt_words(felts: &[SecureField]) -> Vec<u32> {
    felts.iter().flat_map(|f| f.to_m31_array().map(|m| m.0)).collect()
}

#[derive(Default, Clone, Debug)]
pub struct LoggingChannel {
    inner: Blake2sChannel,
}
impl Channel for LoggingChannel {
    const BYTES_PER_HASH: usize = <Blake2sChannel as Channel>::BYTES_PER_HASH;

    fn trailing_zeros(&self) -> u32 {
        self.inner.trailing_zeros()
    }
    fn mix_u32s(&mut self, data: &[u32]) {
        self.inner.mix_u32s(data);
        record("mix_u32s", &self.inner, data);
    }
    fn mix_felts(&mut self, felts: &[SecureField]) {
        self.inner.mix_felts(felts);
        record("mix_felts", &self.inner, &felt_words(felts));
    }
    fn mix_u64(&mut self, value: u64) {
        self.inner.mix_u64(value);
        record("mix_u64", &self.inner, &[value as u32, (value >> 32) as u32]);
    }
    fn draw_felt(&mut self) -> SecureField {
        let f = self.inner.draw_felt();
        record("draw_felt", &self.inner, &felt_words(&[f]));
        f
    }
    fn draw_felts(&mut self, n_felts: usize) -> Vec<SecureField> {
        let f = self.inner.draw_felts(n_felts);
        record("draw_felts", &self.inner, &felt_words(&f));
        f
    }
    fn draw_random_bytes(&mut self) -> Vec<u8> {
        let b = self.inner.draw_random_bytes();
        let words: Vec<u32> = b.chunks_exact(4).map(|c| u32::from_le_bytes(c.try_into().unwrap())).collect();
        record("draw_random_bytes", &self.inner, &words);
        b
    }
}

#[derive(Default)]
pub struct LoggingMerkleChannel;
impl MerkleChannel for LoggingMerkleChannel {
    type C = LoggingChannel;
    type H = Blake2sMerkleHasher;

    fn mix_root(channel: &mut Self::C, root: Blake2sHash) {
        Blake2sMerkleChannel::mix_root(&mut channel.inner, root);
        let words: Vec<u32> = root.0.chunks_exact(4).map(|c| u32::from_le_bytes(c.try_into().unwrap())).collect();
        record("mix_root", &channel.inner, &words);
    }
}
impl GrindOps<LoggingChannel> for SimdBackend {
    fn grind(channel: &LoggingChannel, pow_bits: u32) -> u64 {
        <SimdBackend as GrindOps<Blake2sChannel>>::grind(&channel.inner, pow_bits)
    }
}
impl BackendForChannel<LoggingMerkleChannel> for SimdBackend {}

// ---- inputs -------------------------------------------------------------------------------------------------------------
/// crates/prover/tests/prover.rs:33-112, verbatim in structure: four cells that never change, no instruction.
fn unchanged_memory_input() -> ProverInput {
    let initial_memory_data = [
        (M31(0), QM31::from_u32_unchecked(1, 2, 3, 4), M31(0), M31(0)),
        (M31(1), QM31::from_u32_unchecked(5, 6, 7, 8), M31(0), M31(0)),
        (M31(2), QM31::from_u32_unchecked(9, 10, 11, 12), M31(0), M31(0)),
        (M31(3), QM31::from_u32_unchecked(13, 14, 15, 16), M31(0), M31(0)),
    ];
    let initial_memory: HashMap<M31, (QM31, M31, M31)> =
        initial_memory_data.iter().map(|(a, v, c, m)| (*a, (*v, *c, *m))).collect();
    let memory = Memory { initial_memory: initial_memory.clone(), final_memory: initial_memory, clock_update_data: vec![] };
    let ranges = PublicAddressRanges::default();
    let (initial_tree, initial_root) = build_partial_merkle_tree::<Poseidon2Hash>(&memory.initial_memory, TreeType::Initial, &ranges);
    let (final_tree, final_root) = build_partial_merkle_tree::<Poseidon2Hash>(&memory.final_memory, TreeType::Final, &ranges);
    let mut poseidon2_inputs = Vec::<HashInput>::with_capacity(initial_tree.len() + final_tree.len());
    initial_tree.iter().for_each(|n| poseidon2_inputs.push(n.to_hash_input()));
    final_tree.iter().for_each(|n| poseidon2_inputs.push(n.to_hash_input()));
    ProverInput {
        merkle_trees: MerkleTrees { initial_tree, final_tree, initial_root, final_root },
        public_address_ranges: PublicAddressRanges { program: 0..0, input: 0..0, output: 0..0 },
        memory,
        instructions: Instructions::default(),
        poseidon2_inputs,
    }
}

/// One runner segment as the HIP repository's synthetic VM emits it (tools/make_ref_cases.py).
#[derive(Deserialize)]
struct CaseFile {
    name: String,
    /// (pc, fp) per VM state: steps + 1 entries
    trace: Vec<[u32; 2]>,
    /// (address, v0, v1, v2, v3) per logged access, in execution order
    memory_trace: Vec<[u32; 5]>,
    /// memory at segment start: cell k = address k, value words
    initial_memory: Vec<[u32; 4]>,
    /// program, input, output [start, end)
    ranges: [u32; 6],
}
fn case_input(case: &CaseFile) -> ProverInput {
    let segment = Segment {
        // first segments only: every cell starts at clock 0 with multiplicity 0 (runner/src/vm/mod.rs:306-375)
        initial_memory: case
            .initial_memory
            .iter()
            .enumerate()
            .map(|(a, v)| (M31(a as u32), (QM31::from_u32_unchecked(v[0], v[1], v[2], v[3]), M31(0), M31(0))))
            .collect(),
        memory_trace: RefCell::new(
            case.memory_trace
                .iter()
                .map(|e| MemoryEntry { addr: M31(e[0]), value: QM31::from_u32_unchecked(e[1], e[2], e[3], e[4]) })
                .collect(),
        ),
        trace: case.trace.iter().map(|s| State { pc: M31(s[0]), fp: M31(s[1]) }).collect(),
    };
    let r = &case.ranges;
    let ranges = PublicAddressRanges { program: r[0]..r[1], input: r[2]..r[3], output: r[4]..r[5] };
    import_from_runner_output(segment, ranges).expect("import_from_runner_output")
}

// ---- the dump -----------------------------------------------------------------------------------------------------------
fn hex(b: &[u8]) -> String {
    b.iter().map(|x| format!("{x:02x}")).collect()
}

fn golden_dir() -> PathBuf {
    std::env::var("CAIROM_GOLDEN_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from(env!("CARGO_MANIFEST_DIR")).join("../../tests/golden"))
}

fn dump(name: &str, mut input: ProverInput) {
    // the flattening BEFORE proving (prove_cairo_m drains the bundles), memory rows in the order the reference's HashMaps
    // iterate — the maps are not touched between here and the prover's own iteration, so it sees the same order
    let flat = Flat::snapshot(&input, MemoryOrder::AsIterated);
    TRANSCRIPT.lock().unwrap().clear();
    let proof = prove_cairo_m::<LoggingMerkleChannel>(&mut input, None).expect("prove_cairo_m");
    let transcript: Vec<String> = TRANSCRIPT
        .lock()
        .unwrap()
        .iter()
        .map(|e| {
            format!(
                "{{\"op\":\"{}\",\"digest\":\"{}\",\"n_words\":{},\"words\":[{}]}}",
                e.op,
                hex(&e.digest),
                e.n_words,
                e.words.iter().map(|w| w.to_string()).collect::<Vec<_>>().join(",")
            )
        })
        .collect();
    let commitments: Vec<String> = proof.stark_proof.commitments.iter().map(|c| format!("\"{}\"", hex(&c.0))).collect();
    let proof_json = sonic_rs::to_string(&proof).expect("serialise proof");
    let interaction_pow = proof.interaction_pow;
    // the reference's own verifier must accept what it just produced (with the logging channel, then with the plain one)
    let again: cairo_m_prover::Proof<Blake2sMerkleHasher> = sonic_rs::from_str(&proof_json).expect("round trip");
    verify_cairo_m::<LoggingMerkleChannel>(proof, None).expect("verify (logging channel)");
    verify_cairo_m::<Blake2sMerkleChannel>(again, None).expect("verify (Blake2sMerkleChannel)");
    let c = REGULAR_96_BITS;
    let out = format!(
        "{{\"name\":\"{name}\",\"source\":\"reference\",\"stwo_rev\":\"ab57a1c\",\"pcs_config\":[{},{},{},{}],\n\"input\":{},\n\"transcript\":[{}],\n\"commitments\":[{}],\"interaction_pow\":{},\"verified\":true,\n\"proof\":{}}}\n",
        c.pow_bits,
        c.fri_config.log_blowup_factor,
        c.fri_config.log_last_layer_degree_bound,
        c.fri_config.n_queries,
        flat.to_json(),
        transcript.join(",\n"),
        commitments.join(","),
        interaction_pow,
        proof_json
    );
    let path = golden_dir().join(format!("ref_{name}.json"));
    std::fs::write(&path, out).expect("write golden");
    println!("wrote {} ({} transcript steps)", path.display(), transcript.len());
}

#[test]
fn dump_reference_goldens() {
    dump("unchanged_memory", unchanged_memory_input());
    let cases = golden_dir().join("cases");
    let mut files: Vec<PathBuf> = std::fs::read_dir(&cases)
        .map(|d| d.filter_map(|e| e.ok().map(|e| e.path())).filter(|p| p.to_string_lossy().ends_with(".case.json")).collect())
        .unwrap_or_default();
    files.sort();
    for f in files {
        let text = std::fs::read_to_string(&f).expect("read case");
        let case: CaseFile = sonic_rs::from_str(&text).expect("parse case");
        let input = case_input(&case);
        dump(&case.name, input);
    }
}

// ---- per-op vectors: ref_ops.json -----------------------------------------------------------------------------------------------
// The whole-proof goldens above localise a disagreement to a transcript step; these localise it to ONE Stwo backend operation on a
// seeded input the HIP repository regenerates itself (tests/test_ref_ops.py: `lcg` below, same constants): interpolate, evaluate on
// the double domain, eval_at_point, the Merkle root of a mixed-degree tree, fold_line, fold_circle_into_line, grind, and the order
// in which ColumnSampleBatch::new_vec groups sample points (framing switch `sample_batch`).  Stwo signatures as of `ab57a1c`, from
// memory — this file has never been compiled (see the header); adjust the calls, keep the JSON keys.
use stwo_prover::core::backend::Column;
use stwo_prover::core::circle::{CirclePoint, Coset, SECURE_FIELD_CIRCLE_GEN};
use stwo_prover::core::fields::secure_column::SecureColumnByCoords;
use stwo_prover::core::fri::FriOps;
use stwo_prover::core::pcs::quotients::{ColumnSampleBatch, PointSample};
use stwo_prover::core::poly::circle::{CanonicCoset, CircleEvaluation, PolyOps, SecureEvaluation};
use stwo_prover::core::poly::line::{LineDomain, LineEvaluation};
use stwo_prover::core::poly::BitReversedOrder;
use stwo_prover::core::vcs::prover::MerkleProver;

const P31: u64 = (1 << 31) - 1;
/// x <- x * 6364136223846793005 + 1442695040888963407 (mod 2^64); value = (x >> 33) mod (2^31 - 1)
fn lcg(seed: u64, n: usize) -> Vec<M31> {
    let mut s = seed;
    (0..n)
        .map(|_| {
            s = s.wrapping_mul(6364136223846793005).wrapping_add(1442695040888963407);
            M31::from_u32_unchecked(((s >> 33) % P31) as u32)
        })
        .collect()
}
fn words(v: &[M31]) -> String {
    v.iter().map(|x| x.0.to_string()).collect::<Vec<_>>().join(",")
}
fn qwords(q: SecureField) -> String {
    let a = q.to_m31_array();
    format!("[{},{},{},{}]", a[0].0, a[1].0, a[2].0, a[3].0)
}
fn secure_col(seed: u64, n: usize) -> SecureColumnByCoords<SimdBackend> {
    let mut c = SecureColumnByCoords::<SimdBackend>::zeros(n);
    let v = lcg(seed, 4 * n);
    for i in 0..n {
        c.set(i, SecureField::from_m31_array([v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]]));
    }
    c
}
fn secure_words(c: &SecureColumnByCoords<SimdBackend>) -> String {
    // coordinate-major: 4 columns of n words, like cm_fri_fold_line's four handles
    (0..4).map(|k| format!("[{}]", words(&c.columns[k].to_cpu()))).collect::<Vec<_>>().join(",")
}

#[test]
fn dump_reference_ops() {
    const LOG: u32 = 6;
    let n = 1usize << LOG;
    let domain = CanonicCoset::new(LOG).circle_domain();
    let big = CanonicCoset::new(LOG + 1).circle_domain();
    let twiddles = SimdBackend::precompute_twiddles(big.half_coset);
    // interpolate / evaluate / eval_at_point: seed 1
    let eval = CircleEvaluation::<SimdBackend, M31, BitReversedOrder>::new(domain, lcg(1, n).into_iter().collect());
    let poly = eval.interpolate_with_twiddles(&twiddles);
    let coeffs = poly.coeffs.to_cpu();
    let lde = poly.evaluate_with_twiddles(big, &twiddles).values.to_cpu();
    let pt = SECURE_FIELD_CIRCLE_GEN.mul(7);
    let at = poly.eval_at_point(pt);
    // Merkle: columns of 2^5, 2^5 and 2^3 rows (seeds 2, 3, 4): the mixed-degree root
    let (ca, cb, cc) = (lcg(2, 32), lcg(3, 32), lcg(4, 8));
    let cols: Vec<<SimdBackend as stwo_prover::core::backend::ColumnOps<M31>>::Column> =
        vec![ca.iter().copied().collect(), cb.iter().copied().collect(), cc.iter().copied().collect()];
    let tree = MerkleProver::<SimdBackend, Blake2sMerkleHasher>::commit(cols.iter().collect());
    // FRI folds: a line evaluation of 2^LOG points (seed 5) folded once; a circle evaluation of 2^LOG points (seed 6) folded into
    // a zero line of 2^(LOG-1); alpha = the four words of seed 7
    let a = lcg(7, 4);
    let alpha = SecureField::from_m31_array([a[0], a[1], a[2], a[3]]);
    let line = LineEvaluation::<SimdBackend>::new(LineDomain::new(Coset::half_odds(LOG)), secure_col(5, n));
    let folded = SimdBackend::fold_line(&line, alpha, &twiddles);
    let src = SecureEvaluation::<SimdBackend, BitReversedOrder>::new(domain, secure_col(6, n));
    let mut dst = LineEvaluation::<SimdBackend>::new_zero(LineDomain::new(domain.half_coset));
    SimdBackend::fold_circle_into_line(&mut dst, &src, alpha, &twiddles);
    // grind: a fresh channel after mix_u64(0x0123456789abcdef), 10 bits
    let mut ch = Blake2sChannel::default();
    ch.mix_u64(0x0123456789abcdef);
    let digest_before = ch.digest();
    let nonce = <SimdBackend as GrindOps<Blake2sChannel>>::grind(&ch, 10);
    // ColumnSampleBatch::new_vec: column 0 sampled at p1, column 1 at [p2, p1], column 2 at p1 — batches in Stwo's order
    let (p1, p2) = (SECURE_FIELD_CIRCLE_GEN.mul(5), SECURE_FIELD_CIRCLE_GEN.mul(3));
    let v = SecureField::from_m31_array([a[0], a[0], a[0], a[0]]);
    let s0 = vec![PointSample { point: p1, value: v }];
    let s1 = vec![PointSample { point: p2, value: v }, PointSample { point: p1, value: v }];
    let s2 = vec![PointSample { point: p1, value: v }];
    let batches = ColumnSampleBatch::new_vec(&[&s0, &s1, &s2]);
    let batch_json: Vec<String> = batches
        .iter()
        .map(|b| {
            format!(
                "{{\"point_x\":{},\"point_y\":{},\"columns\":[{}]}}",
                qwords(b.point.x),
                qwords(b.point.y),
                b.columns_and_values.iter().map(|(c, _)| c.to_string()).collect::<Vec<_>>().join(",")
            )
        })
        .collect();
    let out = format!(
        "{{\"source\":\"reference\",\"stwo_rev\":\"ab57a1c\",\"log\":{LOG},\n\"interpolate\":[{}],\n\"evaluate\":[{}],\n\"eval_at_point\":{{\"x\":{},\"y\":{},\"value\":{}}},\n\"merkle_root\":\"{}\",\n\"alpha\":{},\n\"fold_line\":[{}],\n\"fold_circle_into_line\":[{}],\n\"grind\":{{\"digest\":\"{}\",\"bits\":10,\"nonce\":{}}},\n\"sample_batches\":{{\"p1_x\":{},\"p2_x\":{},\"batches\":[{}]}}}}\n",
        words(&coeffs),
        words(&lde),
        qwords(pt.x),
        qwords(pt.y),
        qwords(at),
        hex(&tree.root().0),
        qwords(alpha),
        secure_words(&folded.values),
        secure_words(&dst.values),
        hex(&digest_before.0),
        nonce,
        qwords(p1.x),
        qwords(p2.x),
        batch_json.join(",")
    );
    let path = golden_dir().join("ref_ops.json");
    std::fs::write(&path, out).expect("write ref_ops.json");
    println!("wrote {}", path.display());
}
