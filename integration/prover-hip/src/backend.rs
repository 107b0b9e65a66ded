//! `HipBackend` — Stwo's backend-trait surface (`stwo_prover::core::backend`) over the per-op C ABI of libcairom_hip.so
//! (include/cairom_hip.h), the way `SimdBackend` sits behind `prove_cairo_m` today:
//!
//! ```text
//! prover.rs:28   where SimdBackend: BackendForChannel<MC>          ->  HipBackend: BackendForChannel<Blake2sMerkleChannel>
//! prover.rs:56   SimdBackend::precompute_twiddles(..)              ->  PolyOps::precompute_twiddles      = cm_twiddles_precompute
//! prover.rs:71   tree_builder.extend_evals(..) / .commit(channel)   ->  PolyOps::{interpolate, evaluate}  = cm_interpolate, cm_evaluate
//!                                                                      MerkleOps::commit_on_layer         = cm_merkle_commit_layer
//! prover.rs:90   SimdBackend::grind(channel, pow_bits)              ->  GrindOps::grind                    = cm_grind
//! prover.rs:131  prove::<SimdBackend, _>(..)                        ->  eval_at_point, accumulate_quotients, fold_*, decompose,
//!                                                                      accumulate, generate_secure_powers, batch_inverse,
//!                                                                      bit_reverse_column = the cm_* function of the same name
//! ```
//!
//! Columns live in HBM: `HipColumn<BaseField>` is one `cm_handle` (u32[len] of canonical M31 values, evaluations in Stwo's
//! BitReversedOrder); a secure column is four of them (`SecureColumnByCoords` already is, upstream); a hash column is 8 u32 per
//! node.  Every trait call is one library call on the caller's stream 0 and returns when the kernel has finished, so the
//! Rust side may hand the handle to the next op or read it back (`to_cpu`) immediately; rayon workers may call concurrently
//! (the library is re-entrant, cairom_hip.h "Conventions").
//!
//! What these traits do NOT reach: Cairo-M's components implement `ComponentProver<SimdBackend>` through
//! `FrameworkComponent` and name `SimdBackend` / `PackedM31` in every `write_trace`, so trace generation, the LogUp
//! interaction trace and constraint evaluation cannot be routed through a `Backend` parameter without changing the
//! reference.  They are exported separately (`cm_trace_write`, `cm_histogram`, `cm_interaction_write`,
//! `cm_constraints_accumulate`; wrappers at the bottom of this file), and the whole-path twin `prove_cairo_m_hip` (lib.rs)
//! uses the fused driver instead of stitching ops together.  `HipBackend` is the drop-in for code that IS generic over the
//! backend: Stwo's own `CommitmentSchemeProver`, `FriProver`, quotient and accumulation code.
//!
//! Shipped as source: the HIP repository's image has no Rust toolchain, this file has never been compiled there.  The C side of
//! every call below is exercised through the same entry points by tests/test_gpu_poly_merkle.py, test_gpu_fri_quotients.py and
//! test_gpu_components.py, and tests/test_rust_shim.py keeps ffi.rs in step with the header.
#![allow(clippy::missing_safety_doc)]
use std::fmt::Debug;
use std::marker::PhantomData;

use stwo_prover::core::air::accumulation::AccumulationOps;
use stwo_prover::core::backend::cpu::CpuBackend;
use stwo_prover::core::backend::{Backend, BackendForChannel, Col, Column, ColumnOps};
use stwo_prover::core::channel::Blake2sChannel;
use stwo_prover::core::circle::{CirclePoint, Coset};
use stwo_prover::core::fields::m31::BaseField;
use stwo_prover::core::fields::qm31::SecureField;
use stwo_prover::core::fields::secure_column::{SECURE_EXTENSION_DEGREE, SecureColumnByCoords};
use stwo_prover::core::fields::FieldOps;
use stwo_prover::core::fri::FriOps;
use stwo_prover::core::lookups::gkr_prover::{GkrMultivariatePolyOracle, GkrOps, Layer};
use stwo_prover::core::lookups::mle::{Mle, MleOps};
use stwo_prover::core::lookups::utils::UnivariatePoly;
use stwo_prover::core::pcs::quotients::{ColumnSampleBatch, QuotientOps};
use stwo_prover::core::poly::circle::{CanonicCoset, CircleDomain, CircleEvaluation, CirclePoly, PolyOps, SecureEvaluation};
use stwo_prover::core::poly::line::LineEvaluation;
use stwo_prover::core::poly::twiddles::TwiddleTree;
use stwo_prover::core::poly::BitReversedOrder;
use stwo_prover::core::proof_of_work::GrindOps;
use stwo_prover::core::vcs::blake2_hash::Blake2sHash;
use stwo_prover::core::vcs::blake2_merkle::{Blake2sMerkleChannel, Blake2sMerkleHasher};
use stwo_prover::core::vcs::ops::MerkleOps;

use crate::ffi::*;

fn ck(rc: i32, what: &str) {
    if rc != 0 {
        panic!("libcairom_hip: {what}: status {rc}: {}", crate::last_error());
    }
}

#[derive(Copy, Clone, Debug, Default)]
pub struct HipBackend;
impl Backend for HipBackend {}
impl BackendForChannel<Blake2sMerkleChannel> for HipBackend {}

// ---- columns ----------------------------------------------------------------------------------------------------------
/// `words_per_item` u32 words per element in one device allocation: 1 for `BaseField`, 8 for `Blake2sHash`.
pub struct HipColumn<T> {
    handle: cm_handle,
    len: usize,
    _t: PhantomData<T>,
}
unsafe impl<T> Send for HipColumn<T> {}
unsafe impl<T> Sync for HipColumn<T> {}

/// How a column element crosses the boundary as u32 words.
pub trait Words: Copy {
    const N: usize;
    fn to_words(&self, out: &mut [u32]);
    fn from_words(w: &[u32]) -> Self;
}
impl Words for BaseField {
    const N: usize = 1;
    fn to_words(&self, out: &mut [u32]) {
        out[0] = self.0;
    }
    fn from_words(w: &[u32]) -> Self {
        BaseField::from_u32_unchecked(w[0])
    }
}
impl Words for Blake2sHash {
    const N: usize = 8;
    fn to_words(&self, out: &mut [u32]) {
        for (o, c) in out.iter_mut().zip(self.0.chunks_exact(4)) {
            *o = u32::from_le_bytes(c.try_into().unwrap());
        }
    }
    fn from_words(w: &[u32]) -> Self {
        let mut b = [0u8; 32];
        for (c, x) in b.chunks_exact_mut(4).zip(w) {
            c.copy_from_slice(&x.to_le_bytes());
        }
        Blake2sHash(b)
    }
}

impl<T: Words> HipColumn<T> {
    pub fn handle(&self) -> cm_handle {
        self.handle
    }
    fn alloc(len: usize) -> Self {
        crate::ensure_init();
        let mut h: cm_handle = 0;
        ck(unsafe { cm_col_alloc((len * T::N) as u64, &mut h) }, "cm_col_alloc");
        Self { handle: h, len, _t: PhantomData }
    }
    pub fn from_host(values: &[T]) -> Self {
        let c = Self::alloc(values.len());
        let mut w = vec![0u32; values.len() * T::N];
        for (v, chunk) in values.iter().zip(w.chunks_exact_mut(T::N)) {
            v.to_words(chunk);
        }
        ck(unsafe { cm_col_h2d(c.handle, w.as_ptr(), w.len() as u64, 0) }, "cm_col_h2d");
        c
    }
}
impl<T> Drop for HipColumn<T> {
    fn drop(&mut self) {
        unsafe { cm_col_free(self.handle) };
    }
}
impl<T: Words> Clone for HipColumn<T> {
    fn clone(&self) -> Self {
        let c = Self::alloc(self.len);
        ck(unsafe { cm_col_copy(c.handle, self.handle, (self.len * T::N) as u64, 0) }, "cm_col_copy");
        c
    }
}
impl<T> Debug for HipColumn<T> {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "HipColumn(len = {}, handle = {:#x})", self.len, self.handle)
    }
}
impl<T: Words> FromIterator<T> for HipColumn<T> {
    fn from_iter<I: IntoIterator<Item = T>>(iter: I) -> Self {
        let v: Vec<T> = iter.into_iter().collect();
        Self::from_host(&v)
    }
}
impl<T: Words + Debug> Column<T> for HipColumn<T> {
    fn zeros(len: usize) -> Self {
        let c = Self::alloc(len);
        ck(unsafe { cm_col_zero(c.handle, (len * T::N) as u64, 0) }, "cm_col_zero");
        c
    }
    unsafe fn uninitialized(len: usize) -> Self {
        Self::alloc(len)
    }
    fn to_cpu(&self) -> Vec<T> {
        let mut w = vec![0u32; self.len * T::N];
        ck(unsafe { cm_col_d2h(self.handle, w.as_mut_ptr(), w.len() as u64, 0) }, "cm_col_d2h");
        w.chunks_exact(T::N).map(T::from_words).collect()
    }
    fn len(&self) -> usize {
        self.len
    }
    fn at(&self, index: usize) -> T {
        let mut w = vec![0u32; T::N];
        ck(unsafe { cm_col_read(self.handle, (index * T::N) as u64, w.as_mut_ptr(), T::N as u64, 0) }, "cm_col_read");
        T::from_words(&w)
    }
    fn set(&mut self, index: usize, value: T) {
        let mut w = vec![0u32; T::N];
        value.to_words(&mut w);
        ck(unsafe { cm_col_write(self.handle, (index * T::N) as u64, w.as_ptr(), T::N as u64, 0) }, "cm_col_write");
    }
}

/// `Column<SecureField>`: four coordinate columns (the layout `SecureColumnByCoords` has upstream).
#[derive(Clone, Debug)]
pub struct HipSecureColumn {
    pub coords: [HipColumn<BaseField>; SECURE_EXTENSION_DEGREE],
}
impl HipSecureColumn {
    fn handles(&self) -> [cm_handle; 4] {
        [self.coords[0].handle, self.coords[1].handle, self.coords[2].handle, self.coords[3].handle]
    }
}
impl FromIterator<SecureField> for HipSecureColumn {
    fn from_iter<I: IntoIterator<Item = SecureField>>(iter: I) -> Self {
        let v: Vec<SecureField> = iter.into_iter().collect();
        let coord = |k: usize| HipColumn::from_host(&v.iter().map(|x| x.to_m31_array()[k]).collect::<Vec<_>>());
        Self { coords: [coord(0), coord(1), coord(2), coord(3)] }
    }
}
impl Column<SecureField> for HipSecureColumn {
    fn zeros(len: usize) -> Self {
        Self { coords: std::array::from_fn(|_| <HipColumn<BaseField> as Column<BaseField>>::zeros(len)) }
    }
    unsafe fn uninitialized(len: usize) -> Self {
        Self { coords: std::array::from_fn(|_| unsafe { <HipColumn<BaseField> as Column<BaseField>>::uninitialized(len) }) }
    }
    fn to_cpu(&self) -> Vec<SecureField> {
        let c: Vec<Vec<BaseField>> = self.coords.iter().map(|c| c.to_cpu()).collect();
        (0..self.len()).map(|i| SecureField::from_m31_array([c[0][i], c[1][i], c[2][i], c[3][i]])).collect()
    }
    fn len(&self) -> usize {
        self.coords[0].len
    }
    fn at(&self, index: usize) -> SecureField {
        SecureField::from_m31_array(std::array::from_fn(|k| self.coords[k].at(index)))
    }
    fn set(&mut self, index: usize, value: SecureField) {
        let v = value.to_m31_array();
        for k in 0..4 {
            self.coords[k].set(index, v[k]);
        }
    }
}

impl ColumnOps<BaseField> for HipBackend {
    type Column = HipColumn<BaseField>;
    fn bit_reverse_column(column: &mut Self::Column) {
        let h = [column.handle];
        ck(unsafe { cm_bit_reverse(h.as_ptr(), 1, column.len.ilog2(), 0) }, "cm_bit_reverse");
    }
}
impl ColumnOps<SecureField> for HipBackend {
    type Column = HipSecureColumn;
    fn bit_reverse_column(column: &mut Self::Column) {
        let h = column.handles();
        ck(unsafe { cm_bit_reverse(h.as_ptr(), 4, column.len().ilog2(), 0) }, "cm_bit_reverse");
    }
}
impl ColumnOps<Blake2sHash> for HipBackend {
    type Column = HipColumn<Blake2sHash>;
    fn bit_reverse_column(_column: &mut Self::Column) {
        unimplemented!("Stwo never bit-reverses a hash column")
    }
}
impl FieldOps<BaseField> for HipBackend {
    fn batch_inverse(column: &Self::Column, dst: &mut Self::Column) {
        ck(unsafe { cm_batch_inverse_m31(column.handle, dst.handle, column.len as u64, 0) }, "cm_batch_inverse_m31");
    }
}
impl FieldOps<SecureField> for HipBackend {
    fn batch_inverse(column: &HipSecureColumn, dst: &mut HipSecureColumn) {
        let (i, o) = (column.handles(), dst.handles());
        ck(unsafe { cm_batch_inverse_qm31(i.as_ptr(), o.as_ptr(), column.len() as u64, 0) }, "cm_batch_inverse_qm31");
    }
}

fn secure_handles(c: &SecureColumnByCoords<HipBackend>) -> [cm_handle; 4] {
    [c.columns[0].handle, c.columns[1].handle, c.columns[2].handle, c.columns[3].handle]
}
fn words(f: SecureField) -> [u32; 4] {
    f.to_m31_array().map(|m| m.0)
}

// ---- PolyOps ------------------------------------------------------------------------------------------------------------
/// The library keeps one table set per `cm_twiddles_precompute` (tables for `CanonicCoset(log)` and every smaller canonic
/// domain); `TwiddleTree::twiddles` / `itwiddles` both carry the handle.
pub struct HipTwiddles {
    pub handle: cm_handle,
    pub log_size: u32,
}
impl Drop for HipTwiddles {
    fn drop(&mut self) {
        unsafe { cm_twiddles_free(self.handle) };
    }
}
unsafe impl Send for HipTwiddles {}
unsafe impl Sync for HipTwiddles {}

impl PolyOps for HipBackend {
    type Twiddles = std::sync::Arc<HipTwiddles>;

    fn new_canonical_ordered(coset: CanonicCoset, values: Col<Self, BaseField>) -> CircleEvaluation<Self, BaseField, BitReversedOrder> {
        // natural (canonic-coset) order -> circle-domain order -> bit-reversed storage.  Cairo-M never calls this (its traces
        // are written straight into bit-reversed columns, components/mod.rs:168-177); done on the host for completeness.
        let cpu = CpuBackend::new_canonical_ordered(coset, values.to_cpu());
        CircleEvaluation::new(cpu.domain, HipColumn::from_host(&cpu.values))
    }

    fn interpolate(eval: CircleEvaluation<Self, BaseField, BitReversedOrder>, itwiddles: &TwiddleTree<Self>) -> CirclePoly<Self> {
        let log = eval.domain.log_size();
        let h = [eval.values.handle];
        ck(unsafe { cm_interpolate(h.as_ptr(), 1, log, itwiddles.itwiddles.handle, 0) }, "cm_interpolate");
        CirclePoly::new(eval.values)
    }

    fn interpolate_columns(
        columns: impl IntoIterator<Item = CircleEvaluation<Self, BaseField, BitReversedOrder>>,
        twiddles: &TwiddleTree<Self>,
    ) -> Vec<CirclePoly<Self>> {
        // one launch per size group instead of one per column (tree_builder.extend_evals, prover.rs:72, 81, 101)
        let evals: Vec<_> = columns.into_iter().collect();
        let mut by_log: std::collections::BTreeMap<u32, Vec<cm_handle>> = Default::default();
        for e in &evals {
            by_log.entry(e.domain.log_size()).or_default().push(e.values.handle);
        }
        for (log, hs) in &by_log {
            ck(unsafe { cm_interpolate(hs.as_ptr(), hs.len() as u32, *log, twiddles.itwiddles.handle, 0) }, "cm_interpolate");
        }
        evals.into_iter().map(|e| CirclePoly::new(e.values)).collect()
    }

    fn eval_at_point(poly: &CirclePoly<Self>, point: CirclePoint<SecureField>) -> SecureField {
        let h = [poly.coeffs.handle];
        let (x, y) = (words(point.x), words(point.y));
        let pt = [x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]];
        let mut out = [0u32; 4];
        ck(unsafe { cm_eval_at_point(h.as_ptr(), 1, poly.log_size(), pt.as_ptr(), out.as_mut_ptr(), 0) }, "cm_eval_at_point");
        SecureField::from_u32_unchecked(out[0], out[1], out[2], out[3])
    }

    fn extend(poly: &CirclePoly<Self>, log_size: u32) -> CirclePoly<Self> {
        // coefficients are stored so that extension is zero padding
        let n = poly.coeffs.len;
        let out = <HipColumn<BaseField> as Column<BaseField>>::zeros(1 << log_size);
        ck(unsafe { cm_col_copy(out.handle, poly.coeffs.handle, n as u64, 0) }, "cm_col_copy");
        CirclePoly::new(out)
    }

    fn evaluate(poly: &CirclePoly<Self>, domain: CircleDomain, twiddles: &TwiddleTree<Self>) -> CircleEvaluation<Self, BaseField, BitReversedOrder> {
        let out = unsafe { <HipColumn<BaseField> as Column<BaseField>>::uninitialized(domain.size()) };
        let (c, o) = ([poly.coeffs.handle], [out.handle]);
        ck(
            unsafe { cm_evaluate(c.as_ptr(), 1, poly.log_size(), domain.log_size(), twiddles.twiddles.handle, o.as_ptr(), 0) },
            "cm_evaluate",
        );
        CircleEvaluation::new(domain, out)
    }

    fn evaluate_polynomials(
        polynomials: &stwo_prover::core::ColumnVec<CirclePoly<Self>>,
        log_blowup_factor: u32,
        twiddles: &TwiddleTree<Self>,
    ) -> Vec<CircleEvaluation<Self, BaseField, BitReversedOrder>> {
        // one launch per size group (tree_builder.commit, prover.rs:73, 82, 102)
        let outs: Vec<HipColumn<BaseField>> = polynomials
            .iter()
            .map(|p| unsafe { <HipColumn<BaseField> as Column<BaseField>>::uninitialized(1 << (p.log_size() + log_blowup_factor)) })
            .collect();
        let mut by_log: std::collections::BTreeMap<u32, (Vec<cm_handle>, Vec<cm_handle>)> = Default::default();
        for (p, o) in polynomials.iter().zip(&outs) {
            let e = by_log.entry(p.log_size()).or_default();
            e.0.push(p.coeffs.handle);
            e.1.push(o.handle);
        }
        for (log, (c, o)) in &by_log {
            ck(
                unsafe { cm_evaluate(c.as_ptr(), c.len() as u32, *log, *log + log_blowup_factor, twiddles.twiddles.handle, o.as_ptr(), 0) },
                "cm_evaluate",
            );
        }
        polynomials
            .iter()
            .zip(outs)
            .map(|(p, o)| CircleEvaluation::new(CanonicCoset::new(p.log_size() + log_blowup_factor).circle_domain(), o))
            .collect()
    }

    fn precompute_twiddles(coset: Coset) -> TwiddleTree<Self> {
        // prover.rs:56-60 passes CanonicCoset(k).circle_domain().half_coset, i.e. a coset of log size k - 1
        crate::ensure_init();
        let log_size = coset.log_size() + 1;
        let mut h: cm_handle = 0;
        ck(unsafe { cm_twiddles_precompute(log_size, &mut h) }, "cm_twiddles_precompute");
        let t = std::sync::Arc::new(HipTwiddles { handle: h, log_size });
        TwiddleTree { root_coset: coset, twiddles: t.clone(), itwiddles: t }
    }
}

// ---- QuotientOps / FriOps / AccumulationOps -----------------------------------------------------------------------------
impl QuotientOps for HipBackend {
    fn accumulate_quotients(
        domain: CircleDomain,
        columns: &[&CircleEvaluation<Self, BaseField, BitReversedOrder>],
        random_coeff: SecureField,
        sample_batches: &[ColumnSampleBatch],
        log_blowup_factor: u32,
    ) -> SecureEvaluation<Self, BitReversedOrder> {
        let _ = log_blowup_factor; // the kernel evaluates the denominators per row of `domain` directly
        let cols: Vec<cm_handle> = columns.iter().map(|c| c.values.handle).collect();
        let (mut points, mut off, mut idx, mut vals) = (Vec::new(), vec![0u32], Vec::new(), Vec::new());
        for b in sample_batches {
            points.extend(words(b.point.x));
            points.extend(words(b.point.y));
            for (col, v) in &b.columns_and_values {
                idx.push(*col as u32);
                vals.extend(words(*v));
            }
            off.push(idx.len() as u32);
        }
        let batches = cm_sample_batches {
            n_batches: sample_batches.len() as u32,
            points: points.as_ptr(),
            batch_off: off.as_ptr(),
            col_index: idx.as_ptr(),
            values: vals.as_ptr(),
        };
        let out = SecureColumnByCoords::<Self>::zeros(domain.size());
        let oh = secure_handles(&out);
        let rc = words(random_coeff);
        // twiddle tables for the point coordinates of `domain`: kept per process for this size
        let tw = domain_twiddles(domain.log_size());
        ck(
            unsafe {
                cm_accumulate_quotients(domain.log_size(), cols.as_ptr(), cols.len() as u32, &batches, rc.as_ptr(), oh.as_ptr(), tw.handle, 0)
            },
            "cm_accumulate_quotients",
        );
        SecureEvaluation::new(domain, out)
    }
}
/// `accumulate_quotients` receives no `TwiddleTree`: keep one per process that covers the largest domain seen.
fn domain_twiddles(log_size: u32) -> std::sync::Arc<HipTwiddles> {
    use std::sync::{Arc, Mutex, OnceLock};
    static CACHE: OnceLock<Mutex<Option<Arc<HipTwiddles>>>> = OnceLock::new();
    let mut g = CACHE.get_or_init(|| Mutex::new(None)).lock().unwrap();
    if g.as_ref().map(|t| t.log_size < log_size + 1).unwrap_or(true) {
        let mut h: cm_handle = 0;
        ck(unsafe { cm_twiddles_precompute(log_size + 1, &mut h) }, "cm_twiddles_precompute");
        *g = Some(Arc::new(HipTwiddles { handle: h, log_size: log_size + 1 }));
    }
    g.as_ref().unwrap().clone()
}

impl FriOps for HipBackend {
    fn fold_line(eval: &LineEvaluation<Self>, alpha: SecureField, twiddles: &TwiddleTree<Self>) -> LineEvaluation<Self> {
        let log_n = eval.len().ilog2();
        let out = SecureColumnByCoords::<Self>::zeros(eval.len() / 2);
        let (i, o, a) = (secure_handles(&eval.values), secure_handles(&out), words(alpha));
        ck(unsafe { cm_fri_fold_line(i.as_ptr(), a.as_ptr(), log_n, twiddles.itwiddles.handle, o.as_ptr(), 0) }, "cm_fri_fold_line");
        LineEvaluation::new(eval.domain().double(), out)
    }
    fn fold_circle_into_line(
        dst: &mut LineEvaluation<Self>,
        src: &SecureEvaluation<Self, BitReversedOrder>,
        alpha: SecureField,
        twiddles: &TwiddleTree<Self>,
    ) {
        let log_n = src.domain.log_size();
        let (d, s, a) = (secure_handles(&dst.values), secure_handles(&src.values), words(alpha));
        ck(
            unsafe { cm_fri_fold_circle_into_line(d.as_ptr(), s.as_ptr(), a.as_ptr(), log_n, twiddles.itwiddles.handle, 0) },
            "cm_fri_fold_circle_into_line",
        );
    }
    fn decompose(eval: &SecureEvaluation<Self, BitReversedOrder>) -> (SecureEvaluation<Self, BitReversedOrder>, SecureField) {
        let g = eval.values.clone();
        let (h, mut lambda) = (secure_handles(&g), [0u32; 4]);
        ck(unsafe { cm_fri_decompose(h.as_ptr(), eval.domain.log_size(), lambda.as_mut_ptr(), 0) }, "cm_fri_decompose");
        (SecureEvaluation::new(eval.domain, g), SecureField::from_u32_unchecked(lambda[0], lambda[1], lambda[2], lambda[3]))
    }
}

impl AccumulationOps for HipBackend {
    fn accumulate(column: &mut SecureColumnByCoords<Self>, other: &SecureColumnByCoords<Self>) {
        let (d, s) = (secure_handles(column), secure_handles(other));
        ck(unsafe { cm_accumulate(d.as_ptr(), s.as_ptr(), column.len() as u64, 0) }, "cm_accumulate");
    }
    fn generate_secure_powers(felt: SecureField, n_powers: usize) -> Vec<SecureField> {
        let (f, mut out) = (words(felt), vec![0u32; 4 * n_powers]);
        ck(unsafe { cm_generate_secure_powers(f.as_ptr(), n_powers as u64, out.as_mut_ptr()) }, "cm_generate_secure_powers");
        out.chunks_exact(4).map(|w| SecureField::from_u32_unchecked(w[0], w[1], w[2], w[3])).collect()
    }
}

// ---- MerkleOps / GrindOps -----------------------------------------------------------------------------------------------
impl MerkleOps<Blake2sMerkleHasher> for HipBackend {
    fn commit_on_layer(
        log_size: u32,
        prev_layer: Option<&Col<Self, Blake2sHash>>,
        columns: &[&Col<Self, BaseField>],
    ) -> Col<Self, Blake2sHash> {
        let out = unsafe { <HipColumn<Blake2sHash> as Column<Blake2sHash>>::uninitialized(1 << log_size) };
        let cols: Vec<cm_handle> = columns.iter().map(|c| c.handle).collect();
        ck(
            unsafe {
                cm_merkle_commit_layer(log_size, prev_layer.map(|p| p.handle).unwrap_or(0), cols.as_ptr(), cols.len() as u32, out.handle, 0)
            },
            "cm_merkle_commit_layer",
        );
        out
    }
}
impl GrindOps<Blake2sChannel> for HipBackend {
    fn grind(channel: &Blake2sChannel, pow_bits: u32) -> u64 {
        crate::ensure_init();
        let digest = channel.digest().0;
        let mut nonce = 0u64;
        ck(unsafe { cm_grind(digest.as_ptr(), pow_bits, &mut nonce, 0) }, "cm_grind");
        nonce
    }
}

// ---- GkrOps: required by `Backend`, unused by Cairo-M (its LogUp is the plain column form): host round trip --------------
impl MleOps<BaseField> for HipBackend {
    fn fix_first_variable(mle: Mle<Self, BaseField>, assignment: SecureField) -> Mle<Self, SecureField> {
        let cpu = CpuBackend::fix_first_variable(Mle::<CpuBackend, BaseField>::new(mle.into_evals().to_cpu()), assignment);
        Mle::new(cpu.into_evals().into_iter().collect())
    }
}
impl MleOps<SecureField> for HipBackend {
    fn fix_first_variable(mle: Mle<Self, SecureField>, assignment: SecureField) -> Mle<Self, SecureField> {
        let cpu = CpuBackend::fix_first_variable(Mle::<CpuBackend, SecureField>::new(mle.into_evals().to_cpu()), assignment);
        Mle::new(cpu.into_evals().into_iter().collect())
    }
}
impl GkrOps for HipBackend {
    fn gen_eq_evals(y: &[SecureField], v: SecureField) -> Mle<Self, SecureField> {
        Mle::new(CpuBackend::gen_eq_evals(y, v).into_evals().into_iter().collect())
    }
    fn next_layer(_layer: &Layer<Self>) -> Layer<Self> {
        unimplemented!("GKR lookups are not on Cairo-M's path (prove_cairo_m never builds a GKR layer)")
    }
    fn sum_as_poly_in_first_variable(_h: &GkrMultivariatePolyOracle<'_, Self>, _claim: SecureField) -> UnivariatePoly<SecureField> {
        unimplemented!("GKR lookups are not on Cairo-M's path")
    }
}

// ---- per-component AIR ops (not reachable through the Backend traits: SURVEY 8b) -------------------------------------------
/// Device-resident `ProverInput` (cm_input_upload) + the Cairo-M component operations on caller-owned columns.
pub struct DeviceInput(pub *mut cm_device_input);
impl Drop for DeviceInput {
    fn drop(&mut self) {
        unsafe { cm_input_free(self.0) };
    }
}
impl DeviceInput {
    /// `<component>::Claim::write_trace` (e.g. opcodes/store_fp_imm.rs:147-296): fills `cols` (the component's trace columns,
    /// 2^log_size rows each, padding rows included).  Component ids: cairom_hip.h.
    pub fn write_trace(&self, component: i32, cols: &[HipColumn<BaseField>]) {
        let h: Vec<cm_handle> = cols.iter().map(|c| c.handle).collect();
        ck(unsafe { cm_trace_write(self.0, component, h.as_ptr(), 0) }, "cm_trace_write");
    }
    pub fn log_size(&self, component: i32) -> u32 {
        let mut l = 0u32;
        ck(unsafe { cm_component_log_size(self.0, component, &mut l) }, "cm_component_log_size");
        l
    }
}
/// `<component>::InteractionClaim::write_interaction_trace` + `LogupTraceGenerator::finalize_last`: returns the claimed sum.
pub fn write_interaction_trace(
    component: i32,
    trace: &[HipColumn<BaseField>],
    preprocessed: &[HipColumn<BaseField>; CM_N_PREPROCESSED],
    log_size: u32,
    relations: &cm_relations,
    out: &[HipColumn<BaseField>],
) -> SecureField {
    let (t, p, o): (Vec<_>, Vec<_>, Vec<_>) =
        (trace.iter().map(|c| c.handle).collect(), preprocessed.iter().map(|c| c.handle).collect(), out.iter().map(|c| c.handle).collect());
    let mut cs = [0u32; 4];
    ck(
        unsafe { cm_interaction_write(component, t.as_ptr(), p.as_ptr(), log_size, relations, o.as_ptr(), cs.as_mut_ptr(), 0) },
        "cm_interaction_write",
    );
    SecureField::from_u32_unchecked(cs[0], cs[1], cs[2], cs[3])
}
/// `FrameworkComponent::evaluate_constraint_quotients_on_domain` of one component into `acc` (the size group's accumulator).
#[allow(clippy::too_many_arguments)]
pub fn accumulate_constraint_quotients(
    component: i32,
    trace_lde: &[HipColumn<BaseField>],
    interaction_lde: &[HipColumn<BaseField>],
    preprocessed_lde: &[HipColumn<BaseField>; CM_N_PREPROCESSED],
    log_size: u32,
    relations: &cm_relations,
    coeff_powers: &[SecureField],
    claimed_sum: SecureField,
    acc: &mut SecureColumnByCoords<HipBackend>,
) {
    let (t, i, p): (Vec<_>, Vec<_>, Vec<_>) = (
        trace_lde.iter().map(|c| c.handle).collect(),
        interaction_lde.iter().map(|c| c.handle).collect(),
        preprocessed_lde.iter().map(|c| c.handle).collect(),
    );
    let cp: Vec<u32> = coeff_powers.iter().flat_map(|f| words(*f)).collect();
    let (cs, a) = (words(claimed_sum), secure_handles(acc));
    ck(
        unsafe {
            cm_constraints_accumulate(component, t.as_ptr(), i.as_ptr(), p.as_ptr(), log_size, relations, cp.as_ptr(), cs.as_ptr(), a.as_ptr(), 0)
        },
        "cm_constraints_accumulate",
    );
}
