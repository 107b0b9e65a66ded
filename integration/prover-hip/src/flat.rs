//! `ProverInput` (crates/prover/src/adapter/mod.rs:27-83) flattened into the C ABI's `cm_prover_input`
//! (include/cairom_hip.h).  Shared by `prove_cairo_m_hip` and by the golden-vector harness (tests/golden_dump.rs), which
//! also serialises the flattening as JSON so that the HIP repository's tests can replay the reference's exact input.
use cairo_m_prover::adapter::{ExecutionBundle, ProverInput};
use num_traits::Zero;
use stwo_prover::core::fields::m31::M31;
use stwo_prover::core::fields::qm31::QM31;

use crate::OPCODE_GROUPS;
use crate::ffi::*;

/// Row order of the memory component.  The reference iterates `HashMap`s (components/memory.rs:104-133): whatever order
/// `initial_memory.iter()` / `final_memory.iter()` yield is the order of the committed rows, so two runs of the reference
/// need not agree with each other.  `AscendingAddress` makes proofs reproducible (what `prove_cairo_m_hip` uses);
/// `AsIterated` records the order the reference itself is about to use — the golden harness dumps that, and the library
/// commits the rows in whatever order it is handed.
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub enum MemoryOrder {
    AscendingAddress,
    AsIterated,
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

/// Owned flattening of a `ProverInput`; `view()` borrows it as the C struct.
pub struct Flat {
    pub bundles: Vec<Vec<cm_bundle>>,
    pub data_accesses: Vec<cm_data_access>,
    pub initial_memory: Vec<cm_memory_cell>,
    pub final_memory: Vec<cm_memory_cell>,
    pub clock_updates: Vec<cm_clock_update>,
    pub initial_tree: Vec<cm_merkle_node>,
    pub final_tree: Vec<cm_merkle_node>,
    pub regs: [u32; 4],
    pub roots: [u32; 2],
    pub ranges: [[u32; 2]; 3],
}

impl Flat {
    /// Consumes the bundles like `prove_cairo_m` does (opcodes/mod.rs:53-58 drains `states_by_opcodes`).
    pub fn new(input: &mut ProverInput, order: MemoryOrder) -> Self {
        let f = Self::snapshot(input, order);
        for states in input.instructions.states_by_opcodes.values_mut() {
            states.clear();
        }
        f
    }

    /// Same flattening without touching `input` (the golden harness proves the very same `ProverInput` afterwards).
    pub fn snapshot(input: &ProverInput, order: MemoryOrder) -> Self {
        let ins = &input.instructions;
        let bundles = OPCODE_GROUPS
            .iter()
            .map(|group| {
                let mut v = Vec::new();
                for opcode in group.iter() {
                    if let Some(states) = ins.states_by_opcodes.get(opcode) {
                        v.extend(states.iter().map(bundle));
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
        let mut init: Vec<_> = input.memory.initial_memory.iter().collect();
        let mut fin: Vec<_> = input.memory.final_memory.iter().collect();
        if order == MemoryOrder::AscendingAddress {
            init.sort_by_key(|(a, _)| a.0);
            fin.sort_by_key(|(a, _)| a.0);
        }
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

    pub fn view(&self) -> cm_prover_input {
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

    /// The `input` object of a `tests/golden/ref_*.json` file: every array as rows of plain numbers, in the layout
    /// `cairo_m_amd.lib.prover_input_arrays` / `ArrayInput` use on the Python side (bundle = 12 words: pc, fp, clock,
    /// inst_prev_clock, inst[6], span_start, span_len; memory cell = address, value[4], clock, multiplicity; ...).
    pub fn to_json(&self) -> String {
        fn rows<T, const N: usize>(v: &[T], f: impl Fn(&T) -> [u32; N]) -> String {
            let parts: Vec<String> = v
                .iter()
                .map(|x| format!("[{}]", f(x).iter().map(|w| w.to_string()).collect::<Vec<_>>().join(",")))
                .collect();
            format!("[{}]", parts.join(","))
        }
        let mut out = String::from("{");
        out += &format!("\"regs\":[{},{},{},{}],", self.regs[0], self.regs[1], self.regs[2], self.regs[3]);
        out += &format!("\"roots\":[{},{}],", self.roots[0], self.roots[1]);
        out += &format!(
            "\"ranges\":[{},{},{},{},{},{}],",
            self.ranges[0][0], self.ranges[0][1], self.ranges[1][0], self.ranges[1][1], self.ranges[2][0], self.ranges[2][1]
        );
        for (k, b) in self.bundles.iter().enumerate() {
            out += &format!(
                "\"bundles{k}\":{},",
                rows(b, |x: &cm_bundle| [
                    x.pc, x.fp, x.clock, x.inst_prev_clock, x.inst[0], x.inst[1], x.inst[2], x.inst[3], x.inst[4], x.inst[5],
                    x.span_start, x.span_len
                ])
            );
        }
        out += &format!(
            "\"data_accesses\":{},",
            rows(&self.data_accesses, |a: &cm_data_access| [a.address, a.prev_clock, a.prev_value, a.value])
        );
        let cell = |c: &cm_memory_cell| [c.address, c.value[0], c.value[1], c.value[2], c.value[3], c.clock, c.multiplicity];
        out += &format!("\"initial_memory\":{},", rows(&self.initial_memory, cell));
        out += &format!("\"final_memory\":{},", rows(&self.final_memory, cell));
        out += &format!(
            "\"clock_updates\":{},",
            rows(&self.clock_updates, |c: &cm_clock_update| [c.address, c.prev_clock, c.value[0], c.value[1], c.value[2], c.value[3]])
        );
        let node = |n: &cm_merkle_node| [n.index, n.depth, n.left_value, n.right_value, n.parent_value, n.left_mult, n.right_mult, n.parent_mult];
        out += &format!("\"initial_tree\":{},", rows(&self.initial_tree, node));
        out += &format!("\"final_tree\":{}", rows(&self.final_tree, node));
        out += "}";
        out
    }
}
