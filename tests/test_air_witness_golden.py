"""The `witness` half of the opcode AIR descriptions (cairo_m_amd/csrc/air/opcodes_*.hpp) against golden trace cells
derived MECHANICALLY from the reference's `Claim::write_trace` row closures: tools/rsref/rs_witness.py parses each
closure with a small Rust-subset interpreter (tools/rsref/rs_interp.py) and executes it on the packed bundles of the
all-opcode program below — no hand transcription.  Every cell of every column must match, live rows and padding rows
(ExecutionBundle::default() lanes) alike, for all 26 opcode components (store_fp_fp and store_fp_imm
derive per-lane hints in a pre-pack closure, which is interpreted too; tests/test_air_hot_independent.py additionally
holds a hand-written numpy model of the six hot components).  The HIP witness kernels
instantiate the same descriptions and are compared with the oracle cell by cell in tests/test_gpu_components.py (and on
this very program in tests/test_gpu_workloads.py)."""
import os

import numpy as np
import pytest

from cairo_m_amd.lib import prover_input_arrays, vm_run
from cairo_m_amd.workloads import all_opcodes_program

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "air_witness_vectors.npz"))
OPCODE_FILES = [
    "assert_eq_fp_imm", "call_abs_imm", "jmp_imm", "jnz_fp_imm", "ret", "store_imm", "store_fp_fp", "store_fp_imm",
    "double_deref_fp_imm", "double_deref_fp_fp", "store_frame_pointer", "u32_store_imm", "u32_store_add_fp_imm",
    "u32_store_mul_fp_imm", "u32_store_div_fp_imm", "u32_store_eq_fp_fp", "u32_store_eq_fp_imm", "u32_store_lt_fp_imm",
    "u32_store_lt_fp_fp", "u32_store_add_fp_fp", "u32_store_sub_fp_fp", "u32_store_mul_fp_fp", "u32_store_div_fp_fp",
    "u32_store_bitwise_fp_fp", "u32_store_bitwise_fp_imm", "store_le_fp_imm"]      # air::ComponentId order


@pytest.fixture(scope="module")
def run():
    prog, steps = all_opcodes_program(int(GOLD["iters"][0]), int(GOLD["seed"][0]))
    inp = vm_run(prog, entry_pc=0, args=(), n_returns=0)
    assert inp.steps == steps == int(GOLD["steps"][0])
    yield inp
    inp.free()


def test_program_is_valid(oracle, run):
    """every constraint vanishes on every row and the LogUp sums cancel (tests/prover.rs:351-370)"""
    rc, err = oracle.assert_constraints(run.view)
    assert rc == 0, err
    n = [prover_input_arrays(run.view)[f"bundles{c}"].shape[0] for c in range(26)]
    assert all(x > 0 for c, x in enumerate(n) if c not in (15, 16)) and n[15] == n[16] == 0


@pytest.mark.parametrize("cid", [c for c in range(26) if OPCODE_FILES[c] in GOLD.files])
def test_witness_matches_reference_derived_cells(oracle, run, cid):
    want = GOLD[OPCODE_FILES[cid]]
    got = oracle.component_trace(run.view, cid)
    assert got.shape == want.shape, (OPCODE_FILES[cid], got.shape, want.shape)
    bad = np.argwhere(got != want)
    assert bad.size == 0, f"{OPCODE_FILES[cid]}: first differing (column, row) {bad[:5].tolist()}"


@pytest.mark.parametrize("name,cid", [("memory", 26), ("merkle", 27)])
def test_builtin_witness_matches_reference_derived_cells(oracle, run, name, cid):
    """memory.rs:157-195 / merkle.rs:154-201 closures interpreted on the boundary-memory rows / partial-tree nodes of the run."""
    want = GOLD[name]
    got = oracle.component_trace(run.view, cid)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_clock_update_witness_matches_reference_derived_cells(oracle, run):
    """clock_update.rs:123-152 on synthetic (address, prev_clock, value) entries (the small program has no clock-update row)."""
    import ctypes as C
    from cairo_m_amd.lib import ProverInputView
    cu = np.ascontiguousarray(GOLD["clock_update_input"], dtype=np.uint32)
    v = ProverInputView.from_buffer_copy(C.cast(run.view, C.POINTER(ProverInputView)).contents)
    v.clock_updates = cu.ctypes.data
    v.n_clock_updates = cu.shape[0]
    got = oracle.component_trace(C.cast(C.pointer(v), C.c_void_p), 28)
    want = GOLD["clock_update"]
    assert got.shape == want.shape and np.array_equal(got, want)


def test_poseidon2_witness_matches_reference_derived_cells(oracle, run):
    """poseidon2.rs:210-319 (round loops + the file's helper functions, interpreted) on the first 200 hash inputs of the
    run's partial Merkle trees: live rows equal the oracle's poseidon2 trace, and an all-padding packed row equals an
    all-padding row of the oracle's (443 columns each).  Round constants: KAT-pinned (tools/rsref/rs_poseidon2.py)."""
    want = GOLD["poseidon2"]
    got = oracle.component_trace(run.view, 29)
    a = prover_input_arrays(run.view)
    nodes = np.concatenate([a["initial_tree"], a["final_tree"]])
    assert np.array_equal(GOLD["poseidon2_inputs"][:, :2], nodes[:200, 2:4])
    assert got.shape[0] == want.shape[0] == 443 and got.shape[1] >= 2048
    assert np.array_equal(got[:, :192], want[:, :192])                       # 12 fully live packed rows
    assert np.array_equal(got[1:, nodes.shape[0] + 16:nodes.shape[0] + 32], want[1:, 224:240])   # padding rows (enabler column aside)
    assert not got[0, nodes.shape[0]:].any() and not want[0, 200:].any()


def test_coverage():
    assert set(OPCODE_FILES) <= set(GOLD.files)          # all 26 opcode components
