"""Programs the reference's COMPILER emitted (not the builder): all 122 CASM listings the reference's codegen snapshots hold (the other two
of the 124 snapshots are expected compile errors) — loops, recursion, mutual recursion, call chains, felt arithmetic
incl. field division, every u32 operation, bitwise operations, comparisons, casts, assertions, structs, tuples, stack arrays with
constant and variable indices (StoreFramePointer + the four double-dereference opcodes), constant arrays in the read-only data
behind the instructions, heap allocations at the top of the 2^28-cell address space (tests/golden/casm/,
tools/casm/make_casm_fixtures.py).
CPU: the library's VM + adapter run them with the runner's calling convention and return what the SOURCE says they return, and the
resulting ProverInput satisfies every AIR constraint with cancelling LogUp sums (the reference's own check,
crates/prover/tests/prover.rs:351-370 `assert_constraints`)."""
import pytest

from tests.casm_fixtures import heap_program, load, run_case

FIXTURES = load()


def test_the_fixture_set_is_what_the_verdict_asked_for():
    names = {f["name"] for f in FIXTURES}
    assert len(FIXTURES) == 122                                                    # every listing the snapshots hold
    for must in ("loops_in_cairo_m___while_loop", "loops_in_cairo_m___for_loop", "loops_in_cairo_m___nested_loops",
                 "recursion_in_cairo_m___fibonacci_sequence", "bitwise_operations___bitwise_and", "bitwise_operations___bitwise_or",
                 "bitwise_operations___bitwise_xor", "multiple_functions_in_cairo_m___mutual_recursion"):
        assert must in names, must
    u32_ops = set(range(15, 42))
    assert sum(1 for f in FIXTURES if u32_ops & set(f["opcodes"])) >= 20          # programs with u32 opcodes
    assert all("fn " not in str(v) for f in FIXTURES for v in f.values())          # data only: no source text travels
    # every opcode the compiler used across the set
    assert {0, 1, 2, 3, 4, 6, 8, 9, 10, 11, 13, 14, 15, 16, 17, 18, 19, 21, 22, 23, 24, 28, 30, 34, 36, 37, 38, 42, 43, 44, 45, 48,
            50} <= set().union(*(f["opcodes"] for f in FIXTURES))
    assert sum(1 for f in FIXTURES if f["provable"]) >= 110
    assert sum(1 for f in FIXTURES if f.get("data")) >= 6                          # rodata / heap-cursor cells behind the instructions


@pytest.mark.parametrize("fx", FIXTURES, ids=[f["name"] for f in FIXTURES])
def test_vm_returns_what_the_source_says(fx, oracle):
    for k, case in enumerate(fx["cases"]):
        inp, got = run_case(fx, case)
        try:
            assert got == case["expected"], (fx["name"], case["args"], got, case["expected"])
            if k == 0:
                rc, err = oracle.assert_constraints(inp.view)
                if fx["provable"]:
                    assert rc == 0, (fx["name"], err)
                else:
                    # what the reference's own AIR cannot prove must be REJECTED, not proved: a same-step second access of a cell
                    # (prev_clock == clock, adapter/memory.rs:470-535) leaves range_check_20; U32StoreEq* leaves an unbalanced sum
                    assert rc != 0, (fx["name"], fx["unprovable_reason"])
                    if "range_check_20" in fx["unprovable_reason"]:
                        assert "rc20" in err, err
        finally:
            inp.free()


def test_heap_cells_at_the_top_of_the_address_space(oracle):
    """The compiler's heap allocation with its one in-place step moved to a fresh cell (tests/casm_fixtures.py heap_program):
    cells at 2^28 - 1, 2^28 - 2, 2^28 - 3 go through the memory component and the partial Merkle tree next to the program's own
    small addresses, and every constraint holds.  Cut into segments of 8 steps, the later segments START with a heap
    (crates/runner/src/vm/mod.rs:205-221: `heap[i]` maps to MAX_ADDRESS - i): those cells are in the initial memory of the segment
    whether it touches them or not."""
    from cairo_m_amd.lib import prover_input_arrays, vm_run
    prog, entry, nret, expected = heap_program()
    inp = vm_run(prog, entry_pc=entry, n_returns=nret)
    a = prover_input_arrays(inp.view)
    fin = {int(r[0]): int(r[1]) for r in a["final_memory"]}
    top = 2**28 - 1
    assert [fin[top - 2 + i] for i in range(3)] == [7, 8, 9]           # p[0], p[1], p[2]: base = MAX_ADDRESS - (cursor + size - 1)
    assert [fin[a["regs"][1] - 2 - nret + i] for i in range(nret)] == expected
    rc, err = oracle.assert_constraints(inp.view)
    assert rc == 0, err
    n_seg = inp.n_segments
    inp.free()
    assert n_seg == 1
    seen_untouched_heap_cell = False
    for s in range(4):
        inp = vm_run(prog, entry_pc=entry, n_returns=nret, max_steps=8, segment=s)
        assert inp.n_segments == 4
        a = prover_input_arrays(inp.view)
        init = {int(r[0]): r for r in a["initial_memory"]}
        heap_cells = sorted(k for k in init if k > 2**27)
        if s >= 2:
            assert heap_cells == [top - 2, top - 1, top], (s, heap_cells)   # the heap vector grew to 3 cells when p[0] was written
            seen_untouched_heap_cell |= any(int(init[k][6]) == 0 for k in heap_cells)
        rc, err = oracle.assert_constraints(inp.view)
        assert rc == 0, (s, err)
        inp.free()
    assert seen_untouched_heap_cell


def test_values_the_reference_states_itself():
    """Two mdtest programs carry their result in the reference's own markdown (`//! expected: V`, mdtest/README.md:56:
    01-basics/08-type-casts.md:46, 05-edge-cases/01-error-handling.md:25) — the only expected values in the tree that neither the
    fixture script's evaluator nor this repository computed.  The VM must return them."""
    stated = [f for f in FIXTURES if f.get("reference_expected") is not None]
    assert len(stated) >= 2
    # 33 more have their expected values pinned by the Rust equivalent the reference's authors wrote next to the program (the
    # fixture script evaluates it with tools/casm/rust_eval.py and stops on a disagreement with the source evaluator)
    assert sum(1 for f in FIXTURES if (f.get("rust_equivalent_cases") or 0) >= 1) >= 30
    for fx in stated:
        inp, got = run_case(fx, fx["cases"][0])
        inp.free()
        assert got == [fx["reference_expected"]], (fx["name"], got)
