"""tests/golden/casm/*.json: 122 programs the reference's compiler emitted (CASM listings of
crates/compiler/codegen/tests/snapshots/*.snap) with the values their entry functions return, computed from the snapshots' source
texts by tools/casm/cm_eval.py — made by tools/casm/make_casm_fixtures.py in the build container (data only)."""
import glob
import json
import os

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "casm")


def load():
    return [json.load(open(p)) for p in sorted(glob.glob(os.path.join(DIR, "*.json")))]


def run_case(fx, case, lib=None):
    """-> (HostInput, returned words): the program on the library's VM (cm_vm_run) with the runner's calling convention
    (crates/runner/src/lib.rs:384-467: arguments, return slots, old fp, return pc below the frame pointer)"""
    from cairo_m_amd.lib import prover_input_arrays, vm_run
    # `data`: the cells the listing places behind the instructions (constant arrays, the heap cursor) — four words each, loaded like
    # instructions (crates/runner/src/vm/mod.rs:75-101: instructions and values are one linear image, program_length covers both)
    inp = vm_run(fx["instructions"] + fx.get("data", []), entry_pc=fx["entry_pc"], args=case["args"], n_returns=fx["n_returns"], lib=lib)
    a = prover_input_arrays(inp.view)
    fp = a["regs"][1]
    fin = {int(r[0]): int(r[1]) for r in a["final_memory"]}
    nr = fx["n_returns"]
    return inp, [fin.get(fp - 2 - nr + i) for i in range(nr)]


def heap_program():
    """-> (program cells, entry_pc, n_returns, expected words): the compiler's `new felt[3]` listing (allocate_felt_and_read_back: the
    allocator reads the heap cursor behind the instructions, hands out cells counted down from MAX_ADDRESS = 2^28 - 1, the body
    writes and reads them through double dereferences) with ONE change that makes it provable: instruction 4, which the compiler
    emitted as `[fp + 12] = [fp + 12] + (-1)` — a cell read and written in one step, which the reference's AIR cannot prove (see
    make_casm_fixtures.py) — writes its result to the unused frame cell 100 instead, and instruction 6, its only reader, reads 100."""
    fx = next(f for f in load() if f["name"] == "pointers_and_heap_allocation__new____allocate_felt_and_read_back")
    ins = [list(w) for w in fx["instructions"]]
    assert ins[4] == [4, 12, 2**31 - 2, 12] and ins[6] == [1, 13, 12, 14]
    ins[4][3] = 100
    ins[6][2] = 100
    return ins + fx["data"], fx["entry_pc"], fx["n_returns"], fx["cases"][0]["expected"]
