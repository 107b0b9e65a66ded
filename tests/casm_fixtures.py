"""tests/golden/casm/*.json: 85 programs the reference's compiler emitted (CASM listings of
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
    inp = vm_run(fx["instructions"], entry_pc=fx["entry_pc"], args=case["args"], n_returns=fx["n_returns"], lib=lib)
    a = prover_input_arrays(inp.view)
    fp = a["regs"][1]
    fin = {int(r[0]): int(r[1]) for r in a["final_memory"]}
    nr = fx["n_returns"]
    return inp, [fin.get(fp - 2 - nr + i) for i in range(nr)]
