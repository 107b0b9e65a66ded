#!/usr/bin/env python3
"""Case files for the reference-side golden harness (integration/prover-hip/tests/golden_dump.rs).

Each tests/golden/cases/<name>.case.json is ONE runner segment as this repository's synthetic VM emits it — the VM trace, the
memory access log, the memory at segment start and the public ranges (include/cairom_hip.h cm_runner_segment) — as plain
numbers.  The Rust harness feeds it to the reference's own `import_from_runner_output` and `prove_cairo_m`, and writes
tests/golden/ref_<name>.json (input as the reference's adapter built it, transcript, roots, proof), which
tests/test_ref_golden.py then compares with the oracle and the HIP prover step by step.

The programs are hand-assembled (no compiler in this image); what they are is stated in each file's `program_note`.

    python tools/make_ref_cases.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_case(name, note, seg):
    from cairo_m_amd.lib import runner_segment_arrays
    a = runner_segment_arrays(seg.view)
    doc = {"name": name, "program_note": note, "trace": a["trace"].tolist(), "memory_trace": a["memory_trace"].tolist(),
           "initial_memory": a["initial_memory"].tolist(), "ranges": a["ranges"]}
    path = os.path.join(ROOT, "tests", "golden", "cases", f"{name}.case.json")
    with open(path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print(f"{name}: {a['trace'].shape[0] - 1} steps, {a['memory_trace'].shape[0]} logged accesses, "
          f"{a['initial_memory'].shape[0]} initial cells -> {os.path.getsize(path)} bytes")


def main():
    from cairo_m_amd.lib import synth_fibonacci_segment, vm_segment
    from cairo_m_amd.workloads import all_opcodes_program
    from tests.ref_inputs import recursive_fib_program
    os.makedirs(os.path.join(ROOT, "tests", "golden", "cases"), exist_ok=True)
    s = synth_fibonacci_segment(5)
    write_case("fib_loop_5", "hand-assembled fibonacci_loop of SURVEY 8d (19 instructions), argument 5, 62 steps, one segment", s)
    s.free()
    s = vm_segment(recursive_fib_program(), entry_pc=0, args=(5,), n_returns=1)
    write_case("fib_rec_5", "hand-assembled recursive fibonacci (tests/ref_inputs.py::recursive_fib_program), argument 5: 14 calls", s)
    s.free()
    prog, steps = all_opcodes_program(3)
    s = vm_segment(prog, entry_pc=0, args=(), n_returns=0)
    write_case("all_opcodes_3", f"cairo_m_amd/workloads.py::all_opcodes_program(3): every opcode component live, {steps} steps "
               "(the two u32_store_eq components excepted: their reference AIR cannot balance, DESIGN.md §2 QUIRK)", s)
    s.free()


if __name__ == "__main__":
    main()
