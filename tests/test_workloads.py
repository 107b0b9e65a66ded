"""Workload generators (cairo_m_amd/workloads.py) on the CPU: the synthetic VM's SHA-256 digest equals hashlib's (a
functional check of the u32 opcode semantics that owes nothing to the AIR text), and both programs satisfy every AIR
constraint with cancelling LogUp sums (the reference's assert_constraints check, tests/prover.rs:351-370)."""
import hashlib

import pytest

from cairo_m_amd.lib import prover_input_arrays, vm_run
from cairo_m_amd.workloads import all_opcodes_program, sha256_pad, sha256_program


@pytest.mark.parametrize("msg", [b"", b"abc", b"The quick brown fox jumps over the lazy dog" * 2])
def test_sha256_on_the_vm_matches_hashlib(oracle, msg):
    prog, slots = sha256_program(msg)
    inp = vm_run(prog, entry_pc=0, args=(), n_returns=0)
    a = prover_input_arrays(inp.view)
    fp = a["regs"][1]
    fin = {int(r[0]): int(r[1]) for r in a["final_memory"]}
    dig = b"".join(((fin[fp + s + 1] << 16) | fin[fp + s]).to_bytes(4, "big") for s in slots)
    assert dig == hashlib.sha256(msg).digest()
    assert len(sha256_pad(msg)) % 16 == 0
    if len(msg) <= 3:
        rc, err = oracle.assert_constraints(inp.view)
        assert rc == 0, err
    inp.free()


def test_all_opcodes_program_steps_and_constraints(oracle):
    for iters in (1, 2, 9):
        prog, steps = all_opcodes_program(iters)
        inp = vm_run(prog, entry_pc=0, args=(), n_returns=0)
        assert inp.steps == steps
        if iters == 9:
            rc, err = oracle.assert_constraints(inp.view)
            assert rc == 0, err
            n = [prover_input_arrays(inp.view)[f"bundles{c}"].shape[0] for c in range(26)]
            assert sum(x > 0 for x in n) == 24          # every opcode component but the two (unprovable) u32 eq ones
        inp.free()
