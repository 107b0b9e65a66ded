"""The reference's own prover tests (crates/prover/tests/prover.rs) restated as HIP-vs-oracle parity tests: every proof is
bit-identical to the oracle's and accepted by both verifiers.

* prover.rs:33-112   test_prove_and_verify_unchanged_memory — hand-built ProverInput, no instruction at all
* prover.rs:175-200  recursive fibonacci — call_abs_imm / ret at depth
* prover.rs:203-243  test_hash_continuity_fibonacci — max_steps = 10, the final root of a segment is the next one's initial root
* prover.rs:372-446  test_fibonacci_public_memory_contents — program / input / output values read back from the proof
* benches/prover_speed_benchmark.rs:78-120 — the 1 KiB SHA-256, bit-exact
plus the two witness components that only whole proofs covered: clock_update (28) against the reference-derived golden
cells, poseidon2 (29) at FULL length against per-column digests of the reference-derived trace."""
import hashlib
import json
import os

import numpy as np
import pytest

from cairo_m_amd.lib import ArrayInput, prover_input_arrays, synth_fibonacci, vm_run
from tests.ref_inputs import P, recursive_fib_program, unchanged_memory_input

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "air_witness_vectors.npz"))


def _bit_exact(backend, oracle, inp):
    p = backend.prove(inp)
    got = p.words()
    want, _ = oracle.prove(inp.view)
    assert got.size == want.size and np.array_equal(got, want)
    assert p.verify()[0] == 0 and oracle.verify(got)[0] == 0
    return p


def test_unchanged_memory_no_instructions(backend, oracle):
    inp = unchanged_memory_input()
    p = _bit_exact(backend, oracle, inp)
    j = json.loads(p.json())
    pd = j["public_data"]
    assert pd["clock"] == 0 and pd["initial_root"] == pd["final_root"] != 0
    assert pd["initial_registers"] == pd["final_registers"] == {"pc": 0, "fp": 0}
    assert all(c["log_size"] == 4 for c in j["claim"]["opcodes"].values())      # 26 idle opcode components: 16 padding rows each
    p.free()


@pytest.mark.parametrize("n", [5, 12])
def test_recursive_fibonacci_deep_calls(backend, oracle, n):
    prog = recursive_fib_program()
    inp = vm_run(prog, entry_pc=0, args=(n,), n_returns=1)
    a = prover_input_arrays(inp.view)
    fib = [0, 1]
    for _ in range(n):
        fib.append(fib[-1] + fib[-2])
    n_calls = a["bundles1"].shape[0]
    assert n_calls == a["bundles4"].shape[0] - 1 and n_calls == 2 * fib[n + 1] - 2     # every call returns; + the top-level ret
    fin = {int(r[0]): int(r[1]) for r in a["final_memory"]}
    assert fin[a["regs"][1] - 3] == fib[n]                                             # return slot of the entry frame
    p = _bit_exact(backend, oracle, inp)
    p.free()
    inp.free()


def test_hash_continuity_with_max_steps_10(backend, oracle):
    roots, cells = [], []      # fibonacci_loop(5) = 10 * 5 + 12 = 62 steps, cut every 10 steps: 7 segments
    for seg in range(7):
        inp = synth_fibonacci(5, max_steps=10, segment=seg)
        assert inp.steps == (10 if seg < 6 else 2)
        a = prover_input_arrays(inp.view)
        cells.append((set(a["initial_memory"][:, 0].tolist()), set(a["final_memory"][:, 0].tolist())))
        p = _bit_exact(backend, oracle, inp)
        pd = json.loads(p.json())["public_data"]
        roots.append((pd["initial_root"], pd["final_root"]))
        p.free()
        inp.free()
    # The final root of a segment is the initial root of the next one whenever the next segment starts from the cells the
    # previous one ended with.  A cell that a segment WRITES first and that lies beyond the memory it was handed gets the
    # written value as its "initial" value (adapter/memory.rs:493-503, restated as is), so such a segment's initial tree has a
    # leaf the previous final tree did not: with this hand-assembled program and a 10-step cut that happens exactly once
    # (segment 1 first-writes fp+4 / fp+7); the reference's own test runs the COMPILED fib_loop, whose frame is written earlier.
    fresh = [k for k in range(6) if cells[k + 1][0] != cells[k][1]]
    assert fresh == [0] and cells[1][0] - cells[0][1] == {27, 30}
    for k in range(6):
        if k not in fresh:
            assert roots[k][1] == roots[k + 1][0], f"final root of segment {k} must be the initial root of segment {k + 1}"


def test_public_memory_contents(backend, oracle):
    """program / input / output entries of the proof's PublicData (public_data.rs:132-186) against the program and the run."""
    n = 7
    inp = synth_fibonacci(n)
    p = _bit_exact(backend, oracle, inp)
    pm = json.loads(p.json())["public_data"]["public_memory"]
    a = prover_input_arrays(inp.view)
    init = {int(r[0]): [int(x) for x in r[1:5]] for r in a["initial_memory"]}
    qm = lambda v: [v[0][0], v[0][1], v[1][0], v[1][1]]
    prog_entries = [e for e in pm["program"] if e is not None]
    assert len(prog_entries) == a["ranges"][1] - a["ranges"][0] == 19
    for e in prog_entries:
        addr, value, _clock = e
        assert qm(value) == init[addr]                       # the program words, instruction by instruction
    assert prog_entries[0][1][0][0] == 9                     # pc 0 is a StoreImm
    (ia, iv, _), = [e for e in pm["input"] if e is not None]
    assert qm(iv) == [n, 0, 0, 0] and ia == a["ranges"][2]
    (oa, ov, _), = [e for e in pm["output"] if e is not None]
    fib = [0, 1]
    for _ in range(n):
        fib.append(fib[-1] + fib[-2])
    assert qm(ov) == [fib[n], 0, 0, 0] and oa == a["ranges"][4]
    p.free()
    inp.free()


def test_sha256_1kib_bit_exact(backend, oracle):
    """benches/prover_speed_benchmark.rs:78-120: SHA-256 of a 1 KiB message (17 blocks)."""
    from cairo_m_amd.workloads import sha256_program
    from tests.test_gpu_workloads import _digest
    msg = bytes(range(256)) * 4
    prog, slots = sha256_program(msg)
    inp = vm_run(prog, entry_pc=0, args=(), n_returns=0)
    assert _digest(inp, slots) == hashlib.sha256(msg).digest()
    p = _bit_exact(backend, oracle, inp)
    p.free()
    inp.free()


def _run_all_opcodes():
    from cairo_m_amd.workloads import all_opcodes_program
    prog, steps = all_opcodes_program(int(GOLD["iters"][0]), int(GOLD["seed"][0]))
    inp = vm_run(prog, entry_pc=0, args=(), n_returns=0)
    assert inp.steps == steps
    return inp


def test_clock_update_witness_equals_reference_derived_cells(backend):
    """component 28 on the HIP kernel against cells derived from clock_update.rs:123-152 (no oracle in between)."""
    run = _run_all_opcodes()
    arrays = prover_input_arrays(run.view)
    arrays["clock_updates"] = GOLD["clock_update_input"]
    inp = ArrayInput(arrays)
    dev = backend.upload_input(inp)
    want = GOLD["clock_update"]
    log = backend.component_log_size(dev, 28)
    assert want.shape == (7, 1 << log)
    cols = [backend.col_alloc(1 << log) for _ in range(7)]
    backend.trace_write(dev, 28, cols)
    got = np.stack([backend.download(h, 1 << log) for h in cols])
    assert np.array_equal(got, want)
    for h in cols:
        backend.col_free(h)
    backend.free_input(dev)
    run.free()


def test_poseidon2_witness_full_length_equals_reference_derived_digests(backend):
    """component 29, every row (1092 live + padding = 2048 rows x 443 columns): one Blake2s-256 digest per column of the
    trace the reference's closure (poseidon2.rs:210-319, interpreted) produces — tools/rsref/rs_poseidon2.py."""
    run = _run_all_opcodes()
    dev = backend.upload_input(run)
    n_cols, n_rows, n_live = [int(x) for x in GOLD["poseidon2_full_shape"]]
    a = prover_input_arrays(run.view)
    assert a["initial_tree"].shape[0] + a["final_tree"].shape[0] == n_live
    log = backend.component_log_size(dev, 29)
    assert (1 << log) == n_rows and n_cols == 443
    cols = [backend.col_alloc(n_rows) for _ in range(n_cols)]
    backend.trace_write(dev, 29, cols)
    for c, h in enumerate(cols):
        col = backend.download(h, n_rows)
        assert hashlib.blake2s(col.tobytes()).digest() == GOLD["poseidon2_full_digests"][c].tobytes(), f"poseidon2 column {c}"
        backend.col_free(h)
    backend.free_input(dev)
    run.free()
