"""GPU: compiler-emitted programs (tests/golden/casm/, see tests/test_casm_programs.py) through the whole HIP prover — the proof
equals the CPU oracle's word for word, and both verifiers accept it.  North star: "bit-identical to the reference CPU backend on
the same compiled program and arguments"."""
import numpy as np
import pytest

from tests.casm_fixtures import load, run_case

pytestmark = pytest.mark.gpu
FIXTURES = [f for f in load() if f["provable"]]


@pytest.mark.parametrize("fx", FIXTURES, ids=[f["name"] for f in FIXTURES])
def test_compiler_emitted_program_proof_bit_exact(fx, backend, oracle):
    case = fx["cases"][-1]
    inp, got = run_case(fx, case, backend.L)
    try:
        assert got == case["expected"]
        proof = backend.prove(inp)
        words = proof.words()
        want, _ = oracle.prove(inp.view)
        assert words.size == want.size and np.array_equal(words, want), fx["name"]
        assert proof.verify()[0] == 0
        assert oracle.verify(words)[0] == 0
        proof.free()
    finally:
        inp.free()
