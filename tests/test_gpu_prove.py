"""GPU parity for the whole hot path: cm_prove_segment (HIP) vs the CPU oracle on the same ProverInput —
the flat proof words (commitments, sampled values, FRI layers, decommitments, PoW nonces) must be
bit-identical, and the oracle verifier must accept the HIP proof."""
import json

import numpy as np
import pytest

from cairo_m_amd.lib import synth_fibonacci

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [3, 100])
def test_fibonacci_proof_bit_exact(backend, oracle, n):
    inp = synth_fibonacci(n)
    proof = backend.prove(inp)
    got = proof.words()
    want, cells = oracle.prove(inp.view)
    assert proof.stats()["cells"] == cells
    assert got.size == want.size, (got.size, want.size)
    diff = np.nonzero(got != want)[0]
    assert diff.size == 0, f"first differing word {diff[:5]}"
    rc, err = oracle.verify(got)
    assert rc == 0, err
    # tampering is rejected
    bad = got.copy()
    bad[bad.size // 2] ^= 1
    assert oracle.verify(bad)[0] != 0
    # JSON has the serde shape of Proof<H>
    j = json.loads(proof.json())
    assert list(j.keys()) == ["claim", "interaction_claim", "public_data", "stark_proof", "interaction_pow"]
    assert len(j["stark_proof"]["commitments"]) == 4 and len(j["claim"]["opcodes"]) == 26
    proof.free()
    inp.free()
