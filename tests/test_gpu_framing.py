"""GPU half of the framing switches + the transcript log: under EVERY setting of the named switches (default, each
alternate alone, all alternates) the HIP proof equals the oracle's proof word for word and the two transcripts agree step
by step (same Channel call, same digest after it, same words mixed in / drawn) — so whichever reading a reference-produced
golden (tests/test_ref_golden.py) turns out to need, product and oracle already agree on it."""
import numpy as np
import pytest

from cairo_m_amd.lib import get_framing, set_framing, set_transcript_log, synth_fibonacci
from tests.ref_inputs import unchanged_memory_input

pytestmark = pytest.mark.gpu

SETTINGS = ["", "mix_u64=u32s", "hash_node=rfc", "sample_batch=sorted", "pcs_mix=blq",
            "mix_u64=u32s,hash_node=rfc,sample_batch=sorted,pcs_mix=blq"]


def _same_transcript(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for i, (x, y) in enumerate(zip(a, b)):
        assert x == y, f"transcript step {i}: product {x} != oracle {y}"


@pytest.mark.parametrize("spec", SETTINGS)
def test_hip_equals_oracle_under_every_framing(backend, oracle, spec):
    inp = synth_fibonacci(100)
    try:
        set_framing(spec, backend.L)
        oracle.set_framing(spec)
        set_transcript_log(True, backend.L)
        p = backend.prove(inp)
        got = p.words()
        want, _, tr_o = oracle.prove(inp.view, transcript=True)
        assert got.size == want.size and np.array_equal(got, want), get_framing(backend.L)
        _same_transcript(p.transcript(), tr_o)
        assert p.verify()[0] == 0 and oracle.verify(got)[0] == 0
        p.free()
    finally:
        set_transcript_log(False, backend.L)
        set_framing("", backend.L)
        oracle.set_framing("")
        inp.free()


def test_a_proof_made_under_another_framing_is_refused(backend, oracle):
    inp = synth_fibonacci(3)
    try:
        set_framing("hash_node=rfc", backend.L)
        p = backend.prove(inp)
        w = p.words()
        assert p.verify()[0] == 0
        set_framing("", backend.L)
        assert p.verify()[0] != 0 and oracle.verify(w)[0] != 0
        p.free()
    finally:
        set_framing("", backend.L)
        inp.free()


def test_transcript_shape_follows_prove_cairo_m(backend, oracle):
    """The logged calls in the order of prover.rs:36-131 on the reference's hand-built input (tests/prover.rs:33-112): 4 x
    mix_u64 (PcsConfig), public data (mix_u32s), root 0, 34 claim log sizes, root 1, the interaction nonce, 8 relation
    draws, 34 claimed sums, root 2, random coefficient, root 3, OODS point, sampled values, ..."""
    inp = unchanged_memory_input()
    try:
        set_transcript_log(True, backend.L)
        p = backend.prove(inp)
        tr = p.transcript()
        want, _, tr_o = oracle.prove(inp.view, transcript=True)
        assert np.array_equal(p.words(), want)
        _same_transcript(tr, tr_o)
        ops = [e["op"] for e in tr]
        assert ops[:4] == ["mix_u64"] * 4
        i = 4
        while ops[i] == "mix_u32s":
            i += 1
        assert i >= 6 and ops[i] == "mix_root"                         # public data, then the preprocessed root
        assert ops[i + 1:i + 35] == ["mix_u64"] * 34 and ops[i + 35] == "mix_root"      # claim, trace root
        assert ops[i + 36] == "mix_u64"                                # interaction proof-of-work nonce
        assert ops[i + 37:i + 45] == ["draw_felts"] * 8                # Relations::draw
        assert ops[i + 45:i + 79] == ["mix_felts"] * 34 and ops[i + 79] == "mix_root"   # interaction claim, interaction root
        assert ops[i + 80:i + 83] == ["draw_felt", "mix_root", "draw_felt"]              # random coeff, composition root, OODS
        assert ops[i + 83] == "mix_felts" and ops[i + 84] == "draw_felt"                 # sampled values, quotient coefficient
        last_mix = max(k for k, o in enumerate(ops) if o == "mix_u64")                   # the proof-of-work nonce ...
        assert ops[last_mix - 1] == "mix_felts"                                          # ... behind the last FRI layer
        assert last_mix < len(ops) - 1 and set(ops[last_mix + 1:]) == {"draw_random_bytes"}   # then only the query draws
        p.free()
    finally:
        set_transcript_log(False, backend.L)
        inp.free()


def test_framing_cannot_change_under_a_running_proof(backend):
    """One proof = one framing: the trees, the transcript and the quotient planning of a proof read the process-wide setting at
    their own points, so cm_set_framing is refused (status 1, setting unchanged) while a prover is alive and accepted again
    afterwards.  While six proofs run on the library's threads this thread keeps asking for the ALTERNATE node hashing: every
    request that arrives while a prover is alive must be refused; one that slips in between two proofs is undone at once (and the
    refusals of the undo count as well); afterwards the default is in force and a change is accepted."""
    import threading
    inp = synth_fibonacci(100_000)
    dev = backend.upload_input(inp)
    proofs, errors = [], []

    def work():
        try:
            proofs.extend(backend.prove_many([dev] * 6, inflight=2))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    t = threading.Thread(target=work)
    refused = accepted = 0
    try:
        backend.prove_device(dev).free()          # warm pools: the part below is steady-state proving (~6 ms per proof)
        t.start()
        while t.is_alive():
            if backend.L.cm_set_framing(b"hash_node=rfc") != 0:
                refused += 1
                assert "hash_node=raw" in get_framing(backend.L)      # a refused request changes nothing
                continue
            accepted += 1                           # slipped in while no prover was alive: put the default back
            while backend.L.cm_set_framing(b"") != 0:
                refused += 1                        # (a proof started under the alternate framing and holds it until it is done)
        t.join()
        assert not errors, errors
        assert refused > 0, (refused, accepted)
        assert backend.L.cm_set_framing(b"") == 0 and "hash_node=raw" in get_framing(backend.L)
        assert backend.L.cm_set_framing(b"hash_node=rfc") == 0 and "hash_node=rfc" in get_framing(backend.L)   # idle: accepted
        assert len(proofs) == 6
    finally:
        if t.is_alive():
            t.join()
        for p in proofs:
            p.free()
        set_framing("", backend.L)
        backend.free_input(dev)
        inp.free()
