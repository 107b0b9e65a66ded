"""Product-side verifier (cm_verify_proof_words = verify_cairo_m, crates/prover/src/verifier.rs:17-95): host code, so
it is tested on CPU against proofs made by the CPU oracle prover; the GPU suite checks it on HIP proofs.
Two independently written verifiers (oracle/overifier.hpp and cairo_m_amd/csrc/verifier.hip) must agree."""
import ctypes as C

import numpy as np
import pytest

from cairo_m_amd.lib import load_library, synth_fibonacci, vm_run


def _verify(L, words, cfg=None):
    w = np.ascontiguousarray(words, dtype=np.uint32)
    rc = L.cm_verify_proof_words(w.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint64(w.size),
                                 (C.c_uint32 * 4)(*cfg) if cfg else None)
    buf = C.create_string_buffer(512)
    L.cm_last_error(buf, C.c_size_t(512))
    return rc, buf.value.decode(errors="replace")


def test_product_verifier_accepts_oracle_proof_and_rejects_tampering(oracle):
    L = load_library()
    inp = synth_fibonacci(7)
    words, _ = oracle.prove(inp.view)
    assert oracle.verify(words)[0] == 0
    rc, err = _verify(L, words)
    assert rc == 0, err
    rng = np.random.default_rng(1)
    rejected = 0
    for pos in rng.integers(8, words.size - 1, size=40):
        bad = words.copy()
        bad[pos] ^= 1
        rc_p, _ = _verify(L, bad)
        rc_o, _ = oracle.verify(bad)
        assert (rc_p == 0) == (rc_o == 0), f"verifiers disagree on a flip at word {pos}"
        rejected += rc_p != 0
    assert rejected >= 38          # (a flip inside an unused `present = 0` public entry can be inert)
    assert _verify(L, words[:-3])[0] != 0      # truncated stream
    inp.free()


def test_product_verifier_on_u32_program(oracle):
    from tests.test_oracle_air import u32_program
    L = load_library()
    inp = vm_run(u32_program(), entry_pc=0, args=(), n_returns=0)
    words, _ = oracle.prove(inp.view)
    rc, err = _verify(L, words)
    assert rc == 0, err
    inp.free()


@pytest.mark.parametrize("bound", [1, 2])
def test_nonzero_last_layer_degree_bound(oracle, bound):
    """PcsConfig with log_last_layer_degree_bound > 0 (prove_cairo_m takes any Option<PcsConfig>, prover.rs:23-29): FRI
    stops 2^(bound+1) points early, the last layer is interpolated (Stwo LineEvaluation::interpolate) and 2^bound
    coefficients go into the proof in LinePoly's bit-reversed order.  Both verifiers accept, and evaluate that polynomial
    at the folded queries, so any flip in it is rejected."""
    L = load_library()
    inp = synth_fibonacci(7)
    cfg = (5, 1, bound, 20)
    words, _ = oracle.prove(inp.view, cfg=cfg)
    assert oracle.verify(words, cfg)[0] == 0
    rc, err = _verify(L, words, cfg)
    assert rc == 0, err
    # a verifier expecting REGULAR_96_BITS refuses a proof made under this (weaker) config
    assert _verify(L, words)[0] != 0 and oracle.verify(words)[0] != 0
    # locate the last-layer polynomial: its (count, 4*count words, log_size) record is unique in the stream
    n = 1 << bound
    hits = [i for i in range(words.size - 4 * n - 1) if words[i] == n and words[i + 1 + 4 * n] == bound
            and all(w < 2**31 - 1 for w in words[i + 1:i + 1 + 4 * n]) and np.count_nonzero(words[i + 1:i + 1 + 4 * n]) >= 3 * n]
    assert hits, "last-layer record not found"
    for k in range(4 * n):
        bad = words.copy()
        bad[hits[-1] + 1 + k] ^= 1
        assert _verify(L, bad, cfg)[0] != 0 and oracle.verify(bad, cfg)[0] != 0
    inp.free()


def test_verifier_does_not_take_the_security_level_from_the_proof(oracle):
    """verify_cairo_m takes pcs_config from the CALLER (verifier.rs:17-31).  A proof made with (pow_bits 0, 1 query) is a
    valid proof under that config and must be rejected by a verifier expecting REGULAR_96_BITS — and rewriting the config
    words of the stream to claim REGULAR_96_BITS does not help (the transcript and the query count no longer match)."""
    L = load_library()
    inp = synth_fibonacci(5)
    weak = (0, 1, 0, 1)
    words, _ = oracle.prove(inp.view, cfg=weak)
    assert _verify(L, words, weak)[0] == 0 and oracle.verify(words, weak)[0] == 0
    rc, err = _verify(L, words)
    assert rc != 0 and "config" in err
    assert oracle.verify(words)[0] != 0
    forged = words.copy()
    assert list(forged[1:5]) == [0, 1, 0, 1]
    forged[1:5] = [16, 1, 0, 80]
    assert _verify(L, forged)[0] != 0 and oracle.verify(forged)[0] != 0
    inp.free()


def test_public_data_words_must_be_canonical(oracle):
    """proof_from_words rejects public-data words >= P (M31(x) needs x < P: P would alias 0, 2^31 would alias 1)."""
    L = load_library()
    inp = synth_fibonacci(5)
    words, _ = oracle.prove(inp.view)
    assert _verify(L, words)[0] == 0
    n_comp = int(words[5])
    base = 6 + n_comp + 4 * n_comp           # magic, cfg[4], count, log sizes, claimed sums -> public data scalars
    for k in range(7):
        bad = words.copy()
        bad[base + k] = bad[base + k] + (2**31 - 1) if bad[base + k] < 2**31 - 1 else bad[base + k]
        rc, err = _verify(L, bad)
        assert rc != 0 and "malformed" in err, (k, err)
    inp.free()


@pytest.mark.parametrize("cfg", [(5, 2, 1, 12)])
def test_log_blowup_factor_above_one(oracle, cfg):
    """PcsConfig with log_blowup_factor 2 (3 is covered on the GPU, tests/test_gpu_prove.py): the committed LDE domain differs from the constraint-evaluation domain.
    Both verifiers accept the oracle's proof under that config and reject flips."""
    L = load_library()
    inp = synth_fibonacci(6)
    words, _ = oracle.prove(inp.view, cfg=cfg)
    assert oracle.verify(words, cfg)[0] == 0
    rc, err = _verify(L, words, cfg)
    assert rc == 0, err
    rng = np.random.default_rng(2)
    for pos in rng.integers(8, words.size - 1, size=12):
        bad = words.copy()
        bad[pos] ^= 1
        assert (_verify(L, bad, cfg)[0] == 0) == (oracle.verify(bad, cfg)[0] == 0)
    inp.free()


def test_a_column_with_more_sampled_values_than_its_mask_is_rejected(oracle):
    """The proof object keeps a column's sampled values in an inline pair (proof.hpp SampleVec: every AIR here samples one or two
    points per column) that spills into a vector for anything longer: a foreign word stream with THREE values in a column must
    parse (the spill path) and be refused as a structure error — by both verifiers."""
    L = load_library()
    inp = synth_fibonacci(7)
    words, _ = oracle.prove(inp.view)
    inp.free()
    w = [int(x) for x in words]
    i = 5                                  # magic, four config words
    nc = w[i]; i += 1 + nc + 4 * nc        # claim: log sizes, claimed sums
    i += 7                                 # public registers / clock / roots
    for _ in range(3):                     # program, input, output entries (7 words each)
        c = w[i]; i += 1 + 7 * c
    i += 2                                 # interaction proof of work
    nt = w[i]; i += 1 + 8 * nt             # commitments
    assert nt == 4
    ncol = w[i]; i += 1                    # tree 0: columns
    assert 1 <= ncol < 64
    ns = w[i]
    assert ns in (1, 2)                    # the first preprocessed column's samples
    bad = w[:i] + [3] + w[i + 1:i + 1 + 4 * ns] + [0, 0, 0, 0] * (3 - ns) + w[i + 1 + 4 * ns:]
    rc, err = _verify(L, np.array(bad, dtype=np.uint32))
    assert rc != 0 and "InvalidStructure" in err, (rc, err)
    assert oracle.verify(np.array(bad, dtype=np.uint32))[0] != 0
