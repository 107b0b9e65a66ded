"""Pinning to the REFERENCE: replay of golden files written by the reference-side harness
(integration/prover-hip/tests/golden_dump.rs: reference `prove_cairo_m` + Stwo SimdBackend under a logging channel).

A golden file (tests/golden/ref_<case>.json[.gz]) carries the exact ProverInput the reference proved (memory rows in the order
its HashMaps iterated), the channel digest after every Fiat-Shamir call, the four roots, the interaction nonce and the proof
JSON.  For each file:
  * CPU  (-m "not gpu"): the ORACLE proves the same input -> transcript equal step by step, roots, nonce, proof JSON equal;
  * GPU  (-m gpu):       the HIP prover does, through the C ABI.
The first differing transcript step is reported together with the framing switch (include/cairom_hip.h cm_set_framing)
that governs it, so a mismatch is a switch flip away from green.

No reference-produced file can exist in the build image (no Rust toolchain, Stwo submodule empty): until a maintainer runs the
harness, `test_reference_goldens_present` is SKIPPED and says so — parity with Stwo stays "unpinned".  The machinery itself is
exercised by tests/golden/selfmade_*.json, which this repository's own oracle wrote (its "source" field says so; it pins nothing).
"""
import ctypes as C
import glob
import gzip
import json
import os

import numpy as np
import pytest

from cairo_m_amd.lib import ArrayInput, Proof, load_library, set_framing, set_transcript_log

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_FILES = sorted(glob.glob(os.path.join(GOLDEN, "ref_*.json")) + glob.glob(os.path.join(GOLDEN, "ref_*.json.gz")))
SELF_FILES = sorted(glob.glob(os.path.join(GOLDEN, "selfmade_*.json")))
# framing under which the goldens are replayed: "" = defaults; set CM_REF_FRAMING="hash_node=rfc,..." to try alternates
FRAMING = os.environ.get("CM_REF_FRAMING", "")


def load(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        return json.load(f)


def hint(step, entry):
    """which named switch governs a transcript step"""
    op = entry["op"]
    if step < 4:
        return "steps 0-3 are PcsConfig::mix_into: switch `mix_u64` (raw | u32s) if step 0 differs, `pcs_mix` (bql | blq) if step 2 or 3 does"
    if op == "mix_u64":
        return "switch `mix_u64` (raw | u32s)"
    if op == "mix_root":
        return ("a Merkle root: switch `hash_node` (raw | rfc) — or, for the FRI first-layer root (the mix_root after the "
                "sampled-values mix_felts + draw_felt), switch `sample_batch` (insertion | sorted)")
    if op in ("draw_felt", "draw_felts", "draw_random_bytes"):
        return "a draw only differs if an earlier mix did: look at the first differing step above this one"
    return "no switch covers this step: the restatement itself differs from the reference here"


def compare_transcripts(got, want, who):
    for i, (g, w) in enumerate(zip(got, want)):
        if g["op"] != w["op"] or g["digest"] != w["digest"] or g["n_words"] != w["n_words"] or g["words"] != w["words"]:
            raise AssertionError(f"{who}: transcript diverges from the golden at step {i}: got {g}, golden {w}.  {hint(i, w)}")
    assert len(got) == len(want), f"{who}: {len(got)} transcript steps, golden has {len(want)}"


def words_to_proof(words, L):
    h = C.c_void_p()
    w = np.ascontiguousarray(words, dtype=np.uint32)
    assert L.cm_proof_from_words(w.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint64(w.size), C.byref(h)) == 0
    return Proof(L, h)


def check_against_golden(g, transcript, proof, who):
    compare_transcripts(transcript, g["transcript"], who)
    assert [r.hex() for r in proof.commitments()] == g["commitments"], f"{who}: commitment roots differ"
    j = json.loads(proof.json())
    assert j["interaction_pow"] == g["interaction_pow"]
    assert j == g["proof"], f"{who}: proof JSON differs from the golden's (first differing top-level key: " \
                            f"{next((k for k in j if j[k] != g['proof'].get(k)), '?')})"


# ---- the framing sweep --------------------------------------------------------------------------------------------------------
# Four named switches (include/cairom_hip.h cm_set_framing), two readings each: 16 framings.  A golden is replayed under the
# default first; when that diverges, the other 15 are tried and the one(s) under which transcript, roots, nonce and proof all
# match are NAMED in the failure — the fix is then to make that reading the default on both sides (cairo_m_amd/csrc/framing.hpp,
# oracle/oframing.hpp), nothing else.  When no framing matches, the report carries, per framing, how far the transcript got.
SWITCHES = [("mix_u64", "raw", "u32s"), ("hash_node", "raw", "rfc"), ("sample_batch", "insertion", "sorted"), ("pcs_mix", "bql", "blq")]


def all_framings():
    out = []
    for mask in range(16):
        out.append(",".join(f"{n}={b if (mask >> i) & 1 else a}" for i, (n, a, b) in enumerate(SWITCHES)))
    return out   # out[0] = the defaults


def first_divergence(got, want):
    for i, (g, w) in enumerate(zip(got, want)):
        if g["op"] != w["op"] or g["digest"] != w["digest"] or g["n_words"] != w["n_words"] or g["words"] != w["words"]:
            return i
    return None if len(got) == len(want) else min(len(got), len(want))


def governing_switch(step, transcript):
    """the switch whose reading decides transcript step `step` (None: no switch covers it)"""
    if step >= len(transcript):
        return None
    if step < 2:
        return "mix_u64"
    if step < 4:
        return "pcs_mix"
    op = transcript[step]["op"]
    if op == "mix_u64":
        return "mix_u64"
    if op == "mix_root":
        first_root = next(i for i, e in enumerate(transcript) if e["op"] == "mix_root")
        # hash_node decides EVERY root, so a wrong reading shows at the first one; a later root that diverges behind matching
        # earlier roots is the FRI first-layer root: the order of the quotient sample batches
        return "hash_node" if step == first_root else "sample_batch"
    return None


def flip(spec, name):
    cur = dict(kv.split("=") for kv in spec.split(","))
    a, b = next((x, y) for n, x, y in SWITCHES if n == name)
    cur[name] = b if cur[name] == a else a
    return ",".join(f"{n}={cur[n]}" for n, _, _ in SWITCHES)


def replay_with_sweep(g, prove_under, who):
    """prove_under(spec) -> (transcript, Proof).  Passes silently when the golden replays under the framing in force (CM_REF_FRAMING
    or the defaults).  Otherwise searches the 16 framings and raises an AssertionError that NAMES the matching one.  The search is
    guided: every switch decides its own transcript steps (governing_switch), so the first diverging step says which switch to
    flip — at most four more proofs; CM_REF_FULL_SWEEP=1 tries all 16 instead (a proof of the smallest input takes ~10 s on the
    CPU oracle)."""
    tr, p = prove_under(FRAMING)
    try:
        check_against_golden(g, tr, p, who)
        return FRAMING
    except AssertionError as first:
        default_report = str(first)
    finally:
        p.free()
    matches, progress = [], []

    def attempt(spec):
        tr, p = prove_under(spec)
        try:
            check_against_golden(g, tr, p, who)
            matches.append(spec)
            return None
        except AssertionError:
            d = first_divergence(tr, g["transcript"])
            progress.append((len(g["transcript"]) if d is None else d, spec))
            return d
        finally:
            p.free()
    if os.environ.get("CM_REF_FULL_SWEEP"):
        for spec in all_framings():
            attempt(spec)
    else:
        spec = all_framings()[0] if not FRAMING else ",".join(
            f"{n}={dict(kv.split('=') for kv in FRAMING.split(',') if kv).get(n, a)}" for n, a, _ in SWITCHES)
        d = first_divergence(tr, g["transcript"])
        flipped = set()
        while d is not None:
            sw = governing_switch(d, g["transcript"])
            if sw is None or sw in flipped:
                break
            flipped.add(sw)
            spec = flip(spec, sw)
            d = attempt(spec)
    if matches:
        raise AssertionError(f"{who}: the golden does NOT replay under the framing in force ({FRAMING or 'defaults'}) but replays "
                             f"completely under: {' | '.join(matches)}.  Make that reading the default (cairo_m_amd/csrc/framing.hpp, "
                             f"oracle/oframing.hpp).  Default-framing report: {default_report[:400]}")
    progress.sort(reverse=True)
    raise AssertionError(f"{who}: no framing tried replays the golden; transcript steps matched before the first divergence, "
                         f"best first: {progress[:4]} (CM_REF_FULL_SWEEP=1 tries all 16).  Default-framing report: {default_report[:600]}")


def test_reference_goldens_present():
    if not REF_FILES:
        pytest.skip("no tests/golden/ref_*.json: nobody has run integration/prover-hip/tests/golden_dump.rs against the reference "
                    "yet (needs Rust + the stwo submodule) — the Stwo half of the oracle remains PARITY-UNPINNED")
    for f in REF_FILES:
        assert load(f)["source"] == "reference"


@pytest.mark.parametrize("path", REF_FILES + SELF_FILES, ids=[os.path.basename(p) for p in REF_FILES + SELF_FILES])
def test_oracle_replays_golden(oracle, path):
    g = load(path)
    L = load_library()
    inp = ArrayInput(g["input"])

    def prove_under(spec):
        oracle.set_framing(spec)
        set_framing(spec, L)
        words, _, tr = oracle.prove(inp.view, cfg=tuple(g["pcs_config"]), transcript=True)
        return tr, words_to_proof(words, L)
    try:
        replay_with_sweep(g, prove_under, "oracle")
    finally:
        oracle.set_framing("")
        set_framing("", L)


@pytest.mark.gpu
@pytest.mark.parametrize("path", REF_FILES + SELF_FILES, ids=[os.path.basename(p) for p in REF_FILES + SELF_FILES])
def test_hip_replays_golden(backend, path):
    g = load(path)
    inp = ArrayInput(g["input"])

    def prove_under(spec):
        set_framing(spec, backend.L)
        p = backend.prove(inp, cfg=tuple(g["pcs_config"]))
        return p.transcript(), p
    try:
        set_transcript_log(True, backend.L)
        spec = replay_with_sweep(g, prove_under, "HIP")
        set_framing(spec, backend.L)
        p = backend.prove(inp, cfg=tuple(g["pcs_config"]))
        assert p.verify(cfg=tuple(g["pcs_config"]))[0] == 0
        p.free()
    finally:
        set_transcript_log(False, backend.L)
        set_framing("", backend.L)


def test_divergence_report_names_the_switch():
    """the comparison machinery itself: a golden whose PoW-config step was made under another `mix_u64` is reported at step 0
    with the switch to flip"""
    g = load(SELF_FILES[0])
    bad = json.loads(json.dumps(g["transcript"]))
    bad[0]["digest"] = "00" * 32
    with pytest.raises(AssertionError, match="step 0.*mix_u64"):
        compare_transcripts(g["transcript"], bad, "oracle")
    bad = json.loads(json.dumps(g["transcript"]))
    k = next(i for i, e in enumerate(bad) if e["op"] == "mix_root")
    bad[k]["digest"] = "11" * 32
    with pytest.raises(AssertionError, match=f"step {k}.*hash_node"):
        compare_transcripts(g["transcript"], bad, "oracle")


def selfmade_golden_under(oracle, L, spec, cfg=(8, 1, 0, 4)):
    """an in-memory golden in the harness's format, written by THIS repository's oracle under framing `spec` (pins nothing)"""
    from tests.ref_inputs import unchanged_memory_arrays
    arrays = unchanged_memory_arrays()
    inp = ArrayInput(arrays)
    try:
        oracle.set_framing(spec)
        set_framing(spec, L)
        words, _, tr = oracle.prove(inp.view, cfg=cfg, transcript=True)
        p = words_to_proof(words, L)
        proof = json.loads(p.json())
        doc = {"name": "selfmade", "source": "self-made", "pcs_config": list(cfg),
               "input": {k: np.asarray(v).tolist() for k, v in arrays.items()}, "transcript": tr,
               "commitments": [r.hex() for r in p.commitments()], "interaction_pow": proof["interaction_pow"], "proof": proof}
        p.free()
        return doc
    finally:
        oracle.set_framing("")
        set_framing("", L)


@pytest.mark.parametrize("spec", ["mix_u64=raw,hash_node=rfc,sample_batch=sorted,pcs_mix=blq"])
def test_sweep_identifies_the_framing_a_golden_was_made_under(oracle, spec):
    """The day a reference-produced golden disagrees with the defaults, the replay must say WHICH of the 16 framings it was made
    under.  Here: a golden this repository's oracle writes under a non-default framing; the replay under the defaults fails and
    the sweep names exactly that framing."""
    if FRAMING:
        pytest.skip("CM_REF_FRAMING is set: the sweep starts from that framing")
    L = load_library()
    g = selfmade_golden_under(oracle, L, spec)
    inp = ArrayInput(g["input"])

    def prove_under(s):
        oracle.set_framing(s)
        set_framing(s, L)
        words, _, tr = oracle.prove(inp.view, cfg=tuple(g["pcs_config"]), transcript=True)
        return tr, words_to_proof(words, L)
    try:
        with pytest.raises(AssertionError) as e:
            replay_with_sweep(g, prove_under, "oracle")
        msg = str(e.value)
        assert "replays completely under: " + spec + "." in msg, msg[:600]
    finally:
        oracle.set_framing("")
        set_framing("", L)
