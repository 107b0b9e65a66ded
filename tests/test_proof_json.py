"""`Proof<H>` JSON (cm_proof_json = sonic_rs::to_string(&proof), crates/prover/src/main.rs:86-91, lib.rs:61-73):
* STRICT SCHEMA: every object of the emitted JSON has exactly the keys, in exactly the order, of the reference's serde
  structs (tests/golden/proof_schema.json, extracted from the reference source by tools/rsref/rs_schema.py; the nested Stwo
  `StarkProof` part is restated from upstream and marked unpinned there);
* ROUND TRIP: the flat word stream rebuilt from the parsed JSON equals cm_proof_words, i.e. the JSON carries the whole
  proof and nothing is lost or reordered;  values are M31 numbers, QM31 as [[a, b], [c, d]], hashes as 32 byte numbers,
  Option::None as null.
Host code only: the proof comes from the CPU oracle, travels as words into the library (cm_proof_from_words)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from cairo_m_amd.lib import Proof, load_library, synth_fibonacci

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCHEMA = json.load(open(os.path.join(ROOT, "tests", "golden", "proof_schema.json")))
P = 2**31 - 1


def _keys(obj, want, where):
    assert isinstance(obj, list) and [k for k, _ in obj] == want, (where, [k for k, _ in obj][:12], want[:12])
    return dict(obj)


def json_to_words(text):
    """Parse with key order preserved, check every object against the schema, rebuild the cm_proof_words stream."""
    doc = json.loads(text, object_pairs_hook=lambda pairs: pairs)
    S, U = SCHEMA, SCHEMA["stark_proof_unpinned"]
    top = _keys(doc, S["Proof"], "Proof")
    w = [0x434D5031]

    def q(x):
        (a, b), (c, d) = x
        for v in (a, b, c, d):
            assert isinstance(v, int) and 0 <= v < P
            w.append(v)

    def h(x):
        assert len(x) == 32 and all(isinstance(b, int) and 0 <= b < 256 for b in x)
        w.extend(np.frombuffer(bytes(x), dtype="<u4").tolist())

    sp = _keys(top["stark_proof"], U["StarkProof"], "StarkProof")
    cfg = _keys(sp["config"], U["PcsConfig"], "PcsConfig")
    fri = _keys(cfg["fri_config"], U["FriConfig"], "FriConfig")
    w += [cfg["pow_bits"], fri["log_blowup_factor"], fri["log_last_layer_degree_bound"], fri["n_queries"]]
    # claims
    order = S["opcodes"] + S["Claim"][1:]
    for which, sch, leaf in (("claim", "Claim", "component_claim"), ("interaction_claim", "InteractionClaim", "component_interaction_claim")):
        c = _keys(top[which], S[sch], sch)
        ops = _keys(c["opcodes"], S["opcodes"], "opcodes::" + sch)
        leaves = [_keys(ops[n], S[leaf], n) for n in S["opcodes"]] + [_keys(c[n], S[leaf], n) for n in S[sch][1:]]
        if which == "claim":
            w.append(len(order))
            w += [l["log_size"] for l in leaves]
        else:
            for l in leaves:
                q(l["claimed_sum"])
    pd = _keys(top["public_data"], S["PublicData"], "PublicData")
    regs = [_keys(pd[k], S["VmRegisters"], k) for k in ("initial_registers", "final_registers")]
    w += [regs[0]["pc"], regs[0]["fp"], regs[1]["pc"], regs[1]["fp"], pd["clock"], pd["initial_root"], pd["final_root"]]
    pm = _keys(pd["public_memory"], S["PublicEntries"], "PublicEntries")
    for k in S["PublicEntries"]:
        w.append(len(pm[k]))
        for e in pm[k]:
            if e is None:                      # Option::None
                w += [0] * 7
            else:
                addr, val, clock = e           # (M31, QM31, M31)
                w += [1, addr]
                q(val)
                w.append(clock)
    w += [top["interaction_pow"] & 0xFFFFFFFF, top["interaction_pow"] >> 32]
    w.append(len(sp["commitments"]))
    for c in sp["commitments"]:
        h(c)
    for tree in sp["sampled_values"]:
        w.append(len(tree))
        for col in tree:
            w.append(len(col))
            for s in col:
                q(s)

    def dec(d):
        d = _keys(d, U["MerkleDecommitment"], "MerkleDecommitment")
        w.append(len(d["hash_witness"]))
        for x in d["hash_witness"]:
            h(x)
        w.append(len(d["column_witness"]))
        w.extend(d["column_witness"])

    for d in sp["decommitments"]:
        dec(d)
    for qv in sp["queried_values"]:
        w.append(len(qv))
        w.extend(qv)
    w += [sp["proof_of_work"] & 0xFFFFFFFF, sp["proof_of_work"] >> 32]
    fp = _keys(sp["fri_proof"], U["FriProof"], "FriProof")

    def layer(l):
        l = _keys(l, U["FriLayerProof"], "FriLayerProof")
        w.append(len(l["fri_witness"]))
        for x in l["fri_witness"]:
            q(x)
        dec(l["decommitment"])
        h(l["commitment"])

    layer(fp["first_layer"])
    w.append(len(fp["inner_layers"]))
    for l in fp["inner_layers"]:
        layer(l)
    lp = _keys(fp["last_layer_poly"], U["LinePoly"], "LinePoly")
    w.append(len(lp["coeffs"]))
    for x in lp["coeffs"]:
        q(x)
    w.append(lp["log_size"])
    return np.array(w, dtype=np.uint32)


def proof_from_words(L, words):
    w = np.ascontiguousarray(words, dtype=np.uint32)
    h = C.c_void_p()
    rc = L.cm_proof_from_words(w.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint64(w.size), C.byref(h))
    assert rc == 0
    return Proof(L, h)


@pytest.mark.parametrize("cfg", [None, (5, 1, 2, 20)])
def test_proof_json_schema_and_round_trip(oracle, cfg):
    L = load_library()
    inp = synth_fibonacci(9)
    words, _ = oracle.prove(inp.view, cfg=cfg or (16, 1, 0, 80))
    p = proof_from_words(L, words)
    assert np.array_equal(p.words(), words)           # words -> object -> words
    text = p.json()
    back = json_to_words(text)                         # object -> JSON -> (strict schema) -> words
    assert back.size == words.size and np.array_equal(back, words)
    doc = json.loads(text)
    assert len(doc["claim"]["opcodes"]) == 26 and len(doc["stark_proof"]["commitments"]) == 4
    assert doc["public_data"]["public_memory"]["program"][0] is not None
    assert p.verify(cfg)[0] == 0
    p.free()
    inp.free()


def test_schema_fixture_shape():
    assert SCHEMA["Proof"] == ["claim", "interaction_claim", "public_data", "stark_proof", "interaction_pow"]
    assert len(SCHEMA["opcodes"]) == 26 and SCHEMA["Claim"][0] == "opcodes" and len(SCHEMA["Claim"]) == 9
