"""The `eval` half of the AIR descriptions (cairo_m_amd/csrc/air/*.hpp: column order, constraints, relation entries of
every component) against golden vectors derived MECHANICALLY from the reference's own `FrameworkEval::evaluate` text
(tools/rsref/rs_eval.py rewrites each statement of the Rust function body into Python and executes it; no hand
transcription).  For every interpreted component and every seeded row: the constraint values (add_constraint order) and
the relation entries (relation, multiplicity, tuple; add_to_relation order) produced by the oracle's evaluator of the
description must equal the vectors — on arbitrary field elements, so a wrong coefficient, sign, column order or lookup
tuple cannot hide behind a valid trace.  The HIP evaluators instantiate the same descriptions and are compared with the
oracle cell by cell in tests/test_gpu_components.py."""
import ctypes as C
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "air_eval_vectors.json")))["components"]
REL_ID = {"registers": 0, "memory": 1, "merkle": 2, "poseidon2": 3, "range_check_8": 4, "range_check_16": 5,
          "range_check_20": 6, "bitwise": 7}          # air::RelId (relations.rs:7-44, draw order components/mod.rs:311-323)
P = 2**31 - 1


def _names(oracle):
    oracle.L.orc_component_name.restype = C.c_char_p
    return {oracle.L.orc_component_name(C.c_int(c)).decode(): c for c in range(34)}


# reference column id (PreProcessedColumn::id: range_check/mod.rs:69-73, bitwise.rs:338-342) -> air::PreprocId
PP_INDEX = {"bitwise_stacked_col_0": 0, "bitwise_stacked_col_1": 1, "bitwise_stacked_col_2": 2, "bitwise_stacked_col_3": 3,
            "range_check_8": 4, "range_check_16": 5, "range_check_20": 6}


def _eval_row(oracle, cid, row, preproc=None):
    cap = 4096
    r = np.ascontiguousarray(row, dtype=np.uint32)
    pp = np.zeros(7, dtype=np.uint32)
    for cid_str, v in (preproc or {}).items():
        pp[PP_INDEX[cid_str]] = v
    cons, ents = np.zeros(cap, dtype=np.uint32), np.zeros(cap, dtype=np.uint32)
    nc, ne = C.c_uint32(0), C.c_uint32(0)
    u = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
    nb = oracle.L.orc_component_eval_row(C.c_int(cid), u(r), u(pp), u(cons), C.byref(nc), u(ents), C.byref(ne), C.c_uint32(cap))
    assert nb >= 0
    out, i, e = [], 0, ents[:ne.value].tolist()
    while i < len(e):
        n = e[i + 2]
        out.append([e[i], e[i + 1], e[i + 3:i + 3 + n]])
        i += 3 + n
    return cons[:nc.value].tolist(), out, nb


def test_every_interpreted_component_is_covered(oracle):
    names = _names(oracle)
    assert set(GOLD) <= set(names)
    # all 26 opcode components + memory, merkle, clock_update, poseidon2 + the four lookup tables (tools/rsref/rs_lookup.py)
    assert len(GOLD) == 34 and sorted(names[n] for n in GOLD) == list(range(34))


@pytest.mark.parametrize("name", sorted(GOLD))
def test_eval_matches_reference_derived_vectors(oracle, name):
    cid = _names(oracle)[name]
    g = GOLD[name]
    for k, row in enumerate(g["rows"]):
        assert len(row["trace"]) == g["n_trace"]
        cons, ents, n_batches = _eval_row(oracle, cid, row["trace"], row.get("preproc"))
        assert cons == row["constraints"], (name, k, "constraint values")
        want = []
        for rel, mult, vals in row["relations"]:
            rid = REL_ID[rel]
            want.append([rid, mult, vals])
        # the reference pads Memory tuples implicitly (relation size 6, missing values = 0 in combine): compare the
        # given prefix and require the rest to be zero
        assert len(ents) == len(want), (name, k, len(ents), len(want))
        for j, (got, w) in enumerate(zip(ents, want)):
            assert got[0] == w[0] and got[1] == w[1], (name, k, j, "relation / multiplicity")
            assert got[2][:len(w[2])] == w[2] and not any(got[2][len(w[2]):]), (name, k, j, "tuple", got[2], w[2])
        if row["finalize"] == "pairs":
            assert n_batches == (len(want) + 1) // 2
        else:   # finalize_logup(): one fraction per batch (the lookup tables)
            assert row["finalize"] == "single" and n_batches == len(want)
