#!/usr/bin/env python3
"""SELF-MADE golden in the format of the reference harness's ref_*.json (integration/prover-hip/tests/golden_dump.rs), produced
by THIS repository's CPU oracle: tests/golden/selfmade_unchanged_memory.json.  It pins nothing about Stwo — its only purpose
is to keep tests/test_ref_golden.py's loader / replay / step-by-step comparison exercised until a reference-produced file
exists (and its "source" field says so).  A small PCS config (4 queries) keeps the proof object small.

    python tools/make_selfmade_golden.py
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = (8, 1, 0, 4)      # pow_bits, log_blowup_factor, log_last_layer_degree_bound, n_queries


def main():
    import numpy as np
    from cairo_m_amd.lib import Proof, load_library
    from tests.oracle_binding import Oracle
    from tests.ref_inputs import unchanged_memory_arrays
    from cairo_m_amd.lib import ArrayInput
    arrays = unchanged_memory_arrays()
    inp = ArrayInput(arrays)
    orc = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    words, _, transcript = orc.prove(inp.view, cfg=CFG, transcript=True)
    L = load_library()
    h = C.c_void_p()
    assert L.cm_proof_from_words(words.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint64(words.size), C.byref(h)) == 0
    p = Proof(L, h)
    proof = json.loads(p.json())
    roots = [r.hex() for r in p.commitments()]
    doc = {"name": "unchanged_memory",
           "source": "SELF-MADE by this repository's CPU oracle (tools/make_selfmade_golden.py): NOT a reference vector, pins nothing; "
                     "it only exercises tests/test_ref_golden.py",
           "pcs_config": list(CFG),
           "input": {k: (np.asarray(v).tolist()) for k, v in arrays.items()},
           "transcript": transcript, "commitments": roots, "interaction_pow": proof["interaction_pow"], "verified": True,
           "proof": proof}
    path = os.path.join(ROOT, "tests", "golden", "selfmade_unchanged_memory.json")
    with open(path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes,", len(transcript), "transcript steps")


if __name__ == "__main__":
    main()
