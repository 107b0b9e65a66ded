#!/usr/bin/env python3
"""Golden vectors for the poseidon2 component (crates/prover/src/components/poseidon2.rs), derived mechanically from the
reference's source like the other components (tools/rsref/rs_eval.py, rs_witness.py), but through the full interpreter
(tools/rsref/rs_interp.py): `evaluate()` and the `write_trace` row closure are round loops over arrays with `&mut`
elements, calling the file's own helper functions (`apply_m4`, `apply_external_round_matrix`,
`apply_internal_round_matrix`, `square`), all of which are parsed from the reference text and interpreted.

The ROUND CONSTANTS are the one input that cannot come from the reference tree: the reference takes them from the
un-vendored `zkhash` crate at build time (build.rs:25-103).  They are read from cairo_m_amd/csrc/air/poseidon2_consts.hpp,
which tools/gen_poseidon2_m31.py regenerates from the public Poseidon2 parameter procedure and which the reference's own
known-answer test pins (tests/golden/poseidon2_kat.json).

Appends `Poseidon2C` to tests/golden/air_eval_vectors.json and `poseidon2` to tests/golden/air_witness_vectors.npz.
Usage (build container only):  python tools/rsref/rs_eval.py && python tools/rsref/rs_witness.py && python tools/rsref/rs_poseidon2.py
"""
import json
import os
import random
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from rs_interp import Env, Felt, Interp, Packed, N_LANES, P, parse_block, parse_expr, parse_fn, standard_globals  # noqa: E402
from rs_eval import REF, extract_fn_body, strip_comments  # noqa: E402
from rs_witness import Enabler, LookupData, Slots, file_consts  # noqa: E402

SRC = f"{REF}/prover/src/components/poseidon2.rs"
T, FULL_ROUNDS, PARTIAL_ROUNDS = 16, 8, 14      # crates/prover/src/poseidon2.rs (T), build.rs (round numbers); asserted below


def constants():
    h = open(os.path.join(ROOT, "cairo_m_amd", "csrc", "air", "poseidon2_consts.hpp")).read()
    nums = lambda name: [int(x) for x in re.findall(r"(\d+)u", h[h.index(name):].split(";")[0].split("=", 1)[1])]
    ext = nums("P2_EXTERNAL_RC")
    assert len(ext) == FULL_ROUNDS * T
    return ([[Felt(ext[r * T + i]) for i in range(T)] for r in range(FULL_ROUNDS)], [Felt(x) for x in nums("P2_INTERNAL_RC")],
            [Felt(x) for x in nums("P2_INTERNAL_DIAG")])


def fn_source(src, name):
    i = src.index(f"fn {name}")
    b = src.index("{", src.index(")", i))
    # skip a `where` clause: the body brace is the first `{` after the signature that is not inside <>
    depth, j = 0, b
    while True:
        depth += src[j] == "{"
        depth -= src[j] == "}"
        j += 1
        if depth == 0:
            break
    return src[i:j]


def make_interp():
    src = strip_comments(open(SRC).read())
    ext, internal, diag = constants()
    g = standard_globals()
    g.update({"T": T, "FULL_ROUNDS": FULL_ROUNDS, "PARTIAL_ROUNDS": PARTIAL_ROUNDS, "EXTERNAL_ROUND_CONSTS": ext,
              "INTERNAL_ROUND_CONSTS": internal, "INTERNAL_MATRIX": diag, "SECURE_EXTENSION_DEGREE": 4,
              "std::array::from_fn": lambda f: [f(i) for i in range(T)]})     # every array of this file has T elements
    interp = Interp(g)
    file_consts(src, interp)
    assert g["N_TRACE_COLUMNS"] == 1 + T * (1 + FULL_ROUNDS * 3) + 3 * PARTIAL_ROUNDS == 443
    top = Env()
    for name in ("apply_m4", "apply_external_round_matrix", "apply_internal_round_matrix", "square"):
        nm, params, body = parse_fn(fn_source(src, name))
        g[nm] = interp.make_closure(("closure", [("pname", p) for p in params], body), top)
    return src, interp, g


class Recorder:
    def __init__(self, trace):
        self.trace, self.k, self.constraints, self.relations = trace, 0, [], []

    def next_trace_mask(self):
        v = self.trace[self.k]
        self.k += 1
        return Felt(v)

    def add_constraint(self, x):
        self.constraints.append(x.v)

    def add_to_relation(self, e):
        self.relations.append(e)

    def finalize_logup_in_pairs(self):
        self.finalized = "pairs"


class _Rel:
    poseidon2 = "poseidon2"


class _Self:
    relations = _Rel()


def eval_vectors(src, interp, g):
    body = extract_fn_body(src, r"fn evaluate<E: EvalAtRow>\(&self, mut eval: E\) -> E \{")
    block = parse_block("{" + body + "}")
    g.update({"E::EF::from": lambda x: x, "E::F::from": lambda x: x, "E::EF::one": lambda: Felt(1), "E::F::one": lambda: Felt(1),
              "RelationEntry::new": lambda rel, mult, vals: [rel, mult.v, [v.v for v in vals]]})
    rng = random.Random(0x9052)
    n = g["N_TRACE_COLUMNS"]
    rows = []
    for tr in ([rng.randrange(P) for _ in range(n)], [1] + [rng.randrange(P) for _ in range(n - 1)], [0] * n):
        rec = Recorder(tr)
        env = Env()
        env.vars.update({"eval": rec, "self": _Self()})
        interp.eval(block, env)
        assert rec.k == n
        rows.append({"trace": tr, "constraints": rec.constraints, "relations": rec.relations, "finalize": rec.finalized})
    return n, rows


def witness_cells(src, interp, g, inputs, keep_lookup=None):
    """inputs: (n, 16) hash inputs = NodeData::to_hash_input of initial_tree ++ final_tree (adapter/mod.rs:165-176)."""
    rx = r"\.for_each\(\|\(row_index, \(mut row, mut state, lookup_data\)\)\| \{"
    body = extract_fn_body(src, rx)
    block = parse_block("{" + body + "}")
    n = len(inputs)
    log_size = max(4, (max(n, 1) - 1).bit_length())
    n_rows = 1 << log_size
    out = np.zeros((g["N_TRACE_COLUMNS"], n_rows), dtype=np.uint32)
    outer = Env()
    outer.vars.update({"enabler_col": Enabler(n), "zero": Packed.broadcast(Felt(0))})
    for vec_row in range(n_rows // N_LANES):
        lanes = [[int(x) for x in inputs[vec_row * N_LANES + i]] if vec_row * N_LANES + i < n else [0] * T for i in range(N_LANES)]
        state = [Packed([Felt(l[x]) for l in lanes]) for x in range(T)]     # packed_inputs: poseidon2.rs:181-191
        env = Env(outer)
        row, ld = Slots(), LookupData()
        env.vars.update({"row_index": vec_row, "row": row, "state": state, "lookup_data": ld})
        interp.eval(block, env)
        assert sorted(row.d) == list(range(out.shape[0]))
        if keep_lookup is not None:
            keep_lookup.append(ld)      # tools/rsref/rs_logup.py
        for c in range(out.shape[0]):
            out[c, vec_row * N_LANES:(vec_row + 1) * N_LANES] = [x.v for x in row.d[c].lanes]
    return out


def main():
    src, interp, g = make_interp()
    n, rows = eval_vectors(src, interp, g)
    path = os.path.join(ROOT, "tests", "golden", "air_eval_vectors.json")
    doc = json.load(open(path))
    doc["components"]["Poseidon2C"] = {"n_trace": n, "source": "crates/prover/src/components/poseidon2.rs", "rows": rows}
    json.dump(doc, open(path, "w"), separators=(",", ":"))
    print(f"Poseidon2C {n} trace cols, {len(rows[0]['constraints'])} constraints, {len(rows[0]['relations'])} relation entries")
    # witness: the hash inputs of the all-opcode run of rs_witness.py (every node of both partial Merkle trees)
    from cairo_m_amd.lib import prover_input_arrays, vm_run
    from cairo_m_amd.workloads import all_opcodes_program
    from rs_witness import ITERS, SEED
    prog, _ = all_opcodes_program(ITERS, SEED)
    inp = vm_run(prog, entry_pc=0, args=(), n_returns=0)
    a = prover_input_arrays(inp.view)
    nodes = np.concatenate([a["initial_tree"], a["final_tree"]])
    inputs = np.zeros((nodes.shape[0], T), dtype=np.int64)
    inputs[:, 0], inputs[:, 1] = nodes[:, 2], nodes[:, 3]            # (left_value, right_value, 0 x 14): merkle.rs:127-133
    all_inputs = inputs
    inputs = inputs[:200]                                             # 200 live rows + 56 padding rows kept cell by cell (443 columns each)
    cells = witness_cells(src, interp, g, inputs)
    # FULL LENGTH: every node of both trees (all live rows + all padding rows of the 2^log-row trace) through the same
    # interpreted closure; the fixture keeps one Blake2s-256 digest per column (443 x 32 bytes) instead of ~1.8 M cells
    import hashlib
    full = witness_cells(src, interp, g, all_inputs)
    digests = np.stack([np.frombuffer(hashlib.blake2s(np.ascontiguousarray(full[c]).tobytes()).digest(), dtype=np.uint8)
                        for c in range(full.shape[0])])
    assert np.array_equal(full[:, :192], cells[:, :192])
    inp.free()
    wp = os.path.join(ROOT, "tests", "golden", "air_witness_vectors.npz")
    old = dict(np.load(wp))
    old["poseidon2"] = cells
    old["poseidon2_inputs"] = inputs.astype(np.uint32)
    old["poseidon2_full_digests"] = digests                          # blake2s(column c as little-endian u32), full-length trace
    old["poseidon2_full_shape"] = np.array([full.shape[0], full.shape[1], all_inputs.shape[0]])
    np.savez_compressed(wp, **old)
    print("poseidon2 witness", cells.shape, os.path.getsize(wp), "bytes")


if __name__ == "__main__":
    main()
