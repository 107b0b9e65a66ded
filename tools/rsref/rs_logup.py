#!/usr/bin/env python3
"""Golden LOGUP vectors for the 26 opcode components, derived mechanically from the reference's
`InteractionClaim::write_interaction_trace` (e.g. crates/prover/src/components/opcodes/store_fp_imm.rs:308-440) — build container
only, like the other tools/rsref scripts.

What is interpreted, from the reference text:
  1. `Claim::write_trace`'s row closure (tools/rsref/rs_witness.py), this time KEEPING the `lookup_data` it fills
     (`*lookup_data.memory[0] = [..]`, `*lookup_data.range_check_20[2] = ..`) — on the all-opcode program of rs_witness.py;
  2. every column block of `write_interaction_trace`
         let mut col = interaction_trace.new_col();
         ( col.par_iter_mut(), &...lookup_data.K[i], &...lookup_data.K[j] ).into_par_iter().enumerate().for_each(|(i, (writer, a, b))| { .. writer.write_frac(num, den); });
         col.finalize_col();
     — which lookup arrays it zips and the closure that turns them into ONE fraction per row (numerators from the enabler or
     -1, denominators `relations.<kind>.combine(tuple)`, the pairing `num_a * den_b + num_b * den_a` over `den_a * den_b`).
The relation parameters (z, alpha) are seeded random QM31 values stored with the vectors; `combine` is the one Stwo function
restated here (`sum alpha^i * v_i - z`: crates/prover/src/relations.rs uses stwo_constraint_framework::relation!).

Output: tests/golden/air_logup_vectors.npz — per component `<name>` of shape (n_columns, 2^log_size, 4): entry [j, r] = the sum
of the fractions of columns 0..j at row r as a QM31 (what LogupTraceGenerator::finalize_col leaves in column j before
finalize_last prefix-sums the LAST column over the rows), plus `rel_z`, `rel_alpha` (8 x 4 words, air::RelId order).  Data only.
tests/test_gpu_logup_golden.py feeds the reference-derived TRACE (air_witness_vectors.npz) to the HIP k_logup through
cm_interaction_write and compares.

Usage (build container only):  python tools/rsref/rs_logup.py"""
import os
import random
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import rs_witness as W  # noqa: E402
from rs_interp import Env, Felt, Interp, Packed, N_LANES, parse_expr, standard_globals  # noqa: E402
from rs_eval import REF, extract_fn_body, opcode_constants, strip_comments  # noqa: E402

P = 2**31 - 1
REL_ORDER = ["registers", "memory", "merkle", "poseidon2", "range_check_8", "range_check_16", "range_check_20", "bitwise"]


# ---- QM31 = (a + b i) + (c + d i) u, i^2 = -1, u^2 = 2 + i, as 4-tuples of ints ----
def cmul(x, y):
    return ((x[0] * y[0] - x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)


def qmul(x, y):
    a, b, c, d = x[:2], x[2:], y[:2], y[2:]
    ac, bd = cmul(a, c), cmul(b, d)
    r = cmul(bd, (2, 1))
    ad, bc = cmul(a, d), cmul(b, c)
    return ((ac[0] + r[0]) % P, (ac[1] + r[1]) % P, (ad[0] + bc[0]) % P, (ad[1] + bc[1]) % P)


def qinv(x):
    # 1 / (A + B u) = (A - B u) / (A^2 - (2 + i) B^2) over CM31, then the CM31 inverse via the norm
    A, B = x[:2], x[2:]
    b2 = cmul(B, B)
    den = ((cmul(A, A)[0] - cmul(b2, (2, 1))[0]) % P, (cmul(A, A)[1] - cmul(b2, (2, 1))[1]) % P)
    n = (den[0] * den[0] + den[1] * den[1]) % P
    ni = pow(n, P - 2, P)
    di = (den[0] * ni % P, (-den[1]) * ni % P)
    na, nb = cmul(A, di), cmul(((-B[0]) % P, (-B[1]) % P), di)
    return (na[0], na[1], nb[0], nb[1])


class PQ:
    """PackedQM31: 16 lanes of QM31"""
    __slots__ = ("lanes",)

    def __init__(self, lanes):
        self.lanes = list(lanes)
        assert len(self.lanes) == N_LANES

    @staticmethod
    def of(x):
        if isinstance(x, PQ):
            return x
        if isinstance(x, Packed):
            return PQ([(f.v, 0, 0, 0) for f in x.lanes])
        if isinstance(x, Felt):
            return PQ([(x.v, 0, 0, 0)] * N_LANES)
        raise TypeError(f"PackedQM31 from {type(x).__name__}")

    def __add__(self, o): return PQ([tuple((a + b) % P for a, b in zip(x, y)) for x, y in zip(self.lanes, PQ.of(o).lanes)])
    def __sub__(self, o): return PQ([tuple((a - b) % P for a, b in zip(x, y)) for x, y in zip(self.lanes, PQ.of(o).lanes)])
    def __mul__(self, o): return PQ([qmul(x, y) for x, y in zip(self.lanes, PQ.of(o).lanes)])
    def __neg__(self): return PQ([tuple((-a) % P for a in x) for x in self.lanes])


class Relation:
    """stwo_constraint_framework relation!: combine(values) = sum_i alpha^i * values[i] - z"""

    def __init__(self, z, alpha):
        self.z, self.alpha = z, alpha

    def combine(self, values):
        values = [v for v in values]
        acc = PQ([tuple((-c) % P for c in self.z)] * N_LANES)
        ap = (1, 0, 0, 0)
        for v in values:
            pv = PQ.of(v)
            acc = acc + PQ([qmul(ap, lane) for lane in pv.lanes])
            ap = qmul(ap, self.alpha)
        return acc


class Relations:
    pass


class Writer:
    def __init__(self, sink):
        self.sink = sink

    def write_frac(self, num, den):
        self.sink.append((PQ.of(num), PQ.of(den)))


class Col:
    def __init__(self, n_vec_rows):
        self.fracs = [[] for _ in range(n_vec_rows)]
        self.done = False

    def par_iter_mut(self):
        return [Writer(f) for f in self.fracs]

    def finalize_col(self):
        assert all(len(f) == 1 for f in self.fracs), "one write_frac per packed row"
        self.done = True


class LogupGen:
    """LogupTraceGenerator as far as write_interaction_trace drives it: new_col() / finalize_col() in column order"""

    def __init__(self, n_vec_rows):
        self.n, self.cols = n_vec_rows, []

    def new_col(self):
        c = Col(self.n)
        self.cols.append(c)
        return c


class LogupInterp(Interp):
    def method(self, r, name, a):
        from rs_interp import deref
        r0 = deref(r)
        if name in ("into_par_iter", "par_iter") and isinstance(r0, tuple):
            return [tuple(x) for x in zip(*r0)]          # rayon's zip of a tuple of parallel iterators
        if name in ("into_par_iter", "par_iter"):
            return r0
        if name == "ilog2":
            return r0.bit_length() - 1
        return super().method(r, name, a)


def interaction_body(src):
    """write_interaction_trace up to (not including) finalize_last"""
    body = extract_fn_body(src, r"pub fn write_interaction_trace\(")
    return body[:body.index("let (trace, claimed_sum)")]


def logup_columns(src, lookups, n_rows, n_live, consts, relations):
    """interpret write_interaction_trace of `src` on the per-packed-row lookup data -> (n_columns, n_rows, 4): per column and row the
    running sum of the fractions of columns 0..j"""
    from rs_interp import Struct, parse_block
    g = standard_globals()
    g.update(consts)
    n_vec = n_rows // N_LANES
    gen = LogupGen(n_vec)
    g.update({"PackedQM31::from": PQ.of, "PackedQM31::one": lambda: PQ([(1, 0, 0, 0)] * N_LANES),
              "PackedQM31::zero": lambda: PQ([(0, 0, 0, 0)] * N_LANES),
              "LogupTraceGenerator::new": lambda log_size: gen, "Enabler::new": lambda n: W.Enabler(n)})
    interp = LogupInterp(g)
    W.file_consts(src, interp)
    # lookup_data.<kind>[i] = the per-packed-row values the write_trace closure stored
    ld_all = W.LookupData()
    for kind in ("memory", "registers", "range_check_8", "range_check_16", "range_check_20", "bitwise", "merkle", "poseidon2"):
        slots = {}
        for vec_row, ld in enumerate(lookups):
            for idx, val in getattr(ld, kind).d.items():
                slots.setdefault(idx, [None] * n_vec)[vec_row] = val
        setattr(ld_all, kind, [slots[i] for i in range(len(slots))] if slots else [])
    outer = Env()
    outer.vars.update({"relations": relations, "interaction_claim_data": Struct(lookup_data=ld_all, non_padded_length=n_live)})
    interp.eval(parse_block("{" + interaction_body(src) + "}"), outer)
    assert gen.cols and all(c.done for c in gen.cols)
    cum = np.zeros((len(gen.cols), n_rows, 4), dtype=np.uint32)
    running = [[(0, 0, 0, 0)] * N_LANES for _ in range(n_vec)]
    for j, col in enumerate(gen.cols):
        for vec_row in range(n_vec):
            num, den = col.fracs[vec_row][0]
            for lane in range(N_LANES):
                frac = qmul(num.lanes[lane], qinv(den.lanes[lane]))
                running[vec_row][lane] = tuple((a + b) % P for a, b in zip(running[vec_row][lane], frac))
                cum[j, vec_row * N_LANES + lane] = running[vec_row][lane]
    return cum


def logup_table(src, name, per_row, n_rows, relations):
    """write_interaction_trace of a lookup-table component (range_check_macro.rs:125-146, bitwise.rs:170-192): the relation is the
    function's first parameter, `interaction_claim_data.<name>` the packed [table values.., multiplicity] rows"""
    from rs_interp import Struct, parse_block
    g = standard_globals()
    n_vec = n_rows // N_LANES
    gen = LogupGen(n_vec)
    g.update({"PackedQM31::from": PQ.of, "LogupTraceGenerator::new": lambda log_size: gen})
    interp = LogupInterp(g)
    outer = Env()
    outer.vars.update({name: getattr(relations, name), "interaction_claim_data": Struct(**{name: per_row})})
    interp.eval(parse_block("{" + interaction_body(src) + "}"), outer)
    assert len(gen.cols) == 1 and gen.cols[0].done
    cum = np.zeros((1, n_rows, 4), dtype=np.uint32)
    for vec_row in range(n_vec):
        num, den = gen.cols[0].fracs[vec_row][0]
        for lane in range(N_LANES):
            cum[0, vec_row * N_LANES + lane] = qmul(num.lanes[lane], qinv(den.lanes[lane]))
    return cum


def main():
    from cairo_m_amd.lib import prover_input_arrays, vm_run
    from cairo_m_amd.workloads import all_opcodes_program
    prog, steps = all_opcodes_program(W.ITERS, W.SEED)
    inp = vm_run(prog, entry_pc=0, args=(), n_returns=0)
    arrs = prover_input_arrays(inp.view)
    consts = opcode_constants()
    rng = random.Random(0x106)
    rel_z = {k: tuple(rng.randrange(1, P) for _ in range(4)) for k in REL_ORDER}
    rel_a = {k: tuple(rng.randrange(1, P) for _ in range(4)) for k in REL_ORDER}
    relations = Relations()
    for k in REL_ORDER:
        setattr(relations, k, Relation(rel_z[k], rel_a[k]))
    out = {"iters": np.array([W.ITERS]), "seed": np.array([W.SEED]),
           "rel_z": np.array([rel_z[k] for k in REL_ORDER], dtype=np.uint32), "rel_alpha": np.array([rel_a[k] for k in REL_ORDER], dtype=np.uint32)}
    for cid, fname in enumerate(W.OPCODE_FILES):
        bundles = arrs[f"bundles{cid}"]
        lookups = []
        cols = W.interpret_component(fname, bundles, arrs["data_accesses"], consts, keep_lookup=lookups)
        if cols is None:
            cols = W.interpret_prepacked(fname, bundles, arrs["data_accesses"], consts, keep_lookup=lookups)
        src = strip_comments(open(f"{REF}/prover/src/components/opcodes/{fname}.rs").read())
        out[fname] = logup_columns(src, lookups, cols.shape[1], len(bundles), consts, relations)
        print(f"{cid:2d} {fname:28s} {len(bundles):4d} live rows, {out[fname].shape[0]} LogUp columns x {cols.shape[1]} rows")
    # the builtins with a regular closure (rows as in rs_witness.py: memory, merkle, synthetic clock updates)
    consts2, mem, tree, cu = W.builtin_inputs(arrs, consts)
    for name, rows in (("memory", mem), ("merkle", tree), ("clock_update", cu)):
        lookups = []
        cols = W.interpret_builtin(name, rows, len(rows), consts2, keep_lookup=lookups)
        src = strip_comments(open(f"{REF}/prover/src/components/{name}.rs").read())
        out[name] = logup_columns(src, lookups, cols.shape[1], len(rows), consts2, relations)
        print(f"   {name:28s} {len(rows):4d} live rows, {out[name].shape[0]} LogUp columns x {cols.shape[1]} rows")
    # poseidon2: the 200 hash inputs of the witness golden (rs_poseidon2.py), its closure through the full interpreter
    import rs_poseidon2 as P2
    p2src, p2interp, p2g = P2.make_interp()
    nodes = np.concatenate([arrs["initial_tree"], arrs["final_tree"]])
    p2in = np.zeros((nodes.shape[0], P2.T), dtype=np.int64)
    p2in[:, 0], p2in[:, 1] = nodes[:, 2], nodes[:, 3]
    p2in = p2in[:200]
    lookups = []
    cells = P2.witness_cells(p2src, p2interp, p2g, p2in, keep_lookup=lookups)
    out["poseidon2"] = logup_columns(p2src, lookups, cells.shape[1], len(p2in), {k: v for k, v in p2g.items() if isinstance(v, int)}, relations)
    print(f"   {'poseidon2':28s} {len(p2in):4d} live rows, {out['poseidon2'].shape[0]} LogUp columns x {cells.shape[1]} rows")
    # the four lookup tables: one fraction per row, multiplicity / combine(table entry).  Inputs: seeded table values and
    # multiplicities on 64 rows (the per-op entry point takes the log size and the preprocessed columns from the caller), stored
    # with the vectors: `<name>_values` (n_preprocessed, 64), `<name>_mults` (64)
    PRE = f"{REF}/prover/src/preprocessed"
    macro = strip_comments(open(f"{PRE}/range_check/range_check_macro.rs").read())
    rc_mod = open(f"{PRE}/range_check/mod.rs").read()
    tables = [(f"range_check_{bits}", macro.replace("[<range_check_ $bit_size>]", f"range_check_{bits}"), 1)
              for bits, _, _ in re.findall(r"define_range_check!\((\d+), (\w+), (\w+)\);", rc_mod)]
    tables.append(("bitwise", strip_comments(open(f"{PRE}/bitwise.rs").read()), 4))
    n_tab = 64
    for name, tsrc, n_vals in tables:
        vals = np.array([[rng.randrange(P) for _ in range(n_tab)] for _ in range(n_vals)], dtype=np.int64)
        mults = np.array([rng.randrange(1 << 20) for _ in range(n_tab)], dtype=np.int64)
        per_row = [[Packed([Felt(int(vals[k][v * N_LANES + i])) for i in range(N_LANES)]) for k in range(n_vals)] +
                   [Packed([Felt(int(mults[v * N_LANES + i])) for i in range(N_LANES)])] for v in range(n_tab // N_LANES)]
        out[name] = logup_table(tsrc, name, per_row, n_tab, relations)
        out[name + "_values"], out[name + "_mults"] = vals.astype(np.uint32), mults.astype(np.uint32)
        print(f"   {name:28s} {n_tab:4d} rows, {out[name].shape[0]} LogUp column")
    inp.free()
    path = os.path.join(ROOT, "tests", "golden", "air_logup_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
