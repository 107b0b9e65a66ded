#!/usr/bin/env python3
"""Golden `evaluate` vectors for the four lookup-table components — range_check_8 / 16 / 20 and bitwise — derived from the
reference's source text the same way tools/rsref/rs_eval.py derives the other thirty (build container only).

    python tools/rsref/rs_lookup.py        appends RangeCheck8C / RangeCheck16C / RangeCheck20C / BitwiseC to
                                           tests/golden/air_eval_vectors.json

Sources: `define_range_check!` (crates/prover/src/preprocessed/range_check/range_check_macro.rs:171-184, instantiated three times
in range_check/mod.rs:35-43) and `impl FrameworkEval for Eval` (crates/prover/src/preprocessed/bitwise.rs:211-228).  Their
`evaluate` bodies read PREPROCESSED columns by id; what the interpreter needs beyond rs_eval.py's statement forms is therefore
  * `eval.get_preprocessed_column(id)`: the recorder hands out the row's value for that id and records the id,
  * the ids themselves: `RangeCheck::new(n).id()` / `BitwiseCol::new(i, bits).id()` are `format!("...{}", field)` — the format
    strings are read from the reference text (range_check/mod.rs:69-73, bitwise.rs:338-342),
  * `let [a, b, c, d] = std::array::from_fn(|i| <expr>)`,
  * the macro parameters: `[<LOG_SIZE_RC_ $bit_size>]` -> LOG_SIZE_RC_<n>, `&self.relation` -> the relation type named in the
    macro invocation / the struct field (`crate::relations::Bitwise`).
Every row carries "preproc": {column id: value}; tests map the ids onto the library's column order
(air::PreprocId: bitwise_stacked_col_0..3, range_check_8, range_check_16, range_check_20)."""
import json
import os
import random
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rs_eval as R  # noqa: E402

PRE = f"{R.REF}/prover/src/preprocessed"


def fmt_of(src, struct_impl_re):
    """the `format!("..{}", self.<field>)` of an `fn id(&self)` inside the impl that matches `struct_impl_re`"""
    m = re.search(struct_impl_re, src)
    body = src[m.end():]
    f = re.search(r"fn id\(&self\) -> PreProcessedColumnId \{\s*PreProcessedColumnId \{\s*id: format!\(\"([^\"]*)\", self\.(\w+)\)", body)
    return f.group(1), f.group(2)


class PreRecorder(R.Recorder):
    def __init__(self, trace, preproc):
        super().__init__(trace)
        self.preproc, self.read_ids = preproc, []

    def get_preprocessed_column(self, cid):
        self.read_ids.append(cid)
        return R.Felt(self.preproc[cid])


def run_block(stmts, env):
    """rs_eval.run_block plus array destructuring from `std::array::from_fn(|i| expr)`"""
    for st in stmts:
        s = re.sub(r"\s+", " ", st[1]).strip() if st[0] == "stmt" else None
        m = s and re.match(r"let \[([\w, ]+)\] = std::array::from_fn\(\|(\w+)\| (.*)\)$", s)
        if m:
            names = [n.strip() for n in m.group(1).split(",") if n.strip()]
            expr = R.rust_expr_to_py(m.group(3))
            for i, nme in enumerate(names):
                env[nme] = eval(expr, {"__builtins__": {}}, {**env, m.group(2): i})
            continue
        R.run_block([st], env)


def main():
    rng = random.Random(0x100C)
    consts = {"P": R.P}
    rc_mod = open(f"{PRE}/range_check/mod.rs").read()
    invocations = re.findall(r"define_range_check!\((\d+), (\w+), (\w+)\);", rc_mod)      # (bits, module, relation type)
    for bits, _, _ in invocations:
        consts[f"LOG_SIZE_RC_{bits}"] = int(bits)
    rc_fmt, rc_field = fmt_of(R.strip_comments(rc_mod), r"impl PreProcessedColumn for RangeCheck \{")
    assert rc_field == "range"
    macro = R.strip_comments(open(f"{PRE}/range_check/range_check_macro.rs").read())
    rc_body = R.extract_fn_body(macro, r"fn evaluate<E: EvalAtRow>\(&self, mut eval: E\) -> E \{")
    bw_src = R.strip_comments(open(f"{PRE}/bitwise.rs").read())
    bw_body = R.extract_fn_body(bw_src, r"fn evaluate<E: EvalAtRow>\(&self, mut eval: E\) -> E \{")
    bw_fmt, bw_field = fmt_of(bw_src, r"impl PreProcessedColumn for BitwiseCol \{")
    assert bw_field == "col_index"
    bw_consts = R.file_constants(bw_src, consts)
    # the relation a lookup component writes to: the third macro argument / the type of `Eval::relation`
    bw_rel = re.search(r"pub struct Eval \{[^}]*pub relation: crate::relations::(\w+),", bw_src).group(1)

    def snake(name):
        return re.sub(r"(?<=[a-z])(?=[A-Z0-9])|(?<=[0-9])(?=[A-Z])", "_", name).lower().replace("check", "check_").replace("check__", "check_")

    class RangeCheckCol:
        def __init__(self, n): self.range = n
        def id(self): return rc_fmt.replace("{}", str(self.range))

    class BitwiseCols:
        def __init__(self, bits): self.bits = bits
        def ids(self): return [bw_fmt.replace("{}", str(i)) for i in range(4)]

    def rows_for(ids):
        out = []
        for kind in range(6):
            if kind < 3:
                tr, pp = [rng.randrange(R.P)], {i: rng.randrange(R.P) for i in ids}
            elif kind == 3:
                tr, pp = [rng.randrange(1 << 8)], {i: rng.randrange(1 << 8) for i in ids}
            elif kind == 4:
                tr, pp = [1], {i: rng.randrange(1 << 16) for i in ids}
            else:
                tr, pp = [0], {i: 0 for i in ids}
            out.append((tr, pp))
        return out

    def interpret(body, env_extra, relation_name, ids):
        stmts = R.split_statements(body)
        rows = []
        for tr, pp in rows_for(ids):
            env = R.base_env({**consts, **bw_consts})
            env.update(env_extra)

            class _S:
                relation = relation_name
            env["self"] = _S()
            rec = PreRecorder(tr, pp)
            env["eval"] = rec
            run_block(stmts, env)
            assert rec.k == 1 and rec.read_ids == ids, (rec.k, rec.read_ids, ids)
            rows.append({"trace": tr, "preproc": pp, "constraints": rec.constraints, "relations": rec.relations, "finalize": rec.finalized})
        return rows

    out_path = os.path.join(R.ROOT, "tests", "golden", "air_eval_vectors.json")
    doc = json.load(open(out_path))
    for bits, _, rel_type in invocations:
        body = rc_body.replace(f"[<LOG_SIZE_RC_ $bit_size>]", f"LOG_SIZE_RC_{bits}")
        rel_name = re.sub(r"(?<=[a-z])(?=[A-Z])|(?<=[a-z])(?=\d)", "_", rel_type).lower()        # RangeCheck8 -> range_check_8
        ids = [RangeCheckCol(int(bits)).id()]
        rows = interpret(body, {"RangeCheck__new": RangeCheckCol}, rel_name, ids)
        name = f"RangeCheck{bits}C"
        doc["components"][name] = {"n_trace": 1, "preproc_ids": ids,
                                   "source": "crates/prover/src/preprocessed/range_check/range_check_macro.rs (define_range_check!, range_check/mod.rs)",
                                   "rows": rows}
        print(f"{name:24s} 1 trace col, {len(rows[0]['constraints'])} constraints, {len(rows[0]['relations'])} relation entries, relation {rel_name}, finalize {rows[0]['finalize']}")
    ids = BitwiseCols(bw_consts["BITWISE_OPERAND_BITS"]).ids()
    rows = interpret(bw_body, {"Bitwise__new": BitwiseCols}, bw_rel.lower(), ids)
    doc["components"]["BitwiseC"] = {"n_trace": 1, "preproc_ids": ids, "source": "crates/prover/src/preprocessed/bitwise.rs", "rows": rows}
    print(f"{'BitwiseC':24s} 1 trace col, {len(rows[0]['constraints'])} constraints, {len(rows[0]['relations'])} relation entries, relation {bw_rel.lower()}, finalize {rows[0]['finalize']}")
    with open(out_path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote", out_path, os.path.getsize(out_path), "bytes;", len(doc["components"]), "components")


if __name__ == "__main__":
    main()
