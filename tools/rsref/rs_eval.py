#!/usr/bin/env python3
"""Golden vectors for the AIR `evaluate` functions, derived MECHANICALLY from the reference's Rust source text.

No Rust toolchain exists in the build image, so the reference cannot be run.  Its `FrameworkEval::evaluate` bodies
(crates/prover/src/components/**.rs) are, however, straight-line programs over field elements: `let` bindings,
`eval.next_trace_mask()`, `eval.add_constraint(expr)`, `eval.add_to_relation(RelationEntry::new(&rel, mult, &[..]))`,
`eval.add_intermediate(expr)` and one `for x in &[..] { .. }` loop.  This script reads those function bodies from
/root/reference AT GENERATION TIME, rewrites each statement token-wise into the equivalent Python statement
(`a.clone()` -> `a`, `E::F::from(x)` -> `E__F__from(x)`, ...), and executes it over M31 values with a recording
`eval` object.  For seeded random rows it writes

    tests/golden/air_eval_vectors.json   { component: { "rows": [ {"trace": [...], "constraints": [...],
                                            "relations": [[relation, multiplicity, [values...]], ...]} ] } }

i.e. inputs and expected outputs only (no reference text).  tests/test_air_eval_golden.py replays the rows through the
oracle's evaluator of the AIR descriptions in cairo_m_amd/csrc/air/*.hpp: every constraint value, every relation entry
(relation, multiplicity, tuple) and their order must match.  That pins the `eval` half of all interpreted components to
the reference text itself rather than to a second hand transcription.

Usage (in the build container, where /root/reference exists):  python tools/rsref/rs_eval.py
"""
import json
import os
import random
import re
import sys

P = 2**31 - 1
REF = "/root/reference/crates"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Felt:
    """M31 element (also stands in for E::F / E::EF: row evaluation over the base field)."""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v % P

    @staticmethod
    def of(x):
        return x if isinstance(x, Felt) else Felt(int(x))

    def __add__(self, o): return Felt(self.v + Felt.of(o).v)
    def __sub__(self, o): return Felt(self.v - Felt.of(o).v)
    def __mul__(self, o): return Felt(self.v * Felt.of(o).v)
    def __neg__(self): return Felt(-self.v)
    __radd__ = __add__
    __rmul__ = __mul__
    def __rsub__(self, o): return Felt(Felt.of(o).v - self.v)
    def inverse(self): return Felt(pow(self.v, P - 2, P))
    def __repr__(self): return f"Felt({self.v})"


def camel_to_const(name):
    return re.sub(r"(?<=[a-z0-9])(?=[A-Z])", "_", name).upper()


def opcode_constants():
    """`instructions! { Variant = N { .. }; }` -> VARIANT_SNAKE_UPPER = N (crates/common/src/instruction.rs:114)."""
    s = open(f"{REF}/common/src/instruction.rs").read()
    return {camel_to_const(m.group(1)): int(m.group(2)) for m in re.finditer(r"^\s{4}([A-Z][A-Za-z0-9]*) = (\d+) \{", s, re.M)}


def strip_comments(s):
    return re.sub(r"//[^\n]*", "", s)


def extract_fn_body(src, header_re):
    m = re.search(header_re, src)
    if not m:
        return None
    i = src.index("{", m.end() - 1)
    depth, j = 0, i
    while True:
        c = src[j]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                break
        j += 1
    return src[i + 1:j]


def file_constants(src, env):
    """`const NAME: ty = expr;` at file scope (integer expressions only)."""
    out = {}
    for m in re.finditer(r"^(?:pub(?:\([a-z]+\))? )?const ([A-Z0-9_]+): (?:u32|usize|u64|i32) = ([^;]+);", src, re.M):
        expr = rust_expr_to_py(m.group(2)).replace("/", "//")      # integer constants: Rust `/` on u32 truncates
        try:
            out[m.group(1)] = eval(expr, {"__builtins__": {}}, {**env, **out})
        except Exception:
            pass
    return out


def rust_expr_to_py(e):
    e = re.sub(r"\s+", " ", e.strip())
    e = re.sub(r"\.clone\(\)", "", e)
    e = re.sub(r"(?<![A-Za-z0-9_])\d[\d_]*", lambda m: m.group(0).replace("_", ""), e)   # 1_000 -> 1000
    e = re.sub(r"\b(\d+)(?:u8|u16|u32|u64|usize|i32|i64)\b", r"\1", e)      # 2u32 -> 2
    e = re.sub(r"\bas (?:u32|usize|u64|i32)\b", "", e)
    e = re.sub(r"(?<![A-Za-z0-9_\)\]]) ?&(?=[A-Za-z_\[\(])", "", e)         # unary & (reference)
    e = e.replace("::", "__")
    return e


def split_statements(body):
    """Top-level statements of a block: `...;` or `for .. { .. }` (returned as ('for', head, inner))."""
    out, i, n = [], 0, len(body)
    while i < n:
        while i < n and body[i].isspace():
            i += 1
        if i >= n:
            break
        if body.startswith("for ", i):
            j = body.index("{", i)
            # the loop head may contain brackets ( &[ .. ] ): the block brace is the first `{` at depth 0
            depth, j = 0, i
            while not (body[j] == "{" and depth == 0):
                depth += body[j] in "([" 
                depth -= body[j] in ")]"
                j += 1
            head = body[i:j]
            depth, k = 0, j
            while True:
                depth += body[k] == "{"
                depth -= body[k] == "}"
                if depth == 0:
                    break
                k += 1
            out.append(("for", head, body[j + 1:k]))
            i = k + 1
            continue
        depth, j = 0, i
        while j < n and not (body[j] == ";" and depth == 0):
            depth += body[j] in "([{"
            depth -= body[j] in ")]}"
            j += 1
        out.append(("stmt", body[i:j].strip()))
        i = j + 1
    return out


class Recorder:
    """EvalAtRow stand-in: hands out the row's trace values, records constraints and relation entries in order."""

    def __init__(self, trace):
        self.trace, self.k = trace, 0
        self.constraints, self.relations, self.finalized = [], [], None

    def next_trace_mask(self):
        v = self.trace[self.k]
        self.k += 1
        return Felt(v)

    def add_constraint(self, x):
        self.constraints.append(Felt.of(x).v)

    def add_intermediate(self, x):
        return Felt.of(x)

    def add_to_relation(self, entry):
        self.relations.append(entry)

    def finalize_logup_in_pairs(self):
        self.finalized = "pairs"

    def finalize_logup(self):
        self.finalized = "single"


class _Rel:
    def __getattr__(self, name):
        return name


class _Self:
    relations = _Rel()
    relation = "self"


def base_env(consts):
    env = dict(consts)
    env.update({
        "E__F__from": Felt.of, "E__EF__from": Felt.of, "M31__from": Felt.of, "BaseField__from": Felt.of,
        "M31__from_u32_unchecked": Felt.of, "E__F__one": lambda: Felt(1), "E__EF__one": lambda: Felt(1),
        "E__F__zero": lambda: Felt(0), "E__EF__zero": lambda: Felt(0),
        "M31__one": lambda: Felt(1), "M31__zero": lambda: Felt(0), "M31__inverse": lambda x: Felt.of(x).inverse(),
        "RelationEntry__new": lambda rel, mult, vals: [rel, Felt.of(mult).v, [Felt.of(v).v for v in vals]],
        "self": _Self(),
    })
    return env


def run_block(stmts, env):
    for st in stmts:
        if st[0] == "for":
            m = re.match(r"for (\w+) in (.*)$", re.sub(r"\s+", " ", st[1].strip()))
            var, it = m.group(1), eval(rust_expr_to_py(m.group(2)), {"__builtins__": {}}, env)
            inner = split_statements(st[2])
            for x in it:
                env[var] = x
                run_block(inner, env)
            continue
        s = re.sub(r"\s+", " ", st[1]).strip()
        if not s or s == "eval":
            continue
        m = re.match(r"let (?:mut )?(\w+)(?: ?: ?[^=]+)? = (.*)$", s, re.S)
        if m:
            env[m.group(1)] = eval(rust_expr_to_py(m.group(2)), {"__builtins__": {}}, env)
        else:
            eval(rust_expr_to_py(s), {"__builtins__": {}}, env)


def count_trace_masks(body):
    return len(re.findall(r"next_trace_mask\(\)", body))


def interpret_evaluate(path, trace_rows, consts):
    src = strip_comments(open(path).read())
    body = extract_fn_body(src, r"fn evaluate<E: EvalAtRow>\(&self, mut eval: E\) -> E \{")
    env0 = base_env({**consts, **file_constants(src, consts)})
    stmts = split_statements(body)
    n_trace = count_trace_masks(body)
    rows = []
    for tr in trace_rows(n_trace):
        env = dict(env0)
        rec = Recorder(tr)
        env["eval"] = rec
        run_block(stmts, env)
        assert rec.k == n_trace, (path, rec.k, n_trace)
        rows.append({"trace": tr, "constraints": rec.constraints, "relations": rec.relations, "finalize": rec.finalized})
    return n_trace, rows


# component name (= air::component_name in cairo_m_amd/csrc/air/components.hpp) -> reference file
COMPONENTS = {
    "AssertEqFpImm": "opcodes/assert_eq_fp_imm.rs", "CallAbsImm": "opcodes/call_abs_imm.rs", "JmpImm": "opcodes/jmp_imm.rs",
    "JnzFpImm": "opcodes/jnz_fp_imm.rs", "Ret": "opcodes/ret.rs", "StoreImm": "opcodes/store_imm.rs",
    "StoreFpFp": "opcodes/store_fp_fp.rs", "StoreFpImm": "opcodes/store_fp_imm.rs",
    "DoubleDerefFpImm": "opcodes/double_deref_fp_imm.rs", "DoubleDerefFpFp": "opcodes/double_deref_fp_fp.rs",
    "StoreFramePointer": "opcodes/store_frame_pointer.rs", "U32StoreImm": "opcodes/u32_store_imm.rs",
    "U32StoreAddFpImm": "opcodes/u32_store_add_fp_imm.rs", "U32StoreMulFpImm": "opcodes/u32_store_mul_fp_imm.rs",
    "U32StoreDivFpImm": "opcodes/u32_store_div_fp_imm.rs", "U32StoreEqFpFp": "opcodes/u32_store_eq_fp_fp.rs",
    "U32StoreEqFpImm": "opcodes/u32_store_eq_fp_imm.rs", "U32StoreLtFpImm": "opcodes/u32_store_lt_fp_imm.rs",
    "U32StoreLtFpFp": "opcodes/u32_store_lt_fp_fp.rs", "U32StoreAddFpFp": "opcodes/u32_store_add_fp_fp.rs",
    "U32StoreSubFpFp": "opcodes/u32_store_sub_fp_fp.rs", "U32StoreMulFpFp": "opcodes/u32_store_mul_fp_fp.rs",
    "U32StoreDivFpFp": "opcodes/u32_store_div_fp_fp.rs", "U32StoreBitwiseFpFp": "opcodes/u32_store_bitwise_fp_fp.rs",
    "U32StoreBitwiseFpImm": "opcodes/u32_store_bitwise_fp_imm.rs", "StoreLeFpImm": "opcodes/store_le_fp_imm.rs",
    "MemoryC": "memory.rs", "MerkleC": "merkle.rs", "ClockUpdateC": "clock_update.rs",
}


def main():
    consts = opcode_constants()
    consts["P"] = P                      # stwo_prover::core::fields::m31::P
    # `define_range_check!(N, ..)` expands to `pub const LOG_SIZE_RC_N: u32 = N` (range_check_macro.rs:33, range_check/mod.rs:35-43)
    rc = open(f"{REF}/prover/src/preprocessed/range_check/mod.rs").read()
    for m in re.finditer(r"define_range_check!\((\d+),", rc):
        consts[f"LOG_SIZE_RC_{m.group(1)}"] = int(m.group(1))
    # constants some components import from other modules
    for rel, names in (("prover/src/adapter/merkle.rs", ["TREE_HEIGHT"]), ("prover/src/adapter/memory.rs", ["RC20_LIMIT"])):
        s = strip_comments(open(f"{REF}/{rel}").read())
        fc = file_constants(s, consts)
        for nme in names:
            if nme in fc:
                consts[nme] = fc[nme]
    rng = random.Random(0xA1E)

    def rows_for(n):
        out = [[rng.randrange(P) for _ in range(n)] for _ in range(3)]       # arbitrary field elements
        out.append([rng.randrange(1 << 8) for _ in range(n)])               # small values (limb-sized)
        out.append([1] + [rng.randrange(1 << 16) for _ in range(n - 1)])    # enabler = 1
        out.append([0] * n)                                                 # all-zero row
        return out

    doc = {"_about": "generated by tools/rsref/rs_eval.py from the reference's evaluate() bodies; inputs and expected outputs only",
           "components": {}}
    for name, rel in COMPONENTS.items():
        n, rows = interpret_evaluate(f"{REF}/prover/src/components/{rel}", rows_for, consts)
        doc["components"][name] = {"n_trace": n, "source": f"crates/prover/src/components/{rel}", "rows": rows}
        print(f"{name:24s} {n:3d} trace cols, {len(rows[0]['constraints']):3d} constraints, {len(rows[0]['relations']):3d} relation entries")
    out = os.path.join(ROOT, "tests", "golden", "air_eval_vectors.json")
    with open(out, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
