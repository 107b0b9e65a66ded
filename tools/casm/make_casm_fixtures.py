#!/usr/bin/env python3
"""Programs the reference's COMPILER emitted, as prover workloads (build container only; run from the repo root):

    python tools/casm/make_casm_fixtures.py            -> tests/golden/casm/*.json

Every `crates/compiler/codegen/tests/snapshots/*.snap` of the reference holds a Cairo-M source and the CASM listing the compiler
generated for it (insta snapshot of crates/compiler/codegen/tests/mdtest_snapshots.rs).  For each snapshot this script
  * parses the LISTING into instruction words (`idx: opcode operands.. // comment`; `_` = no operand; jump / call operands are
    already memory addresses: an instruction of more than four words occupies two cells, crates/common/src/instruction.rs:314-577),
  * evaluates the SOURCE with tools/casm/cm_eval.py — on fixed argument sets when the entry function takes parameters — to get
    the value the entry function returns, independently of the listing,
and writes {instructions, entry_pc, cases: [{args, expected}], n_returns, opcodes, provable} — data only, no source text.
Where the reference's mdtest markdown pairs the program with a RUST EQUIVALENT written by its authors (39 of them; the reference's
runner compiles both and compares the outputs), tools/casm/rust_eval.py evaluates that Rust block on the same arguments and the
script stops on any disagreement with cm_eval: `rust_equivalent_cases` = the number of argument sets that agreed.
`data` = the memory cells the listing places behind the instructions (constant arrays, the heap cursor).  A snapshot is left out
when it holds no listing (expected compile errors) or the evaluator does not cover its source, nothing else is filtered: in
particular NOT on what this repository's VM returns.  `provable` is false when the program executes U32StoreEqFpFp /
U32StoreEqFpImm (opcodes 24 / 30): the reference's AIR for those two cannot balance its own LogUp sum on a live row
(u32_store_eq_fp_fp.rs:210, u32_store_eq_fp_imm.rs:250-251) — or when an instruction of the listing names one frame cell in two operands
(the heap allocator's `[fp + 12] = [fp + 12] + (-1)`): the second access has prev_clock == clock and the AIR's range_check_20 lookup of
`clock - prev_clock - 1` has no entry for -1.  Such programs are VM-checked, and the tests require the constraint check to REJECT them."""
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cm_eval import Fault, Interp, Ref, Unsupported, to_words  # noqa: E402
import rust_eval  # noqa: E402

SNAPS = "/root/reference/crates/compiler/codegen/tests/snapshots"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "casm")
FELT_ARGS = [[0], [5], [15], [7], [1], [2147483646]]
U32_ARGS = [0xDEADBEEF, 0x0F0F1234, 69, 420, 0, 0xFFFFFFFF, 65536, 65535]


def parse_snapshot(path):
    text = open(path).read()
    m = re.search(r"\nSource:\n(.*?)\n=+\nGenerated CASM:\n(.*)\Z", text, re.S)
    if not m:
        return None
    source, listing = m.group(1), m.group(2)
    ins, labels, addr, data = [], {}, 0, []
    in_data = False
    for line in listing.splitlines():
        lm = re.match(r"^(\w+):\s*$", line)
        if lm:
            labels.setdefault(lm.group(1), addr)
            continue
        sm = re.match(r"^---- (rodata|data) \(base (\d+)\) ----$", line)
        if sm:   # memory cells behind the instructions: constant arrays (rodata), the heap cursor (data); four words per cell
            if int(sm.group(2)) != addr + len(data):
                raise ValueError(f"{path}: {sm.group(1)} section at {sm.group(2)}, expected {addr + len(data)}")
            in_data = True
            continue
        if in_data:
            if line.startswith(";") or not line.strip():
                continue
            dm = re.match(r"^\s*(\d+):\s+(\d+) (\d+) (\d+) (\d+)\s*$", line)
            if not dm or int(dm.group(1)) != addr + len(data):
                raise ValueError(f"{path}: cannot read data line {line!r}")
            data.append([int(dm.group(k)) for k in range(2, 6)])
            continue
        im = re.match(r"^\s*(\d+):\s+(.*?)\s*(//.*)?$", line)
        if not im:
            if line.strip():
                raise ValueError(f"{path}: cannot read listing line {line!r}")
            continue
        if int(im.group(1)) != len(ins):
            raise ValueError(f"{path}: instruction index {im.group(1)} out of sequence")
        words = [w for w in im.group(2).split() if w != "_"]
        if not all(re.fullmatch(r"\d+", w) for w in words):
            raise ValueError(f"{path}: non-numeric operand in {line!r}")
        ins.append([int(w) for w in words])
        addr += (len(words) + 3) // 4
    return source, ins, labels, data


P31 = 2**31 - 1
# the frame cells an instruction touches, per operand (index into the instruction words, width in cells): what the opcodes' comments
# in crates/common/src/instruction.rs:316-577 say (`[fp + dst_off] = [fp + src0_off] + [fp + src1_off]`, u32 values = two cells)
FRAME_OPERANDS = {
    0: [(1, 1), (2, 1), (3, 1)], 1: [(1, 1), (2, 1), (3, 1)], 2: [(1, 1), (2, 1), (3, 1)], 3: [(1, 1), (2, 1), (3, 1)],
    4: [(1, 1), (3, 1)], 6: [(1, 1), (3, 1)], 48: [(1, 1), (3, 1)], 50: [(1, 1)],
    8: [(1, 1), (3, 1)], 42: [(1, 1), (2, 1), (3, 1)], 9: [(2, 1)], 43: [(2, 1)], 14: [(1, 1)],
    15: [(1, 2), (2, 2), (3, 2)], 16: [(1, 2), (2, 2), (3, 2)], 17: [(1, 2), (2, 2), (3, 2)], 18: [(1, 2), (2, 2), (3, 2), (4, 2)],
    36: [(1, 2), (2, 2), (3, 2)], 37: [(1, 2), (2, 2), (3, 2)], 38: [(1, 2), (2, 2), (3, 2)],
    19: [(1, 2), (4, 2)], 21: [(1, 2), (4, 2)], 39: [(1, 2), (4, 2)], 40: [(1, 2), (4, 2)], 41: [(1, 2), (4, 2)], 22: [(1, 2), (4, 2), (5, 2)],
    23: [(3, 2)], 24: [(1, 2), (2, 2), (3, 1)], 28: [(1, 2), (2, 2), (3, 1)], 30: [(1, 2), (4, 1)], 34: [(1, 2), (4, 1)],
    44: [(1, 1), (3, 1)], 45: [(1, 1), (2, 1), (3, 1)],
}


def touches_a_cell_twice(ins):
    """-> index of the first instruction that names one frame cell in two of its operands (`[fp + 12] = [fp + 12] + (-1)`), or None.
    Memory::push gives the second access of a step prev_clock == clock (crates/prover/src/adapter/memory.rs:470-535), and every
    opcode AIR looks `clock - prev_clock - enabler` up in range_check_20 (e.g. opcodes/store_fp_imm.rs): -1 is not in the table, so
    the REFERENCE cannot prove such a step either"""
    for k, w in enumerate(ins):
        seen = set()
        for idx, width in FRAME_OPERANDS.get(w[0], []):
            cells = {(w[idx] + d) % P31 for d in range(width)}
            if cells & seen:
                return k
            seen |= cells
    return None


def ins_cells(ins):
    """one cell per four words of an instruction"""
    return [None for w in ins for _ in range((len(w) + 3) // 4)]


def arg_sets(params):
    if not params:
        return [[]]
    sets = []
    for k in range(4):
        vals = []
        for i, (_, ty) in enumerate(params):
            if ty == "felt":
                vals.append(("felt", FELT_ARGS[(k + i) % len(FELT_ARGS)][0]))
            elif ty == "u32":
                vals.append(("u32", U32_ARGS[(2 * k + i) % len(U32_ARGS)]))
            elif ty == "bool":
                vals.append(("bool", (k + i) % 2 == 0))
            else:   # [felt; n] / [u32; n]
                el, n = ty[1], ty[2]
                items = [("felt", FELT_ARGS[(k + i + j) % len(FELT_ARGS)][0]) if el == "felt" else ("u32", U32_ARGS[(2 * k + i + j) % len(U32_ARGS)])
                         for j in range(n)]
                vals.append(("array", Ref(items, el, False)))
        sets.append(vals)
    return sets


def call_frame_words(vals, program_length):
    """the words below the entry frame, lowest address first (crates/runner/src/lib.rs:405-441): the data of every array argument,
    materialised from the initial frame pointer (= the program length) upwards, then one slot list per argument — a scalar's words, an
    array's ADDRESS"""
    data, slots = [], []
    for v in vals:
        if v[0] == "array":
            slots.append(program_length + len(data))
            data += [w for x in v[1].items for w in to_words(x)]
        else:
            slots += to_words(v)
    return data + slots


def norm_source(src):
    import textwrap
    return "\n".join(l.rstrip() for l in textwrap.dedent(src).strip().splitlines())


def value_ints(v):
    """the integers of a cm_eval value in order (not ABI limbs): what the reference's runner turns into M31 values to compare"""
    if v is None:
        return []
    t = v[0]
    if t in ("felt", "u32"):
        return [v[1]]
    if t == "bool":
        return [1 if v[1] else 0]
    if t == "tuple":
        return [x for it in v[1] for x in value_ints(it)]
    if t == "struct":
        return [x for it in v[2].values() for x in value_ints(it)]
    raise Unsupported(f"{t} in a compared value")


def rust_equivalent_agrees(cm_source, rs_source, entry, params, sets):
    """-> number of argument sets on which the reference authors' Rust statement of the program (tools/casm/rust_eval.py) and
    cm_eval give the same values as M31 (crates/runner/tests/common/mod.rs:150-165); raises SystemExit on a disagreement"""
    ri = rust_eval.Interp(rs_source)
    if entry not in ri.fns:
        raise rust_eval.Unsupported(f"the Rust block has no fn {entry}")
    n = 0
    for vals in sets:
        try:
            cv = Interp(cm_source).call(entry, list(vals))
        except Fault:
            continue
        rargs = []
        for v, (_, pt) in zip(vals, ri.fns[entry][0]):
            rargs.append(v[1] if v[0] == "bool" else [rust_eval.Int(x[1], pt[1]) for x in v[1].items] if v[0] == "array"
                         else rust_eval.Int(v[1], pt))
        try:
            rv = rust_eval.Interp(rs_source).call(entry, rargs)
        except rust_eval.Fault:
            continue
        a, b = [x % P31 for x in value_ints(cv)], [x % P31 for x in rust_eval.flat_ints(rv)]
        if a != b:
            raise SystemExit(f"{entry}{[(v[0], v[1]) for v in vals]}: cm_eval says {a}, the reference's Rust equivalent says {b}")
        n += 1
    return n


def stated_expectations():
    """{entry function name: value} for the mdtest programs whose markdown states the result itself (`//! expected: V` in
    /root/reference/mdtest/**/*.md, mdtest/README.md:56): the only expected values in the tree that nobody here computed"""
    out = {}
    for md in glob.glob("/root/reference/mdtest/*/*.md"):
        for block in re.findall(r"```cairo-m\n(.*?)```", open(md).read(), re.S):
            m = re.search(r"^//! expected: (\d+)\s*$", block, re.M)
            f = re.search(r"^fn (\w+)\(", block, re.M)
            if m and f:
                out[f.group(1)] = int(m.group(1))
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    stated = stated_expectations()
    rust_of = {norm_source(c): r for _, c, r in rust_eval.mdtest_pairs()}
    n_rust = 0
    for old in glob.glob(os.path.join(OUT, "*.json")):
        os.remove(old)
    made, skipped = [], []
    for path in sorted(glob.glob(os.path.join(SNAPS, "*.snap"))):
        name = re.sub(r"^mdtest_codegen_snapshots@|\.snap$", "", os.path.basename(path))
        try:
            parsed = parse_snapshot(path)
            if not parsed:
                skipped.append((name, "no Source / Generated CASM sections"))
                continue
            source, ins, labels, data = parsed
            it = Interp(source)
            # the reference's own rule (crates/runner/tests/common/mod.rs:185-215): test_main > main > the first function that returns a value
            entry = next((n for n in ("test_main", "main") if n in it.fns), None) or next((n for n in it.order if it.fns[n][1] is not None), it.order[0])
            if entry not in labels:
                skipped.append((name, f"entry function {entry} has no label"))
                continue
            params, ret, _ = it.fns[entry]
            if any(ty not in ("felt", "u32", "bool") and not (ty[0] == "array" and ty[1] in ("felt", "u32")) for _, ty in params):
                skipped.append((name, "entry function takes a struct / tuple / nested array"))
                continue
            cases = []
            for vals in arg_sets(params):
                try:
                    r = Interp(source).call(entry, list(vals))
                except Fault:   # the program itself fails on these arguments: no expected value
                    continue
                cases.append({"args": call_frame_words(vals, len(ins_cells(ins)) + len(data)), "expected": to_words(r)})
            if not cases:
                skipped.append((name, "no argument set evaluates"))
                continue
        except Unsupported as e:
            skipped.append((name, f"source outside the evaluator's subset: {e}"))
            continue
        except ValueError as e:   # a listing with a read-only data section (constant arrays): not an instruction stream alone
            skipped.append((name, f"listing: {str(e).split(': ', 1)[-1]}"))
            continue
        opcodes = sorted({i[0] for i in ins})
        twice = touches_a_cell_twice(ins)
        why_not = None
        if {24, 30} & set(opcodes):
            why_not = "U32StoreEq* (opcodes 24 / 30): the reference's AIR cannot balance its own LogUp sum on a live row"
        elif twice is not None:
            why_not = f"instruction {twice} reads and writes one frame cell in the same step: clock - prev_clock - 1 = -1 is not in range_check_20"
        rust_cases = None
        rs = rust_of.get(norm_source(source))
        if rs is not None:
            try:
                rust_cases = rust_equivalent_agrees(source, rs, entry, params, arg_sets(params))
                n_rust += 1
            except rust_eval.Unsupported as e:
                print(f"  rust equivalent of {name} outside the Rust evaluator's subset: {e}")
        ref_expected = None
        if entry in stated and not params:
            ref_expected = stated[entry]
            if cases[0]["expected"] != [ref_expected % P31]:
                raise SystemExit(f"{name}: the evaluator says {cases[0]['expected']}, the reference's markdown says {ref_expected}")
        fx = {"name": name, "snapshot": os.path.relpath(path, "/root/reference"), "entry": entry, "entry_pc": labels[entry],
              "reference_expected": ref_expected, "rust_equivalent_cases": rust_cases,
              "n_returns": len(cases[0]["expected"]), "instructions": ins, "data": data, "cases": cases, "opcodes": opcodes,
              "provable": why_not is None, "unprovable_reason": why_not,
              "made_by": "tools/casm/make_casm_fixtures.py (listing parsed from the snapshot; expected values from tools/casm/cm_eval.py on the snapshot's source)"}
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(fx, f, separators=(",", ":"))
            f.write("\n")
        made.append(name)
    print(f"{len(made)} fixtures, {len(skipped)} snapshots left out; {n_rust} checked against the reference's Rust equivalents")
    for n, why in skipped:
        print(f"  left out  {n}: {why}")


if __name__ == "__main__":
    main()
