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
A snapshot is left out when the evaluator does not cover its source (arrays, structs, tuples, pointers, casts), nothing else is
filtered: in particular NOT on what this repository's VM returns.  `provable` is false when the program executes U32StoreEqFpFp /
U32StoreEqFpImm (opcodes 24 / 30): the reference's AIR for those two cannot balance its own LogUp sum on a live row
(u32_store_eq_fp_fp.rs:210, u32_store_eq_fp_imm.rs:250-251), so such programs are VM-checked only."""
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cm_eval import Interp, Unsupported, to_words  # noqa: E402

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
    ins, labels, addr = [], {}, 0
    for line in listing.splitlines():
        lm = re.match(r"^(\w+):\s*$", line)
        if lm:
            labels.setdefault(lm.group(1), addr)
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
    return source, ins, labels


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
            else:
                vals.append(("bool", (k + i) % 2 == 0))
        sets.append(vals)
    return sets


def main():
    os.makedirs(OUT, exist_ok=True)
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
            source, ins, labels = parsed
            it = Interp(source)
            entry = it.order[0]
            if entry not in labels:
                skipped.append((name, f"entry function {entry} has no label"))
                continue
            params, ret, _ = it.fns[entry]
            if ret is None:
                skipped.append((name, "entry function returns nothing"))
                continue
            cases = []
            for vals in arg_sets(params):
                try:
                    r = Interp(source).call(entry, list(vals))
                except Unsupported as e:
                    if "division by zero" in str(e):
                        continue
                    raise
                cases.append({"args": [w for v in vals for w in to_words(v)], "expected": to_words(r)})
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
        fx = {"name": name, "snapshot": os.path.relpath(path, "/root/reference"), "entry": entry, "entry_pc": labels[entry],
              "n_returns": len(cases[0]["expected"]), "instructions": ins, "cases": cases, "opcodes": opcodes,
              "provable": not ({24, 30} & set(opcodes)),
              "made_by": "tools/casm/make_casm_fixtures.py (listing parsed from the snapshot; expected values from tools/casm/cm_eval.py on the snapshot's source)"}
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(fx, f, separators=(",", ":"))
            f.write("\n")
        made.append(name)
    print(f"{len(made)} fixtures, {len(skipped)} snapshots left out")
    for n, why in skipped:
        print(f"  left out  {n}: {why}")


if __name__ == "__main__":
    main()
