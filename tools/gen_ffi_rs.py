#!/usr/bin/env python3
"""Development aid: prints the `unsafe extern "C" { ... }` block of integration/prover-hip/src/ffi.rs from include/cairom_hip.h
(every exported function, argument by argument).  The output is pasted into ffi.rs by hand; tests/test_rust_shim.py keeps the two
in sync (every header function must have an extern twin of the same arity)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "size_t": "usize", "double": "f64",
        "cm_handle": "cm_handle", "cm_stream_t": "cm_stream_t", "char": "c_char", "void": "c_void"}


def rust_type(ctype, name):
    """ctype: e.g. 'const cm_handle*', 'uint32_t', 'cm_proof**'; name may carry array suffixes: 'alpha[4]', 'roots[4][32]'"""
    const = ctype.startswith("const ")
    t = ctype.replace("const ", "").strip()
    stars = t.count("*")
    t = t.replace("*", "").replace(" const", "").strip()
    base = BASE.get(t, t)
    dims = re.findall(r"\[(\w+)\]", name)
    if dims:                                   # array parameter = pointer to its element (outer dimension decays)
        inner = base
        for d in reversed(dims[1:]):
            inner = f"[{inner}; {d}]"
        base = inner
        stars += 1
    out = base
    for k in range(stars):
        # the innermost pointer level carries the const of the pointee; outer levels of `T**` out-parameters are mutable
        is_inner = k == 0
        out = ("*const " if (const and is_inner) else "*mut ") + out
    return out


def functions(hdr):
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"typedef struct [^{;]*\{.*?\}\s*\w+;", "", hdr, flags=re.S)
    for m in re.finditer(r"^\s*((?:const )?\w+\*?)\s+(cm_\w+)\((.*?)\);", hdr, flags=re.S | re.M):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in re.split(r",(?![^\[]*\])", args):
                a = a.strip()
                fp = re.match(r"(\w+)\s*\(\*(\w+)\)\((.*)\)", a)
                assert not fp, a
                mm = re.match(r"(.*?)(\w+(?:\[\w+\])*)$", a)
                ctype, pname = mm.group(1).strip(), mm.group(2)
                params.append((re.sub(r"\[.*", "", pname), rust_type(ctype, pname)))
        yield name, ret, params


def main():
    hdr = open(os.path.join(ROOT, "include", "cairom_hip.h")).read()
    print('unsafe extern "C" {')
    for name, ret, params in functions(hdr):
        r = rust_type(ret, "")
        ps = ", ".join(f"{n}: {t}" for n, t in params)
        print(f"    pub fn {name}({ps}) -> {r};")
    print("}")


if __name__ == "__main__":
    main()
