"""The Rust shim (integration/prover-hip, shipped as source: no Rust toolchain in the build image) must stay in sync
with the C ABI: every `#[repr(C)]` struct of src/ffi.rs has the fields of its C twin in include/cairom_hip.h, in order,
with matching widths; every extern fn it declares is exported by the library with the same arity; its opcode groups equal
the library's component table."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FFI = open(os.path.join(ROOT, "integration", "prover-hip", "src", "ffi.rs")).read()
LIB = open(os.path.join(ROOT, "integration", "prover-hip", "src", "lib.rs")).read()
HDR = open(os.path.join(ROOT, "include", "cairom_hip.h")).read()


def rust_structs():
    out = {}
    for m in re.finditer(r"pub struct (\w+) \{(.*?)\n\}", FFI, re.S):
        out[m.group(1)] = [(n, re.sub(r"\s+", "", t)) for n, t in re.findall(r"pub (\w+): ([^,\n]+),", m.group(2))]
    return out


def c_structs():
    out = {}
    hdr = re.sub(r"/\*.*?\*/", "", HDR, flags=re.S)
    for m in re.finditer(r"typedef struct \{(.*?)\} (\w+);", hdr, re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ty, names = re.match(r"((?:const )?\w+\*?)\s+(.*)", decl, re.S).groups()
            for nm in names.split(","):
                nm = nm.strip()
                a = re.match(r"(\*?)(\w+)(?:\[(\w+)\])?", nm)
                fields.append((a.group(2), ty + a.group(1), a.group(3)))
        out[m.group(2)] = fields
    return out


def test_repr_c_structs_match_the_header():
    rs, cs = rust_structs(), c_structs()
    for name in ("cm_bundle", "cm_data_access", "cm_memory_cell", "cm_clock_update", "cm_merkle_node", "cm_prover_input", "cm_pcs_config",
                 "cm_sample_batches", "cm_runner_segment"):
        assert name in rs and name in cs, name
        assert [f[0] for f in rs[name]] == [f[0] for f in cs[name]], name
        for (rn, rt), (cn, ct, arr) in zip(rs[name], cs[name]):
            if ct.endswith("*"):
                assert rt.startswith("*const") or rt.startswith("[*const"), (name, rn, rt, ct)
            else:
                want = {"uint32_t": "u32", "uint64_t": "u64"}[ct]
                assert rt == want or rt.startswith(f"[{want};"), (name, rn, rt, ct)
            if arr:
                n = {"CM_N_OPCODE_COMPONENTS": "CM_N_OPCODE_COMPONENTS"}.get(arr, arr)
                assert rt.endswith(f";{n}]"), (name, rn, rt, arr)


def header_functions():
    """name -> argument count of every function include/cairom_hip.h declares"""
    hdr = re.sub(r"/\*.*?\*/", "", HDR, flags=re.S)
    hdr = re.sub(r"typedef struct [^{;]*\{.*?\}\s*\w+;", "", hdr, flags=re.S)
    out = {}
    for m in re.finditer(r"^\s*(?:const )?\w+\*?\s+(cm_\w+)\((.*?)\);", hdr, flags=re.S | re.M):
        args = " ".join(m.group(2).split())
        out[m.group(1)] = 0 if args in ("", "void") else len(re.split(r",(?![^\[]*\])", args))
    return out


def test_every_header_function_has_an_extern_twin_and_is_exported():
    """ffi.rs covers the WHOLE C ABI (the HipBackend trait impls of src/backend.rs bind the per-op entry points):
    every function of the header is declared in the extern block with the same number of arguments and is exported by the
    library; the extern block declares nothing the header does not."""
    L = C.CDLL(os.path.join(ROOT, "cairo_m_amd", "libcairom_hip.so"))
    block = FFI[FFI.index('unsafe extern "C"'):]
    fns = {name: args for name, args in re.findall(r"pub fn (\w+)\((.*?)\) -> [^;]+;", block, re.S)}
    hdr = header_functions()
    assert len(hdr) > 70
    assert sorted(set(hdr) - set(fns)) == [], "header functions without an extern twin in ffi.rs"
    assert sorted(set(fns) - set(hdr)) == [], "extern declarations the header does not have"
    for name, n_args in hdr.items():
        getattr(L, name)
        got = len([a for a in fns[name].split(",") if a.strip()])
        assert got == n_args, (name, got, n_args)


def test_opcode_groups_equal_the_component_table():
    """OPCODE_GROUPS of lib.rs (macro order of define_opcodes!) against the library: vm programs of one opcode land in the
    component with the group's index (tests/golden/proof_schema.json holds the reference's module order)."""
    import json
    schema = json.load(open(os.path.join(ROOT, "tests", "golden", "proof_schema.json")))
    groups = re.search(r"const OPCODE_GROUPS.*?\[\s*(&\[.*?)\n    \]\n\};", LIB, re.S).group(1)
    rows = re.findall(r"&\[([A-Z0-9_, ]+)\]", groups)
    assert len(rows) == 26 == len(schema["opcodes"])
    snake = lambda s: s.lower()
    for row, module in zip(rows, schema["opcodes"]):
        first = snake(row.split(",")[0].strip())
        # the module name is the opcode name or its family (store_add_fp_fp -> store_fp_fp, jmp_abs_imm -> jmp_imm, ...)
        stem = re.sub(r"_(add|sub|mul|div|abs|rel|and|or|xor|rem|to)(?=_|$)", "", first).replace("store_double_deref_fp_fp", "double_deref_fp_fp")
        stem = {"store_double_deref_fp": "double_deref_fp_imm", "u32_store_fp_fp": "u32_store_bitwise_fp_fp" if "AND" in row else "u32_store_fp_fp",
                "u32_store_fp_imm": "u32_store_bitwise_fp_imm" if "AND" in row else "u32_store_fp_imm"}.get(stem, stem)
        assert stem == module or module in (first, stem) or first.replace("_rem", "") == module or \
            re.sub(r"_(add|sub|mul|div|lt|eq)_", "_\\1_", first) == module, (row, module, stem)


# ---- syntax gate (tools/rs_syntax_gate.py): the only compiler substitute in an image without rustc ---------------------------
def _gate():
    import importlib.util
    spec = importlib.util.spec_from_file_location("rs_syntax_gate", os.path.join(ROOT, "tools", "rs_syntax_gate.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _shim_sources():
    base = os.path.join(ROOT, "integration", "prover-hip")
    out = []
    for sub in ("src", "tests"):
        d = os.path.join(base, sub)
        out += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(".rs")]
    out.append(os.path.join(base, "build.rs"))
    return [p for p in out if os.path.exists(p)]


def test_shim_sources_pass_the_syntax_gate():
    g = _gate()
    files = _shim_sources()
    assert len(files) >= 5
    for p in files:
        assert g.check_file(p) > 50, p


def test_syntax_gate_rejects_what_it_is_there_for(tmp_path):
    """The gate is not vacuous: each of these edits of a shim source — the slips a never-compiled crate accumulates — is refused."""
    g = _gate()
    src = open(os.path.join(ROOT, "integration", "prover-hip", "src", "lib.rs")).read()
    edits = {
        "dropped brace": lambda s: s[::-1].replace("}", "", 1)[::-1],
        "dropped paren": lambda s: s.replace("ensure_init();", "ensure_init);", 1),
        "unterminated string": lambda s: s.replace('"', "", 1),
        "let without semicolon": lambda s: re.sub(r"(let rc = unsafe \{[^\n]*\});", r"\1", s, count=1),
        "dangling doc comment": lambda s: s.rstrip() + "\n/// a comment that documents nothing\n",
        "stray keyword in item position": lambda s: s.replace("\npub fn ", "\npub return fn ", 1),
        "field without a type": lambda s: s.replace("pub struct ProofHandleRef(pub *const cm_proof);", "pub struct ProofHandleRef { handle, }"),
        "crossed delimiters": lambda s: s.replace("{", "(", 1),
    }
    for name, f in edits.items():
        bad = f(src)
        assert bad != src, name
        p = tmp_path / "m.rs"
        p.write_text(bad)
        with pytest.raises(g.GateError):
            g.check_file(str(p))
    p = tmp_path / "ok.rs"
    p.write_text(src)
    assert g.check_file(str(p)) > 0


def test_syntax_gate_accepts_compiler_accepted_code():
    """Calibration against code a compiler HAS seen: every .rs file of the reference's prover / runner / common crates passes
    (build container only: /root/reference does not travel)."""
    ref = "/root/reference/crates"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present")
    g = _gate()
    n = 0
    for crate in ("prover", "runner", "common"):
        for d, _, fs in os.walk(os.path.join(ref, crate)):
            for f in fs:
                if f.endswith(".rs"):
                    g.check_file(os.path.join(d, f))
                    n += 1
    assert n > 60
