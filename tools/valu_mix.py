#!/usr/bin/env python3
"""Static VALU instruction mix of the kernels behind bench.py's `roofline.valu_model` (round 6).

Runs in the BUILD container (hipcc cross-compiles gfx950 without a GPU): compiles the listed sources of cairo_m_amd/csrc to
assembly (`hipcc -S --cuda-device-only`), counts the VALU instructions of each kernel by ISSUE-COST BUCKET and writes
profiles/<tag>_valu_mix.json.  bench.py multiplies a class's SQ_INSTS_VALU (wave-instructions per proof, profiles/<tag>_pmc_sq.json)
with the class's mean cost here to get the time the SIMDs' VALU ports need at least for that class.

Buckets and their cost per wave-instruction per SIMD come from the single-opcode lab (tools/valu_lab.hip, profiles/r03k_valu_lab.txt:
chip-wide lane-ops/s with 8 waves per SIMD; ns = 64 lanes x 1024 SIMDs / rate):
  full   32-bit VOP1 / VOP2 / VOPC encodings (v_xor_b32_e32, v_add_u32_e32, v_and, v_lshrrev, v_sub, v_mov, v_cndmask_e32 ...)  65.9 T -> 0.99 ns
  half   every 64-bit encoding: VOP3 (v_add3_u32, v_alignbit_b32, v_perm, v_bfe, v_lshl_add, v_mul_lo/hi_u32, *_e64), SDWA, DPP       37.5 T -> 1.75 ns
  mad64  v_mad_u64_u32                                                                                                                31.46 T -> 2.08 ns
(the lab prices an SDWA xor like a VOP3; the Merkle kernels gain 5 % from the SDWA pair all the same — DESIGN_HISTORY round 3 — so the model is
conservative there.)  Loops are counted once (static text): the mean cost per instruction is what is used, not the count.

usage: python tools/valu_mix.py <tag>     e.g. r06a -> profiles/r06a_valu_mix.json"""
import collections, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cairo_m_amd", "csrc")
NS = {"full": 64 * 1024 / 65.9e12 * 1e9, "half": 64 * 1024 / 37.5e12 * 1e9, "mad64": 64 * 1024 / 31.46e12 * 1e9}
# kprof / pmc class -> (source file, regex over the DEMANGLED-ish mangled names of the kernels that make up the class's VALU time)
CLASSES = {
    "k_merkle_layer": ("kernels_hash.hip", r"k_merkle_layerILb0|k_merkle_narrowILb0"),
    "k_merkle_multi": ("kernels_hash.hip", r"k_merkle_multiILb0"),
    "k_merkle_top": ("kernels_hash.hip", r"k_merkle_topILb0"),
    "k_fold_leaf": ("kernels_fri.hip", r"k_fold_leafILb0"),
    "k_fft_pass<fft>": ("kernels_fft.hip", r"k_fft_pass_rbILb0ELi12ELi12ELi4"),
    "k_fft_pass<ifft>": ("kernels_fft.hip", r"k_fft_pass_rbILb1ELi12ELi12ELi4|k_fft_pass_rbILb1ELi9ELi14ELi4|k_fft_pass_rbILb1ELi7ELi14ELi4"),
    "k_fft_fused_rb": ("kernels_fft.hip", r"k_fft_fused_rbILi[6-9]ELi4"),
    "k_constraints(region)": ("kernels_air_3.hip", r"k_constraintsIN3air(10StoreFpImm|9StoreFpFp|8JnzFpImm|6JmpImm|8StoreImm|10Poseidon2C)"),
    "k_logup(region)": ("kernels_air_2.hip", r"k_logupIN3air(10StoreFpImm|9StoreFpFp|8JnzFpImm|6JmpImm|8StoreImm)"),
    "k_quotients": ("kernels_fri.hip", r"k_quotients_rowsILi2"),
    "k_eval_at_point": ("kernels_poly.hip", r"k_eval_partial_multi"),
}


def bucket(op):
    if not op.startswith("v_"):
        return None
    if op.startswith("v_mad_u64_u32") or op.startswith("v_mad_i64_i32"):
        return "mad64"
    if op.endswith("_e32"):
        return "full"
    return "half"      # _e64, _sdwa, _dpp and the VOP3-only opcodes (printed without a suffix)


def asm_of(src):
    out = os.path.join("/tmp", "valu_mix_" + src.replace(".hip", ".s"))
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".inc"))):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-Wno-unused-command-line-argument",
                               os.path.join(CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kernels(lines):
    """name -> Counter(opcode) for every kernel (function bodies between `name:` and .Lfunc_end)"""
    out, cur = {}, None
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1); out[cur] = collections.Counter(); continue
        if l.startswith(".Lfunc_end"):
            cur = None; continue
        if cur is None:
            continue
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        out[cur][t.split()[0]] += 1
    return out


def main():
    tag = sys.argv[1]
    res = {"ns_per_wave_instruction": NS, "source": "static ISA of HEAD (tools/valu_mix.py); costs from profiles/r03k_valu_lab.txt", "classes": {}}
    cache = {}
    tot = collections.Counter()
    for klass, (src, pat) in CLASSES.items():
        if src not in cache:
            cache[src] = kernels(asm_of(src))
        b = collections.Counter(); names = []
        for name, ops in cache[src].items():
            if re.search(pat, name):
                names.append(name)
                for op, n in ops.items():
                    k = bucket(op)
                    if k:
                        b[k] += n
        n = sum(b.values())
        assert n, (klass, "no kernel matched")
        mean = sum(b[k] * NS[k] for k in b) / n
        res["classes"][klass] = {"kernels": len(names), "static_valu": n, "share": {k: b[k] / n for k in NS}, "mean_ns_per_wave_instruction": mean}
        tot.update(b)
        print(f"{klass:26s} kernels {len(names):2d}  static VALU {n:7d}  full {b['full'] / n:.3f} half {b['half'] / n:.3f} mad64 {b['mad64'] / n:.3f}  mean {mean:.3f} ns")
    n = sum(tot.values())
    res["default_mean_ns_per_wave_instruction"] = sum(tot[k] * NS[k] for k in tot) / n
    json.dump(res, open(os.path.join(ROOT, "profiles", f"{tag}_valu_mix.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
