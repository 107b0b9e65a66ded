"""Independent restatement (numpy, written from the reference files, NOT from cairo_m_amd/csrc/air/*.hpp) of
`Claim::write_trace` for the six opcode components the metric workload (fibonacci_loop) exercises — they carry
> 99.9 % of its committed cells:

    store_fp_imm   crates/prover/src/components/opcodes/store_fp_imm.rs:147-296   (18 columns)
    store_fp_fp    .../store_fp_fp.rs:153-318                                     (20 columns)
    jnz_fp_imm     .../jnz_fp_imm.rs:121-...                                      (12 columns)
    jmp_imm        .../jmp_imm.rs                                                 (7 columns)
    store_imm      .../store_imm.rs                                               (9 columns)
    ret            .../ret.rs                                                     (9 columns)

The C++ AIR text is shared by the oracle and the HIP product; this test pins its column order and
witness semantics for the hot components against a second, separately written transcription: the trace columns
the oracle generates must equal the numpy model cell by cell (padding rows included).  The GPU suite then ties
the HIP trace to the oracle's (bit-identical proofs)."""
import numpy as np
import pytest

from cairo_m_amd.lib import prover_input_arrays, synth_fibonacci

P = 2**31 - 1
RET, STORE_ADD_FP_FP, STORE_ADD_FP_IMM, JMP_ABS_IMM = 11, 0, 4, 12


def inv(x):
    x = np.asarray(x, dtype=np.uint64)
    out = np.zeros_like(x)
    for i, v in enumerate(x.tolist()):
        out[i] = pow(v, P - 2, P) if v else 0
    return out


def mul(a, b):
    return (np.asarray(a, dtype=np.uint64) * np.asarray(b, dtype=np.uint64)) % P


def add(a, b):
    return (np.asarray(a, dtype=np.uint64) + np.asarray(b, dtype=np.uint64)) % P


def sub(a, b):
    return (np.asarray(a, dtype=np.uint64) + P - np.asarray(b, dtype=np.uint64) % P) % P


class Rows:
    """Bundles of one component padded to 2^log rows with ExecutionBundle::default() (adapter/memory.rs:112-124:
    pc = fp = clock = 0, instruction RET, prev_clock 0, empty access span) + per-lane access gathers
    (utils/data_accesses.rs:10-28: field of access k, 0 when k >= span_len)."""

    def __init__(self, bundles, accesses):
        n = bundles.shape[0]
        self.n = n
        self.size = max(16, 1 << int(np.ceil(np.log2(max(n, 1)))))
        b = np.zeros((self.size, 12), dtype=np.uint64)
        b[:n] = bundles
        b[n:, 4] = RET
        self.pc, self.fp, self.clock, self.ipc = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
        self.inst = [b[:, 4 + k] for k in range(6)]
        self.start, self.len = b[:, 10].astype(np.int64), b[:, 11].astype(np.int64)
        self.acc = accesses.astype(np.uint64) if accesses.size else np.zeros((1, 4), dtype=np.uint64)
        self.enabler = (np.arange(self.size) < n).astype(np.uint64)

    def access(self, k, field):  # field: 0 address, 1 prev_clock, 2 prev_value, 3 value
        idx = np.minimum(self.start + k, self.acc.shape[0] - 1)
        return np.where(k < self.len, self.acc[idx, field], 0).astype(np.uint64)


def store_fp_imm(r):
    op, src_off, imm, dst_off = r.inst[0], r.inst[1], r.inst[2], r.inst[3]
    imm_inv = inv(np.where(op == RET, 0, imm))
    flag = np.where(op >= STORE_ADD_FP_IMM, op - STORE_ADD_FP_IMM, 0)          # saturating_sub
    src_val = r.access(0, 3)
    return [r.enabler, r.pc, r.fp, r.clock, r.ipc, src_off, imm, dst_off, r.access(0, 1), src_val, imm_inv,
            r.access(1, 1), r.access(1, 2), r.access(1, 3), (flag // 2) * r.enabler, (flag % 2) * r.enabler,
            mul(src_val, imm), mul(src_val, imm_inv)]


def store_fp_fp(r):
    op = r.inst[0]
    flag = np.where(op == RET, 0, op - STORE_ADD_FP_FP)
    op0, op1 = r.access(0, 3), r.access(1, 3)
    op1_inv = inv(op1)
    return [r.enabler, r.pc, r.fp, r.clock, r.ipc, r.inst[1], r.inst[2], r.inst[3], r.access(0, 1), op0, r.access(1, 1), op1,
            op1_inv, r.access(2, 1), r.access(2, 2), r.access(2, 3), flag // 2, flag % 2, mul(op0, op1), mul(op0, op1_inv)]


def jnz_fp_imm(r):
    off0, imm = r.inst[1], r.inst[2]
    op0 = r.access(0, 3)
    taken = (op0 != 0).astype(np.uint64)
    pc_new = add(add(r.pc, 1), mul(taken, sub(imm, 1)))
    return [r.enabler, r.pc, r.fp, r.clock, r.ipc, off0, imm, r.access(0, 1), op0, inv(op0), taken, pc_new]


def jmp_imm(r):
    return [r.enabler, r.pc, r.fp, r.clock, r.ipc, r.inst[1], mul(r.enabler, sub(r.inst[0], JMP_ABS_IMM))]


def store_imm(r):
    return [r.enabler, r.pc, r.fp, r.clock, r.ipc, r.inst[1], r.inst[2], r.access(0, 1), r.access(0, 2)]


def ret(r):
    return [r.enabler, r.pc, r.fp, r.clock, r.ipc, r.access(1, 1), r.access(1, 3), r.access(0, 1), r.access(0, 3)]


MODELS = {"store_fp_imm": (store_fp_imm, {4, 6}), "store_fp_fp": (store_fp_fp, {0, 1, 2, 3}), "jnz_fp_imm": (jnz_fp_imm, {14}),
          "jmp_imm": (jmp_imm, {12, 13}), "store_imm": (store_imm, {9}), "ret": (ret, {11})}


@pytest.mark.parametrize("n", [4, 37])
def test_hot_component_traces_match_independent_model(oracle, n):
    inp = synth_fibonacci(n)
    a = prover_input_arrays(inp.view)
    seen = set()
    for cid in range(26):
        b = a[f"bundles{cid}"]
        if b.shape[0] == 0:
            continue
        opcodes = set(int(x) for x in np.unique(b[:, 4]))
        name = next((k for k, (_, ops) in MODELS.items() if opcodes <= ops), None)
        assert name is not None, f"component {cid} executes opcodes {opcodes}: not one of the fibonacci components"
        seen.add(name)
        want = np.stack([np.asarray(c, dtype=np.uint64) % P for c in MODELS[name][0](Rows(b, a["data_accesses"]))]).astype(np.uint32)
        got = oracle.component_trace(inp.view, cid)
        assert got.shape == want.shape, (name, got.shape, want.shape)
        bad = np.argwhere(got != want)
        assert bad.size == 0, f"{name}: first differing (column, row) = {bad[0]}, got {got[tuple(bad[0])]} want {want[tuple(bad[0])]}"
    assert seen == set(MODELS), seen
    inp.free()
