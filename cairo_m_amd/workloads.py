"""Synthetic CASM workloads for the prover (neither box has the Cairo-M compiler or runner, so the build
ships deterministic generators whose programs run on the synthetic VM of cairo_m_amd/csrc/host_adapter.hpp).

* ``all_opcodes_program``  — BASELINE configs[4]: a loop whose body touches every opcode-component kind once per
  iteration with operands from xorshift32(0xC0FFEE), constrained to each opcode's validity (u32 limbs < 2^16, divisors
  != 0, no instruction reads and writes the same cell).  Reference analogue: test_data/functions/all_opcodes.cm,
  crates/prover/tests/prover.rs:323-344.
* ``sha256_program``       — BASELINE configs[2]: SHA-256 of a padded message, hand-assembled from the algorithm of
  examples/sha256-cairo-m/src/sha256.cm (u32 add / and / xor / or, rotr as mul + div/rem + or); the VM's digest is
  checked against hashlib in the tests, a functional check of the u32 opcode semantics that owes nothing to the AIR text.

Programs are lists of instructions, each a list of M31 words (opcode first), the form ``cairo_m_amd.lib.vm_run`` /
``vm_segment`` take.  Instruction encodings: crates/common/src/instruction.rs:314-577.
"""
P = 2**31 - 1


def neg(k):
    return (P - k) % P


def xorshift32(seed):
    x = seed & 0xFFFFFFFF
    while True:
        x ^= (x << 13) & 0xFFFFFFFF
        x ^= x >> 17
        x ^= (x << 5) & 0xFFFFFFFF
        yield x


class Asm:
    """Tiny assembler: instructions of more than 4 words occupy two program cells (pc advances by ceil(words / 4));
    operands given as ('rel', label) / ('abs', label) are resolved against the label table."""

    def __init__(self):
        self.ins, self.pcs, self.labels, self.pc = [], [], {}, 0

    def label(self, name):
        self.labels[name] = self.pc

    def emit(self, *words):
        self.ins.append(list(words))
        self.pcs.append(self.pc)
        self.pc += (len(words) + 3) // 4
        return self

    def build(self):
        out = []
        for pc, ins in zip(self.pcs, self.ins):
            ws = []
            for w in ins:
                if isinstance(w, tuple):
                    kind, lab = w
                    tgt = self.labels[lab]
                    w = (tgt - pc) % P if kind == "rel" else tgt
                ws.append(w % P)
            out.append(ws)
        return out


# opcode ids (crates/common/src/instruction.rs:314-577)
ADD, SUB, MUL, DIV, ADDI, MULI, DDEREF, STOREI, CALL, RET, JMPA, JMPR, JNZ = 0, 1, 2, 3, 4, 6, 8, 9, 10, 11, 12, 13, 14
U_ADD, U_SUB, U_MUL, U_DIV, U_ADDI, U_MULI, U_DIVI, U_IMM, U_EQ, U_LT, U_EQI, U_LTI = 15, 16, 17, 18, 19, 21, 22, 23, 24, 28, 30, 34
U_AND, U_OR, U_XOR, U_ANDI, U_ORI, U_XORI = 36, 37, 38, 39, 40, 41
DDEREF_FF, SFP, TO_DDEREF, TO_DDEREF_FF, LE, ASSERT_EQ = 42, 43, 44, 45, 48, 50


def all_opcodes_program(iters, seed=0xC0FFEE):
    """Loop of `iters` iterations; every iteration executes each opcode of every opcode-component kind once (both
    directions of jnz, abs and rel jumps, call + ret) on operands that change through state feedback.
    U32StoreEqFpFp / U32StoreEqFpImm are NOT emitted: the reference AIR looks their instruction words up at the wrong
    place (u32_store_eq_fp_fp.rs:210, u32_store_eq_fp_imm.rs:250-251), so a live row cannot balance the LogUp sum in the
    reference either — those two components run their padding rows only.  Returns (program, steps) with
    steps = prologue + iters * per_iteration + 1."""
    assert iters >= 1
    rng = xorshift32(seed)
    r31 = lambda: next(rng) % P
    r16 = lambda: next(rng) & 0xFFFF
    nz16 = lambda: (next(rng) & 0xFFFF) | 1
    # frame slots (offsets from fp)
    A, B, NZ, C, D, E, G, H, K, LEC, LE2, T, X, X2, T2, T3 = range(16)
    UA, UB, UONE, UC, UD, UE, UDV, UQ, UR, UF, UG, UH, UI, UJ, UK, UL, UM, UN = range(20, 56, 2)
    LT, LT2 = 56, 57
    CNT0, CNT1, PTR, OFF, OFF2, DD0, DD1, CALLF = 60, 61, 62, 63, 64, 65, 66, 70
    a = Asm()
    # ---- prologue
    a.emit(STOREI, r31(), A).emit(STOREI, r31(), B).emit(STOREI, r31() | 1, NZ)
    a.emit(U_IMM, r16(), r16(), UA).emit(U_IMM, r16(), r16(), UB).emit(U_IMM, 1, 0, UONE)
    a.emit(STOREI, iters, CNT0).emit(STOREI, D, OFF).emit(STOREI, DD1, OFF2)
    prologue = len(a.ins)
    a.label("loop")
    n0 = len(a.ins)
    # ---- felt opcodes
    a.emit(ADD, A, B, C).emit(SUB, C, A, D).emit(MUL, C, B, E).emit(DIV, E, NZ, G)
    a.emit(ADDI, G, r31(), H).emit(MULI, H, r31(), K)
    a.emit(LE, K, r31(), LEC).emit(LE, CNT0, iters // 2, LE2)
    imm = r31()
    a.emit(STOREI, imm, T).emit(ASSERT_EQ, T, imm)
    a.emit(SFP, 0, PTR)
    a.emit(DDEREF, PTR, C, X)                 # X = [[PTR] + C-slot] = C
    a.emit(TO_DDEREF, PTR, DD0, X)            # [[PTR] + DD0] = X
    a.emit(DDEREF_FF, PTR, OFF, X2)           # X2 = [[PTR] + [OFF]] = D
    a.emit(TO_DDEREF_FF, PTR, OFF2, X2)       # [[PTR] + [OFF2]] = X2
    a.emit(CALL, CALLF, ("abs", "callee"))
    a.emit(JNZ, LE2, ("rel", "after_skip1"))   # taken in the second half of the run, not taken in the first
    a.emit(STOREI, r31(), T2)
    a.label("after_skip1")
    a.emit(JMPR, ("rel", "after_skip2"))
    a.emit(STOREI, r31(), T2)
    a.label("after_skip2")
    a.emit(JMPA, ("abs", "u32part"))
    a.emit(STOREI, r31(), T2)
    a.label("u32part")
    # ---- u32 opcodes
    a.emit(U_ADD, UA, UB, UC).emit(U_SUB, UC, UB, UD).emit(U_MUL, UA, UB, UE)
    a.emit(U_OR, UB, UONE, UDV)
    a.emit(U_DIV, UE, UDV, UQ, UR)
    a.emit(U_ADDI, UQ, r16(), r16(), UF).emit(U_MULI, UF, r16(), r16(), UG)
    a.emit(U_DIVI, UG, nz16(), r16() & 0xFF, UH, UI)
    a.emit(U_LT, UA, UB, LT).emit(U_LTI, UC, r16(), r16(), LT2)
    a.emit(U_AND, UG, UE, UJ).emit(U_XOR, UJ, UC, UK)
    a.emit(U_ANDI, UK, r16(), r16(), UL).emit(U_ORI, UL, r16(), r16(), UM).emit(U_XORI, UM, r16(), r16(), UN)
    # ---- state feedback + loop counter (no instruction reads and writes one cell)
    a.emit(U_ADD, UN, UR, UA).emit(U_SUB, UD, UI, UB)
    a.emit(ADD, K, B, A).emit(ADD, E, H, B)
    a.emit(ADDI, CNT0, neg(1), CNT1).emit(ADDI, CNT1, 0, CNT0)
    a.emit(JNZ, CNT0, ("rel", "loop"))
    per_iter_listed = len(a.ins) - n0
    a.emit(RET)
    a.label("callee")
    a.emit(STOREI, r31(), 0).emit(RET)
    # executed per iteration: every listed instruction except the three skipped STOREIs (one of them runs in the first
    # half, when LE2 = 0), plus the callee's two
    half = iters // 2            # iterations with CNT0 <= iters // 2  ->  LE2 = 1  ->  jnz taken
    steps = prologue + iters * (per_iter_listed - 3 + 2) + (iters - half) + 1
    return a.build(), steps


# ----------------------------------------------------------------------------------------------------------------
# SHA-256 (FIPS 180-4) on the u32 opcodes, the way examples/sha256-cairo-m/src/sha256.cm computes it: rotr(x, n) =
# (x / 2^n) | (x * 2^(32 - n)) with U32StoreDivRemFpImm / U32StoreMulFpImm / U32StoreOrFpFp (sha256.cm:17-27), shr by
# the quotient alone, ch / maj / sigma from and / xor, `!x` as x ^ 0xFFFFFFFF.
SHA_K = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
SHA_H0 = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]


def sha256_pad(msg: bytes):
    """FIPS 180-4 padding -> list of 32-bit big-endian words (a multiple of 16)."""
    ml = len(msg) * 8
    m = msg + b"\x80" + b"\x00" * ((55 - len(msg)) % 64) + ml.to_bytes(8, "big")
    return [int.from_bytes(m[i:i + 4], "big") for i in range(0, len(m), 4)]


class _ShaGen:
    """Straight-line code generator: every u32 value lives in a fresh pair of frame cells (single assignment), so no
    instruction reads and writes one cell."""

    def __init__(self):
        self.a = Asm()
        self.next = 0

    def slot(self):
        s = self.next
        self.next += 2
        return s

    def imm(self, v):
        d = self.slot()
        self.a.emit(U_IMM, v & 0xFFFF, v >> 16, d)
        return d

    def op3(self, op, x, y):
        d = self.slot()
        self.a.emit(op, x, y, d)
        return d

    def opi(self, op, x, v):
        d = self.slot()
        self.a.emit(op, x, v & 0xFFFF, (v >> 16) & 0xFFFF, d)
        return d

    def shr(self, x, n):
        q, r = self.slot(), self.slot()
        self.a.emit(U_DIVI, x, (1 << n) & 0xFFFF, (1 << n) >> 16, q, r)
        return q

    def rotr(self, x, n):
        hi = self.opi(U_MULI, x, 1 << (32 - n))     # wraps: x << (32 - n)
        return self.op3(U_OR, self.shr(x, n), hi)

    def add(self, x, y):
        return self.op3(U_ADD, x, y)

    def xor(self, x, y):
        return self.op3(U_XOR, x, y)

    def and_(self, x, y):
        return self.op3(U_AND, x, y)


def sha256_program(msg: bytes):
    """SHA-256(msg) as one straight-line CASM program (64 rounds per 64-byte block, message schedule included).
    Returns (program, digest_slots): the eight u32 digest words end up in the frame cells `digest_slots[i]` (lo, hi limb
    in consecutive cells)."""
    g = _ShaGen()
    words = sha256_pad(msg)
    h = [g.imm(v) for v in SHA_H0]
    for blk in range(0, len(words), 16):
        w = [g.imm(v) for v in words[blk:blk + 16]]
        for t in range(16, 64):
            s0 = g.xor(g.xor(g.rotr(w[t - 15], 7), g.rotr(w[t - 15], 18)), g.shr(w[t - 15], 3))
            s1 = g.xor(g.xor(g.rotr(w[t - 2], 17), g.rotr(w[t - 2], 19)), g.shr(w[t - 2], 10))
            w.append(g.add(g.add(w[t - 16], s0), g.add(w[t - 7], s1)))
        a, b, c, d, e, f, gg, hh = h
        for t in range(64):
            S1 = g.xor(g.xor(g.rotr(e, 6), g.rotr(e, 11)), g.rotr(e, 25))
            ch = g.xor(g.and_(e, f), g.and_(g.opi(U_XORI, e, 0xFFFFFFFF), gg))
            t1 = g.opi(U_ADDI, g.add(g.add(hh, S1), g.add(ch, w[t])), SHA_K[t])
            S0 = g.xor(g.xor(g.rotr(a, 2), g.rotr(a, 13)), g.rotr(a, 22))
            maj = g.xor(g.xor(g.and_(a, b), g.and_(a, c)), g.and_(b, c))
            t2 = g.add(S0, maj)
            hh, gg, f, e, d, c, b, a = gg, f, e, g.add(d, t1), c, b, a, g.add(t1, t2)
        h = [g.add(x, y) for x, y in zip(h, [a, b, c, d, e, f, gg, hh])]
    g.a.emit(RET)
    return g.a.build(), h
