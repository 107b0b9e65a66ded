#!/usr/bin/env python3
"""Regenerate the Poseidon2-M31 (t=16, alpha=5, R_F=8, R_P=14) parameters used by the reference.

The reference takes them from the un-vendored crate `zkhash`
(/root/reference/Cargo.lock:5069-5071, crates/prover/build.rs:25-106).  That instance file is the
output of the public Poseidon2 parameter script (HorizenLabs `poseidon2_rust_params.sage`): a Grain
LFSR seeded with (field=1, sbox=0, n=31, t=16, R_F=8, R_P=14) produces the round constants, then the
internal-matrix diagonal candidates.  This script restates that generator in pure Python and checks
the result against the ONLY in-tree known answer: the permutation KAT of
/root/reference/crates/prover/tests/poseidon2.rs:14-34.
"""
import sys

P = 2**31 - 1
T, RF, RP, N = 16, 8, 14, 31


class Grain:
    def __init__(self, field, sbox, n, t, rf, rp):
        bits = []
        def push(v, w):
            bits.extend(int(c) for c in bin(v)[2:].zfill(w))
        push(field, 2); push(sbox, 4); push(n, 12); push(t, 12); push(rf, 10); push(rp, 10)
        bits.extend([1] * 30)
        assert len(bits) == 80
        self.s = bits
        for _ in range(160):
            self._step()

    def _step(self):
        s = self.s
        nb = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        s.pop(0)
        s.append(nb)
        return nb

    def bit(self):
        # self-shrinking: take pairs; if first is 1 output second
        while True:
            b1 = self._step()
            b2 = self._step()
            if b1 == 1:
                return b2

    def bits(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bit()
        return v


def gen_round_constants(g):
    n_consts = RF * T + RP
    out = []
    while len(out) < n_consts:
        v = g.bits(N)
        if v < P:
            out.append(v)
    return out


# ---------------- linear algebra over GF(P) for the min-poly condition ----------------
def mat_mul(A, B):
    n = len(A)
    return [[sum(A[i][k] * B[k][j] for k in range(n)) % P for j in range(n)] for i in range(n)]


def charpoly(M):
    """Characteristic polynomial via Faddeev-LeVerrier (needs division by 1..n, fine mod P)."""
    n = len(M)
    I = [[1 if i == j else 0 for j in range(n)] for i in range(n)]
    c = [0] * (n + 1)
    c[n] = 1
    Mk = [[0] * n for _ in range(n)]
    for k in range(1, n + 1):
        # Mk = M*Mk + c[n-k+1]*I
        Mk = mat_mul(M, Mk)
        for i in range(n):
            Mk[i][i] = (Mk[i][i] + c[n - k + 1]) % P
        AM = mat_mul(M, Mk)
        tr = sum(AM[i][i] for i in range(n)) % P
        c[n - k] = (-tr * pow(k, P - 2, P)) % P
    return c  # c[0] + c[1] x + ... + x^n


def poly_mod(a, m):
    a = a[:]
    dm = len(m) - 1
    inv = pow(m[-1], P - 2, P)
    while len(a) - 1 >= dm and a:
        if a[-1] == 0:
            a.pop(); continue
        f = a[-1] * inv % P
        sh = len(a) - 1 - dm
        for i in range(dm + 1):
            a[sh + i] = (a[sh + i] - f * m[i]) % P
        a.pop()
    while a and a[-1] == 0:
        a.pop()
    return a


def poly_mul_mod(a, b, m):
    if not a or not b:
        return []
    r = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                r[i + j] = (r[i + j] + x * y) % P
    return poly_mod(r, m)


def poly_pow_x_p(base, m):
    """base^P mod m."""
    r = [1]
    b = base
    e = P
    while e:
        if e & 1:
            r = poly_mul_mod(r, b, m)
        b = poly_mul_mod(b, b, m)
        e >>= 1
    return r


def poly_gcd(a, b):
    while b:
        a, b = b, poly_mod(a, b)
    return a


def is_irreducible(f):
    """Rabin test for degree n = 16 (prime divisors of 16: {2})."""
    n = len(f) - 1
    x = [0, 1]
    h = x
    for i in range(1, n + 1):
        h = poly_pow_x_p(h, f)
        if i == n // 2:
            d = h[:] + [0] * max(0, 2 - len(h))
            d[1] = (d[1] - 1) % P
            while d and d[-1] == 0:
                d.pop()
            g = poly_gcd(f, d)
            if len(g) - 1 > 0:
                return False
    return poly_mod(h, f) == poly_mod(x, f)


def check_minpoly_condition(M):
    n = len(M)
    Mt = M
    for _ in range(1, 2 * n + 1):
        cp = charpoly(Mt)
        if not is_irreducible(cp):  # irreducible charpoly <=> minpoly irreducible of full degree
            return False
        Mt = mat_mul(M, Mt)
    return True


def gen_internal_diag(g):
    tries = 0
    while True:
        tries += 1
        diag = [g.bits(N) % P for _ in range(T)]
        M = [[(diag[i] if i == j else 1) for j in range(T)] for i in range(T)]
        if check_minpoly_condition(M):
            return [(d - 1) % P for d in diag], tries


# ---------------- permutation (zkhash Poseidon2::permutation) ----------------
def m4(x):
    t0 = (x[0] + x[1]) % P; t1 = (x[2] + x[3]) % P
    t2 = (2 * x[1] + t1) % P; t3 = (2 * x[3] + t0) % P
    t4 = (4 * t1 + t3) % P; t5 = (4 * t0 + t2) % P
    t6 = (t3 + t5) % P; t7 = (t2 + t4) % P
    return [t6, t5, t7, t4]


def ext(s):
    s = s[:]
    for i in range(4):
        s[4 * i:4 * i + 4] = m4(s[4 * i:4 * i + 4])
    for j in range(4):
        tot = (s[j] + s[j + 4] + s[j + 8] + s[j + 12]) % P
        for i in range(4):
            s[4 * i + j] = (s[4 * i + j] + tot) % P
    return s


def perm(state, rc, diag):
    s = ext(state)
    ri = 0
    for r in range(RF // 2):
        s = [pow((s[i] + rc[ri + i]) % P, 5, P) for i in range(T)]; ri += T
        s = ext(s)
    for r in range(RP):
        s[0] = pow((s[0] + rc[ri]) % P, 5, P); ri += 1
        tot = sum(s) % P
        s = [(s[i] * diag[i] + tot) % P for i in range(T)]
    for r in range(RF // 2):
        s = [pow((s[i] + rc[ri + i]) % P, 5, P) for i in range(T)]; ri += T
        s = ext(s)
    return s


KAT = [0x505d9689, 0x3b64c904, 0x79e2fd81, 0x4ba8015f, 0x24b6d2f5, 0x23845add, 0x521f4314, 0x69dfb019,
       0x2aaae419, 0x6cb4502c, 0x6f7fa65a, 0x75feff24, 0x128d6587, 0x515877e4, 0x037f4dd7, 0x134b427f]

if __name__ == "__main__":
    g = Grain(1, 0, N, T, RF, RP)
    rc = gen_round_constants(g)
    print("rc[0..4] =", [hex(x) for x in rc[:4]], file=sys.stderr)
    diag, tries = gen_internal_diag(g)
    print("diag tries", tries, [hex(x) for x in diag[:4]], file=sys.stderr)
    out = perm(list(range(16)), rc, diag)
    ok = out == KAT
    print("KAT", "PASS" if ok else "FAIL", [hex(x) for x in out[:4]], file=sys.stderr)
    if ok and len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write("// Generated by tools/gen_poseidon2_m31.py — Poseidon2-M31 t=16 parameters (Grain LFSR,\n"
                    "// HorizenLabs poseidon2 parameter script); validated against the reference KAT\n"
                    "// /root/reference/crates/prover/tests/poseidon2.rs:14-34.\n#pragma once\n#include <stdint.h>\nnamespace air {\n")
            ext_rows = [rc[i * 16:(i + 1) * 16] for i in range(4)] + \
                       [rc[64 + 14 + i * 16:64 + 14 + (i + 1) * 16] for i in range(4)]
            f.write("static constexpr uint32_t P2_EXTERNAL_RC[8][16] = {\n")
            for row in ext_rows:
                f.write("  {" + ", ".join("%du" % v for v in row) + "},\n")
            f.write("};\nstatic constexpr uint32_t P2_INTERNAL_RC[14] = {" + ", ".join("%du" % v for v in rc[64:78]) + "};\n")
            f.write("static constexpr uint32_t P2_INTERNAL_DIAG[16] = {" + ", ".join("%du" % v for v in diag) + "};\n}\n")
    sys.exit(0 if ok else 1)
