"""The two small evaluators behind tests/golden/casm/ (tools/casm/cm_eval.py for Cairo-M sources, tools/casm/rust_eval.py for the
Rust equivalents of the reference's mdtests) on programs written here, with values worked out by hand: the fixture script trusts
them, so they get tests of their own (no reference tree needed)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "casm"))
import cm_eval  # noqa: E402
import rust_eval  # noqa: E402

P = 2**31 - 1


def cm(src, name, *args):
    return cm_eval.Interp(src).call(name, list(args))


def test_cairo_m_scalars_and_control_flow():
    src = """
    fn fib(n: felt) -> felt { let a = 0; let b = 1; let i = 0; while i != n { let t = a + b; a = b; b = t; i = i + 1; } return a; }
    fn wrap() -> u32 { let x: u32 = 4294967295; return x + 2; }
    fn inv() -> felt { return 1 / 2; }
    fn mix(x: u32) -> u32 { if x > 10u32 && !(x == 12u32) { return x % 7u32; } else { return x / 3u32; } }
    fn cnt() -> felt { let s = 0; for (let i = 0; i != 10; i = i + 1) { if i == 5 { continue; } if i == 8 { break; } s = s + i; } return s; }
    """
    assert cm(src, "fib", ("felt", 10)) == ("felt", 55)
    assert cm(src, "wrap") == ("u32", 1)
    assert cm(src, "inv") == ("felt", (P + 1) // 2)
    assert cm(src, "mix", ("u32", 20)) == ("u32", 6) and cm(src, "mix", ("u32", 12)) == ("u32", 4)
    assert cm(src, "cnt") == ("felt", 0 + 1 + 2 + 3 + 4 + 6 + 7)
    with pytest.raises(cm_eval.Fault):
        cm("fn z() -> felt { return 1 / 0; }", "z")


def test_cairo_m_aggregates():
    src = """
    struct P { x: u32, y: u32 }
    struct L { a: P, b: P }
    const T: [u32; 3] = [1, 2, 4];
    fn s() -> u32 { let l = L { a: P { x: 1, y: 2 }, b: P { x: 3, y: 4 } }; let m = l; m.b.y = 40; return l.b.y + m.b.y + T[2]; }
    fn t() -> felt { let (a, (b, c)) = (1, (2, 3)); let u = (a, b, c); u.0 = 10; return u.0 + u.1 + u.2; }
    fn arr(i: felt) -> felt { let a: [felt; 3] = [1, 2, 3]; a[1] = 20; let b = a; b[2] = 30; return a[i] + a[2]; }
    fn heap() -> u32 { let p: u32* = new u32[2]; p[0] = 7; p[1] = 8; return p[0] * p[1]; }
    fn cast() -> felt { let x: u32 = 2147483646; return x as felt; }
    fn pair() -> (felt, u32) { return (3, 4); }
    """
    assert cm(src, "s") == ("u32", 4 + 40 + 4)                    # structs are values: `m = l` copies
    assert cm(src, "t") == ("felt", 15)
    assert cm(src, "arr", ("felt", 1)) == ("felt", 20 + 30)       # arrays are references to their storage, as in the compiled code
    assert cm(src, "heap") == ("u32", 56)
    assert cm(src, "cast") == ("felt", P - 1)
    assert cm_eval.to_words(cm(src, "pair")) == [3, 4, 0]          # a u32 is two 16-bit limbs, low first
    with pytest.raises(cm_eval.Fault):
        cm("fn c() -> felt { let x: u32 = 2147483647; return x as felt; }", "c")


def test_rust_subset():
    src = """
    #[derive(Debug)]
    struct Pt { x: u32, y: u32 }
    fn tri(n: i64) -> i64 { let mut s = 0; for i in 0..n { if i == 3 { continue; } s = s + i; } return s; }
    fn w() -> u32 { let m: u32 = u32::MAX; m.wrapping_add(2).wrapping_mul(3) }
    fn sel(a: u32) -> u32 { let v = if a > 100 { 5 } else if a > 50 { 2 } else { 0 }; v + 1 }
    fn agg() -> u32 { let mut ps: Vec<Pt> = Vec::with_capacity(2); ps.push(Pt { x: 1, y: 2 }); ps.push(Pt { x: 3, y: 4 }); let t = (ps[0].x, [ps[1].y, 9]); t.0 + t.1[0] }
    fn m31() -> M31 { M31::from(7) / M31::from(3) }
    fn neg() -> i32 { let a: i32 = 5; a - 9 }
    fn c() -> u32 { const K: u64 = (1u64 << 31) - 1; ((K - 1 + 5) % K) as u32 }
    """
    it = rust_eval.Interp(src)
    I = rust_eval.Int
    assert it.call("tri", [I(6, "i64")]).v == 0 + 1 + 2 + 4 + 5
    assert it.call("w", []).v == 3
    assert [it.call("sel", [I(a, "u32")]).v for a in (200, 60, 7)] == [6, 3, 1]
    assert it.call("agg", []).v == 1 + 4
    assert it.call("m31", []).v == 7 * pow(3, P - 2, P) % P
    assert it.call("neg", []).v == -4
    assert it.call("c", []).v == 4
    with pytest.raises(rust_eval.Fault):
        rust_eval.Interp("fn o() -> u32 { let a: u32 = 4294967295; a + 1 }").call("o", [])   # Rust panics on overflow in debug builds
