#!/usr/bin/env python3
"""A small interpreter for the subset of Rust the reference's `write_trace` row closures are written in.

Test tooling only (never imported by the product, never shipped to the GPU box as anything but a script): the build
image has no rustc, so the reference cannot be run — but its per-row witness code is plain expression code over
`PackedM31` lanes (`let`, closures, `if`/`else`, `for`, tuples / arrays, iterator adaptors such as
`.to_array().iter().zip(..).map(|(x, y)| ..).collect()`), which this module lexes, parses (recursive descent + operator
precedence) and evaluates with Stwo's value semantics:

    M31          -> Felt   (canonical value in [0, P), field arithmetic)
    PackedM31    -> Packed (16 lanes of Felt, lane-wise arithmetic; N_LANES = 16)
    u32 / usize  -> Python int

tools/rsref/rs_witness.py drives it over the reference files to produce golden witness vectors.
"""
import re

P = 2**31 - 1
N_LANES = 16


# ---------------------------------------------------------------------------------------------- values
class Felt:
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = int(v) % P

    def _o(self, o):
        if isinstance(o, Felt):
            return o.v
        raise TypeError(f"M31 arithmetic with {type(o).__name__}")

    def __add__(self, o):
        if isinstance(o, Packed):
            return o.__radd__(self)
        return Felt(self.v + self._o(o))

    def __sub__(self, o):
        if isinstance(o, Packed):
            return Packed([self - x for x in o.lanes])
        return Felt(self.v - self._o(o))

    def __mul__(self, o):
        if isinstance(o, Packed):
            return o.__rmul__(self)
        return Felt(self.v * self._o(o))

    def __neg__(self):
        return Felt(-self.v)

    def __eq__(self, o):
        return isinstance(o, Felt) and o.v == self.v

    def __hash__(self):
        return hash(self.v)

    def inverse(self):
        assert self.v != 0, "M31::inverse of zero"
        return Felt(pow(self.v, P - 2, P))

    def __repr__(self):
        return f"M31({self.v})"


class Packed:
    __slots__ = ("lanes",)

    def __init__(self, lanes):
        lanes = list(lanes)
        assert len(lanes) == N_LANES and all(isinstance(x, Felt) for x in lanes), "PackedM31 needs 16 M31 lanes"
        self.lanes = lanes

    @staticmethod
    def broadcast(f):
        return Packed([f] * N_LANES)

    def _z(self, o):
        if isinstance(o, Packed):
            return o.lanes
        if isinstance(o, Felt):
            return [o] * N_LANES
        raise TypeError(f"PackedM31 arithmetic with {type(o).__name__}")

    def __add__(self, o): return Packed([a + b for a, b in zip(self.lanes, self._z(o))])
    def __radd__(self, o): return Packed([b + a for a, b in zip(self.lanes, self._z(o))])
    def __sub__(self, o): return Packed([a - b for a, b in zip(self.lanes, self._z(o))])
    def __mul__(self, o): return Packed([a * b for a, b in zip(self.lanes, self._z(o))])
    def __rmul__(self, o): return Packed([b * a for a, b in zip(self.lanes, self._z(o))])
    def __neg__(self): return Packed([-a for a in self.lanes])
    def __repr__(self): return f"Packed({[x.v for x in self.lanes]})"


class Ref:
    """`&mut` element of an array (what `iter_mut()` yields): reads and writes go to the slot."""
    __slots__ = ("lst", "i")

    def __init__(self, lst, i):
        self.lst, self.i = lst, i

    def get(self):
        return self.lst[self.i]

    def set(self, v):
        self.lst[self.i] = v


def deref(v):
    return v.get() if isinstance(v, Ref) else v


def copy_value(v):
    """Rust arrays / tuples of Copy types have value semantics: `let a = b;` and `x = b;` copy."""
    return list(v) if isinstance(v, list) else v


class Struct:
    """Plain record (PackedExecutionBundle, DataAccess, ...)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


# ---------------------------------------------------------------------------------------------- lexer
TOKEN_RE = re.compile(r"""
    (?P<ws>\s+|//[^\n]*)
  | (?P<num>0x[0-9a-fA-F_]+|\d[\d_]*)(?P<suffix>u8|u16|u32|u64|usize|i32|i64)?
  | (?P<id>[A-Za-z_][A-Za-z0-9_]*)
  | (?P<op>::|->|=>|==|!=|<=|>=|&&|\|\||<<|>>|\+=|-=|\*=|\.\.=|\.\.|[-+*/%&|^!<>=.,;:()\[\]{}#?])
""", re.X)


def lex(src):
    out, i = [], 0
    while i < len(src):
        m = TOKEN_RE.match(src, i)
        if not m:
            raise SyntaxError(f"cannot lex at {src[i:i + 30]!r}")
        i = m.end()
        if m.group("ws"):
            continue
        if m.group("num"):
            out.append(("num", int(m.group("num").replace("_", ""), 0)))
        elif m.group("id"):
            out.append(("id", m.group("id")))
        else:
            out.append(("op", m.group("op")))
    out.append(("eof", None))
    return out


# ---------------------------------------------------------------------------------------------- parser
BINOPS = [  # lowest to highest precedence
    ["||"], ["&&"], ["==", "!=", "<", ">", "<=", ">="], ["|"], ["^"], ["&"], ["<<", ">>"], ["+", "-"], ["*", "/", "%"],
]


class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self, k=0):
        return self.t[self.i + k]

    def at(self, kind, val=None, k=0):
        t = self.t[self.i + k]
        return t[0] == kind and (val is None or t[1] == val)

    def eat(self, kind, val=None):
        if not self.at(kind, val):
            raise SyntaxError(f"expected {kind} {val!r}, got {self.peek()} near token {self.i}: {self.t[max(0, self.i - 6):self.i + 4]}")
        t = self.t[self.i]
        self.i += 1
        return t

    def opt(self, kind, val=None):
        if self.at(kind, val):
            self.i += 1
            return True
        return False

    # ---- types are skipped: balanced brackets until a stop token at depth 0
    def skip_type(self, stops):
        depth = 0
        while True:
            k, v = self.peek()
            if k == "eof":
                return
            if depth == 0 and k == "op" and v in stops:
                return
            if k == "op" and v in ("(", "[", "<"):
                depth += 1
            elif k == "op" and v in (")", "]", ">"):
                if depth == 0:
                    return
                depth -= 1
            elif k == "op" and v == ">>":
                depth -= 2
            self.i += 1

    # ---- patterns
    def pattern(self):
        if self.opt("op", "("):
            items = []
            while not self.at("op", ")"):
                items.append(self.pattern())
                if not self.opt("op", ","):
                    break
            self.eat("op", ")")
            return ("ptuple", items)
        if self.opt("op", "["):
            items = []
            while not self.at("op", "]"):
                items.append(self.pattern())
                if not self.opt("op", ","):
                    break
            self.eat("op", "]")
            return ("ptuple", items)
        self.opt("op", "&")
        self.opt("id", "mut")
        name = self.eat("id")[1]
        return ("pname", name)

    # ---- blocks and statements
    def block(self):
        self.eat("op", "{")
        stmts, tail = [], None
        while not self.at("op", "}"):
            if self.opt("op", ";"):
                continue
            if self.at("id", "let"):
                self.i += 1
                pat = self.pattern()
                if self.opt("op", ":"):
                    self.skip_type(("=", ";"))
                init = None
                if self.opt("op", "="):
                    init = self.expr()
                self.eat("op", ";")
                stmts.append(("let", pat, init))
                continue
            if self.at("id", "for"):
                self.i += 1
                pat = self.pattern()
                self.eat("id", "in")
                it = self.expr(no_struct=True)
                body = self.block()
                stmts.append(("for", pat, it, body))
                continue
            e = self.expr()
            if self.opt("op", ";"):
                stmts.append(("expr", e))
            elif self.at("op", "}"):
                tail = e
            elif e[0] in ("if", "block"):      # block-like expression statement without `;`
                stmts.append(("expr", e))
            else:
                raise SyntaxError(f"expected ; or }} after expression, got {self.peek()} (expr {e[0]})")
        self.eat("op", "}")
        return ("block", stmts, tail)

    # ---- expressions
    def expr(self, no_struct=False):
        lhs = self.range_expr()
        if self.at("op") and self.peek()[1] in ("=", "+=", "-=", "*="):
            op = self.eat("op")[1]
            rhs = self.expr()
            return ("assign", op, lhs, rhs)
        return lhs

    def range_expr(self):
        if self.at("op", "..") or self.at("op", "..="):      # `..n` / `..=n`: from zero
            lhs = ("num", 0)
        else:
            lhs = self.binop(0)
        if self.at("op", "..") or self.at("op", "..="):
            incl = self.eat("op")[1] == "..="
            if self.at("op", "]") or self.at("op", ")"):
                return ("range", lhs, None, incl)          # `a[1..]`
            rhs = self.binop(0)
            return ("range", lhs, rhs, incl)
        return lhs

    def binop(self, level):
        if level == len(BINOPS):
            return self.cast()
        lhs = self.binop(level + 1)
        while self.at("op") and self.peek()[1] in BINOPS[level]:
            # `|` starts a closure only in prefix position, so here it is always bit-or
            op = self.eat("op")[1]
            rhs = self.binop(level + 1)
            lhs = ("bin", op, lhs, rhs)
        return lhs

    def cast(self):
        e = self.unary()
        while self.at("id", "as"):
            self.i += 1
            ty = self.eat("id")[1]
            e = ("cast", e, ty)
        return e

    def unary(self):
        if self.at("op", "-"):
            self.i += 1
            return ("neg", self.unary())
        if self.at("op", "!"):
            self.i += 1
            return ("not", self.unary())
        if self.at("op", "*"):
            self.i += 1
            return ("deref", self.unary())
        if self.at("op", "&"):
            self.i += 1
            self.opt("id", "mut")
            return self.unary()            # borrow
        if self.at("op", "&&"):
            self.i += 1
            return self.unary()
        return self.postfix()

    def args(self):
        self.eat("op", "(")
        a = []
        while not self.at("op", ")"):
            a.append(self.expr())
            if not self.opt("op", ","):
                break
        self.eat("op", ")")
        return a

    def skip_generics(self):
        # after `::` a `<` opens generic arguments (turbofish)
        if self.at("op", "<"):
            depth = 0
            while True:
                k, v = self.peek()
                self.i += 1
                if (k, v) == ("op", "<"):
                    depth += 1
                elif (k, v) == ("op", ">"):
                    depth -= 1
                elif (k, v) == ("op", ">>"):
                    depth -= 2
                if depth <= 0:
                    return

    def postfix(self):
        e = self.primary()
        while True:
            if self.at("op", "("):
                e = ("call", e, self.args())
            elif self.at("op", "["):
                self.i += 1
                idx = self.expr()
                self.eat("op", "]")
                e = ("index", e, idx)
            elif self.at("op", "."):
                self.i += 1
                if self.at("num"):
                    e = ("field", e, self.eat("num")[1])
                else:
                    name = self.eat("id")[1]
                    if self.at("op", "::"):
                        self.i += 1
                        self.skip_generics()
                    if self.at("op", "("):
                        e = ("method", e, name, self.args())
                    else:
                        e = ("field", e, name)
            elif self.at("op", "?"):
                self.i += 1
            else:
                return e

    def closure(self):
        params = []
        if not self.opt("op", "||"):
            self.eat("op", "|")
            while not self.at("op", "|"):
                params.append(self.pattern())
                if self.opt("op", ":"):
                    self.skip_type((",", "|"))
                if not self.opt("op", ","):
                    break
            self.eat("op", "|")
        if self.opt("op", "->"):
            self.skip_type(("{",))
            body = self.block()
        else:
            body = self.expr()
        return ("closure", params, body)

    def primary(self):
        k, v = self.peek()
        if k == "num":
            self.i += 1
            return ("num", v)
        if k == "op" and v in ("|", "||"):
            return self.closure()
        if k == "op" and v == "(":
            self.i += 1
            items, trailing = [], False
            while not self.at("op", ")"):
                items.append(self.expr())
                trailing = self.opt("op", ",")
                if not trailing:
                    break
            self.eat("op", ")")
            if len(items) == 1 and not trailing:
                return items[0]
            return ("tuple", items)
        if k == "op" and v == "[":
            self.i += 1
            items = []
            if not self.at("op", "]"):
                first = self.expr()
                if self.opt("op", ";"):
                    n = self.expr()
                    self.eat("op", "]")
                    return ("repeat", first, n)
                items.append(first)
                while self.opt("op", ","):
                    if self.at("op", "]"):
                        break
                    items.append(self.expr())
            self.eat("op", "]")
            return ("array", items)
        if k == "op" and v == "{":
            return self.block()
        if k == "id" and v == "if":
            self.i += 1
            cond = self.expr(no_struct=True)
            then = self.block()
            other = None
            if self.opt("id", "else"):
                other = self.primary() if self.at("id", "if") else self.block()
            return ("if", cond, then, other)
        if k == "id" and v == "unsafe":
            self.i += 1
            return self.block()
        if k == "id" and v == "move":
            self.i += 1
            return self.closure()
        if k == "id":
            path = [self.eat("id")[1]]
            while self.at("op", "::"):
                self.i += 1
                if self.at("op", "<"):
                    self.skip_generics()
                    continue
                path.append(self.eat("id")[1])
            return ("path", "::".join(path))
        raise SyntaxError(f"unexpected token {self.peek()} at {self.i}: {self.t[max(0, self.i - 6):self.i + 4]}")


def parse_fn(src):
    """`fn name<..>(a: T, b: &mut U) -> R where .. { body }` -> (name, [param names], body AST)"""
    m = re.search(r"fn (\w+)", src)
    name = m.group(1)
    i = src.index("(", m.end())
    depth, j = 0, i
    while True:
        depth += src[j] == "("
        depth -= src[j] == ")"
        if depth == 0:
            break
        j += 1
    params = []
    depth, cur = 0, ""
    for ch in src[i + 1:j] + ",":
        if ch == "," and depth == 0:
            if cur.strip():
                params.append(cur.split(":")[0].replace("mut", "").strip())
            cur = ""
            continue
        depth += ch in "([<"
        depth -= ch in ")]>"
        cur += ch
    b = src.index("{", j)
    return name, params, Parser(lex(src[b:])).block()


def parse_block(src):
    p = Parser(lex(src))
    b = p.block()
    return b


def parse_expr(src):
    p = Parser(lex(src))
    e = p.expr()
    if not p.at("eof"):
        raise SyntaxError(f"trailing tokens after expression: {p.peek()}")
    return e


# ---------------------------------------------------------------------------------------------- evaluator
class Env:
    def __init__(self, parent=None):
        self.vars, self.parent = {}, parent

    def get(self, name):
        e = self
        while e:
            if name in e.vars:
                return e.vars[name]
            e = e.parent
        raise NameError(name)

    def has(self, name):
        e = self
        while e:
            if name in e.vars:
                return True
            e = e.parent
        return False

    def set_existing(self, name, val):
        e = self
        while e:
            if name in e.vars:
                e.vars[name] = val
                return
            e = e.parent
        raise NameError(name)


def u32(x):
    assert 0 <= x < 2**64, f"integer out of range: {x} (the Rust code would have overflowed)"
    return x


def bind(pat, val, env):
    if pat[0] == "pname":
        if pat[1] != "_":
            env.vars[pat[1]] = val
        return
    vals = list(val)
    assert len(vals) == len(pat[1]), f"pattern arity {len(pat[1])} vs value {len(vals)}"
    for p, v in zip(pat[1], vals):
        bind(p, v, env)


class Interp:
    def __init__(self, globals_):
        self.g = globals_            # name / path -> value or python callable

    # -- helpers
    def truthy(self, v):
        assert isinstance(v, bool), f"condition is not a bool: {v!r}"
        return v

    def call_value(self, f, args):
        return f(*args)

    def make_closure(self, node, env):
        params, body = node[1], node[2]

        def fn(*args):
            e = Env(env)
            assert len(args) == len(params), f"closure arity {len(params)} vs {len(args)}"
            for p, a in zip(params, args):
                bind(p, a, e)
            return self.eval(body, e)
        return fn

    def assign(self, lhs, op, rhs, env):
        val = copy_value(deref(self.eval(rhs, env)))
        if lhs[0] == "deref":
            tgt = self.eval(lhs[1], env) if lhs[1][0] == "path" else None
            if isinstance(tgt, Ref):
                if op != "=":
                    val = self.arith({"+=": "+", "-=": "-", "*=": "*"}[op], tgt.get(), val)
                tgt.set(val)
                return None
            return self.assign_to(lhs[1], op, val, env)
        return self.assign_to(lhs, op, val, env)

    def assign_to(self, lhs, op, val, env):
        if lhs[0] == "array":              # destructuring assignment: [a, b, c] = expr
            vals = list(val)
            assert len(vals) == len(lhs[1])
            for t, v in zip(lhs[1], vals):
                self.assign_to(t, op, v, env)
            return None
        if lhs[0] == "path":
            if op != "=":
                val = self.arith({"+=": "+", "-=": "-", "*=": "*"}[op], env.get(lhs[1]), val)
            env.set_existing(lhs[1], val)
            return None
        if lhs[0] == "index":
            base = self.eval(lhs[1], env)
            idx = self.eval(lhs[2], env)
            if op != "=":
                val = self.arith({"+=": "+", "-=": "-", "*=": "*"}[op], base[idx], val)
            base[idx] = val
            return None
        if lhs[0] == "field":
            base = self.eval(lhs[1], env)
            setattr(base, str(lhs[2]), val)
            return None
        raise SyntaxError(f"cannot assign to {lhs[0]}")

    def arith(self, op, a, b):
        a, b = deref(a), deref(b)
        if isinstance(a, bool) or isinstance(b, bool):
            if op == "==": return a == b
            if op == "!=": return a != b
            if op in ("&&", "&"): return a and b
            if op in ("||", "|"): return a or b
            raise TypeError(f"bool {op}")
        ints = isinstance(a, int) and isinstance(b, int)
        if op == "+": return u32(a + b) if ints else a + b
        if op == "-":
            if ints:
                assert a >= b, f"u32 subtraction underflow: {a} - {b}"
                return a - b
            return a - b
        if op == "*": return u32(a * b) if ints else a * b
        if op in ("==", "!="):
            if isinstance(a, (Felt, int)) and type(a) is type(b):
                return (a == b) if op == "==" else (a != b)
            raise TypeError(f"comparison of {type(a).__name__} and {type(b).__name__}")
        assert ints, f"integer operator {op} on {type(a).__name__}, {type(b).__name__}"
        if op == "/": return a // b
        if op == "%": return a % b
        if op == "<<": return u32(a << b)
        if op == ">>": return a >> b
        if op == "&": return a & b
        if op == "|": return a | b
        if op == "^": return a ^ b
        if op == "<": return a < b
        if op == ">": return a > b
        if op == "<=": return a <= b
        if op == ">=": return a >= b
        raise SyntaxError(op)

    def lookup(self, name, env):
        if env.has(name):
            return env.get(name)
        if name in self.g:
            return self.g[name]
        raise NameError(f"unknown name {name}")

    # -- evaluation
    def eval(self, n, env):
        k = n[0]
        if k == "num":
            return n[1]
        if k == "path":
            return self.lookup(n[1], env)
        if k == "block":
            e = Env(env)
            for st in n[1]:
                self.stmt(st, e)
            return self.eval(n[2], e) if n[2] is not None else None
        if k == "tuple":
            return tuple(self.eval(x, env) for x in n[1])
        if k == "array":
            return [self.eval(x, env) for x in n[1]]
        if k == "repeat":
            return [self.eval(n[1], env)] * self.eval(n[2], env)
        if k == "deref":
            return deref(self.eval(n[1], env))
        if k == "neg":
            return -deref(self.eval(n[1], env))
        if k == "not":
            return not self.truthy(self.eval(n[1], env))
        if k == "cast":
            v = self.eval(n[1], env)
            if isinstance(v, bool):
                return int(v)
            assert isinstance(v, int), f"`as {n[2]}` on {type(v).__name__}"
            return v
        if k == "bin":
            op = n[1]
            if op == "&&":
                return self.truthy(self.eval(n[2], env)) and self.truthy(self.eval(n[3], env))
            if op == "||":
                return self.truthy(self.eval(n[2], env)) or self.truthy(self.eval(n[3], env))
            return self.arith(op, self.eval(n[2], env), self.eval(n[3], env))
        if k == "range":
            a, b = self.eval(n[1], env), self.eval(n[2], env)
            return list(range(a, b + 1 if n[3] else b))
        if k == "if":
            if self.truthy(self.eval(n[1], env)):
                return self.eval(n[2], env)
            return self.eval(n[3], env) if n[3] is not None else None
        if k == "closure":
            return self.make_closure(n, env)
        if k == "assign":
            return self.assign(n[2], n[1], n[3], env)
        if k == "index":
            base = deref(self.eval(n[1], env))
            if n[2][0] == "range" and n[2][2] is None:      # a[k..]
                return list(base[self.eval(n[2][1], env):])
            idx = self.eval(n[2], env)
            if isinstance(idx, list):          # slice by range
                return [base[i] for i in idx]
            return base[idx]
        if k == "field":
            base = deref(self.eval(n[1], env))
            if isinstance(n[2], int):
                if isinstance(base, Felt):
                    assert n[2] == 0
                    return base.v
                return base[n[2]]
            return getattr(base, n[2])
        if k == "call":
            f = self.eval(n[1], env)
            return self.call_value(f, [self.eval(a, env) for a in n[2]])
        if k == "method":
            recv = self.eval(n[1], env)
            args = [self.eval(a, env) for a in n[3]]
            return self.method(recv, n[2], args)
        raise SyntaxError(f"cannot evaluate {k}")

    def stmt(self, st, env):
        if st[0] == "let":
            bind(st[1], copy_value(deref(self.eval(st[2], env))) if st[2] is not None else None, env)
        elif st[0] == "expr":
            self.eval(st[1], env)
        elif st[0] == "for":
            for x in list(self.eval(st[2], env)):
                e = Env(env)
                bind(st[1], x, e)
                self.eval(st[3], e)

    def method(self, r, name, a):
        r = deref(r)
        if name == "iter_mut":
            return [Ref(r, i) for i in range(len(r))]
        if name == "clone" and isinstance(r, list):
            return list(r)
        if name == "fold":
            acc = a[0]
            for x in r:
                acc = a[1](acc, x)
            return acc
        # value-preserving adaptors
        if name in ("clone", "iter", "into_iter", "collect", "try_into", "unwrap", "copied", "cloned", "to_vec",
                    "as_slice", "into", "rev_placeholder"):
            return r
        if name == "to_array":
            assert isinstance(r, Packed)
            return list(r.lanes)
        if name == "map":
            if isinstance(r, Packed):
                raise TypeError("map on PackedM31")
            return [a[0](x) for x in r]
        if name == "zip":
            return [(x, y) for x, y in zip(r, a[0])]
        if name == "enumerate":
            return [(i, x) for i, x in enumerate(r)]
        if name == "for_each":
            for x in r:
                a[0](x)
            return None
        if name == "rev":
            return list(reversed(r))
        if name == "len":
            return len(r)
        if name == "inverse":
            return r.inverse()
        if name == "sort_by_key":       # slice::sort_by_key is stable, like Python's sort
            r.sort(key=a[0])
            return None
        if name == "saturating_sub":
            return max(0, r - a[0])
        if name == "wrapping_sub":
            return (r - a[0]) % 2**32
        if name == "wrapping_add":
            return (r + a[0]) % 2**32
        if name == "min":
            return min(r, a[0])
        if name == "max":
            return max(r, a[0])
        if name == "pow":
            return u32(r ** a[0])
        if name == "sum":
            return sum(r)
        if name == "fill":
            for i in range(len(r)):
                r[i] = a[0]
            return None
        if name == "get":
            i = a[0]
            if isinstance(i, list):
                return [r[j] for j in i] if (not i or i[-1] < len(r)) else None
            return r[i] if i < len(r) else None
        if name == "first":
            return r[0] if r else None
        if name == "unwrap_or":
            return a[0] if r is None else r
        if name == "unwrap_or_else":
            return a[0]() if r is None else r
        if name == "and_then":
            return None if r is None else a[0](r)
        # object-provided methods (Enabler::packed_at, ...)
        f = getattr(r, name, None)
        if callable(f):
            return f(*a)
        raise NameError(f"method .{name}() on {type(r).__name__}")


def m31_from(x):
    """M31::from(u32 / i32 / M31)."""
    if isinstance(x, Felt):
        return x
    if isinstance(x, bool):
        return Felt(int(x))
    assert isinstance(x, int)
    return Felt(x)


def packed_from(x):
    if isinstance(x, Packed):
        return x
    if isinstance(x, list):          # impl From<[M31; N_LANES]> for PackedM31
        return Packed(x)
    return Packed.broadcast(m31_from(x))


def standard_globals():
    g = {
        "M31::from": m31_from, "M31": lambda v: Felt(v), "BaseField::from": m31_from, "M31::zero": lambda: Felt(0), "M31::one": lambda: Felt(1),
        "M31::from_u32_unchecked": lambda v: Felt(v), "M31::inverse": lambda x: x.inverse(),
        "PackedM31::from": packed_from, "PackedM31::broadcast": lambda f: Packed.broadcast(m31_from(f)),
        "PackedM31::from_array": lambda l: Packed(l), "PackedM31::zero": lambda: Packed.broadcast(Felt(0)),
        "PackedM31::one": lambda: Packed.broadcast(Felt(1)),
        "u32::from": lambda b: int(b), "usize::from": lambda b: int(b),
        "std::array::from_fn": None,   # bound per call site (needs the target length)
        "N_LANES": N_LANES, "LOG_N_LANES": 4, "P": P,
    }
    return g
