"""A small evaluator of the RUST subset the reference's mdtests use for their "Rust equivalent" blocks (build container only).
`/root/reference/mdtest/**/*.md` pairs 39 Cairo-M programs with a Rust function its authors wrote to say what the program means; the
reference's test runner compiles both and compares the outputs as M31 values (crates/runner/tests/common/mod.rs:150-165: an i32 /
i64 result goes through `M31::from(i32)`, a u32 through `M31::from(u32)`).  tools/casm/make_casm_fixtures.py runs this evaluator on
those blocks and requires tools/casm/cm_eval.py — whose values are the `expected` side of tests/golden/casm/*.json — to agree
with them: the expected values of those fixtures are then pinned by a statement of the reference's authors, not only by an evaluator
written here.

Subset: `fn f(a: T, mut b: T) -> T { .. }` with a tail expression or `return`, `const N: T = e;`, `struct S { f: T }`, `let [mut]
pat[: T] = e;`, assignment to `x`, `x[i]`, `x.f`, `x.0`, `if / else` (statement and expression), `loop`, `while`, `for i in a..b`,
`break`, `continue`, `assert!(e)`, integer literals with suffixes, `true / false`, tuples, arrays `[a, b]`, struct literals,
`Vec::with_capacity(n)` / `.push(e)`, `e as T`, `.wrapping_add / wrapping_sub / wrapping_mul`, unary `- !`, binary
`* / %  + -  << >>  &  ^  |  == != < > <= >=  && ||`.  Integers are Python ints tagged with their Rust type; arithmetic that would
overflow its type raises Fault (Rust panics in debug builds), the wrapping methods wrap.  Anything else raises Unsupported."""
import re


class Unsupported(Exception):
    pass


class Fault(Unsupported):
    pass


TOKEN = re.compile(r"""\s+|//[^\n]*|\#\[[^\]]*\]|(?P<str>"(?:[^"\\]|\\.)*")|(?P<num>\d[\d_]*)(?P<suf>u32|u64|i32|i64|usize|u8)?|(?P<id>[A-Za-z_]\w*!?)|"""
                   r"(?P<op>->|\.\.|::|==|!=|<=|>=|&&|\|\||<<|>>|[-+*/%&|^!<>=(){};:,.\[\]])")
P31 = 2**31 - 1
RANGE = {"u8": (0, 2**8), "u32": (0, 2**32), "u64": (0, 2**64), "usize": (0, 2**64), "i32": (-2**31, 2**31), "i64": (-2**63, 2**63)}


def lex(src):
    out, i = [], 0
    while i < len(src):
        m = TOKEN.match(src, i)
        if not m:
            raise Unsupported(f"cannot lex {src[i:i + 20]!r}")
        i = m.end()
        if m.group("str") is not None:
            out.append(("str", m.group("str"), None))
        elif m.group("num") is not None:
            out.append(("num", int(m.group("num").replace("_", "")), m.group("suf")))
        elif m.group("id"):
            out.append(("id", m.group("id"), None))
        elif m.group("op"):
            out.append(("op", m.group("op"), None))
    out.append(("eof", None, None))
    return out


class Parser:
    PREC = [["||"], ["&&"], ["==", "!=", "<", ">", "<=", ">="], ["|"], ["^"], ["&"], ["<<", ">>"], ["+", "-"], ["*", "/", "%"]]

    def __init__(self, toks):
        self.t, self.i = toks, 0
        self.structs = {self.t[k + 1][1] for k in range(len(self.t) - 1) if self.t[k][:2] == ("id", "struct")}

    def at(self, kind, val=None, k=0):
        kk, v, _ = self.t[min(self.i + k, len(self.t) - 1)]
        return kk == kind and (val is None or v == val)

    def eat(self, kind, val=None):
        if not self.at(kind, val):
            raise Unsupported(f"expected {val or kind}, got {self.t[self.i]}")
        self.i += 1
        return self.t[self.i - 1]

    def opt(self, kind, val=None):
        if self.at(kind, val):
            self.i += 1
            return True
        return False

    def ty(self):
        if self.opt("op", "("):
            items = []
            while not self.at("op", ")"):
                items.append(self.ty())
                self.opt("op", ",")
            self.eat("op", ")")
            return ("tuple", tuple(items))
        if self.opt("op", "["):
            el = self.ty()
            self.eat("op", ";")
            n = self.eat("num")[1]
            self.eat("op", "]")
            return ("array", el, n)
        name = self.eat("id")[1]
        if name == "Vec":
            self.eat("op", "<")
            el = self.ty()
            self.eat("op", ">")
            return ("vec", el)
        return name

    def program(self):
        fns, structs, consts = [], {}, []
        while not self.at("eof"):
            if self.opt("id", "use"):
                while not self.opt("op", ";"):
                    self.i += 1
                continue
            if self.opt("id", "struct"):
                name = self.eat("id")[1]
                self.eat("op", "{")
                fields = []
                while not self.at("op", "}"):
                    f = self.eat("id")[1]
                    self.eat("op", ":")
                    fields.append((f, self.ty()))
                    self.opt("op", ",")
                self.eat("op", "}")
                structs[name] = fields
                continue
            if self.at("id", "const"):
                consts.append(self.stmt())
                continue
            self.eat("id", "fn")
            name = self.eat("id")[1]
            self.eat("op", "(")
            params = []
            while not self.at("op", ")"):
                self.opt("id", "mut")
                pn = self.eat("id")[1]
                self.eat("op", ":")
                params.append((pn, self.ty()))
                self.opt("op", ",")
            self.eat("op", ")")
            ret = self.ty() if self.opt("op", "->") else None
            fns.append((name, params, ret, self.block()))
        return fns, structs, consts

    def block(self):
        """-> ("block", statements, tail expression or None)"""
        self.eat("op", "{")
        stmts, tail = [], None
        while not self.at("op", "}"):
            s = self.stmt(allow_tail=True)
            if s[0] == "tail":
                tail = s[1]
                break
            stmts.append(s)
        self.eat("op", "}")
        return ("block", stmts, tail)

    def pattern(self):
        if self.opt("op", "("):
            items = []
            while not self.at("op", ")"):
                items.append(self.pattern())
                self.opt("op", ",")
            self.eat("op", ")")
            return ("ptuple", items)
        self.opt("id", "mut")
        return ("pvar", self.eat("id")[1])

    def is_assignment(self):
        k, depth = self.i, 0
        while k < len(self.t):
            kind, v, _ = self.t[k]
            if kind == "op" and v in ("(", "["):
                depth += 1
            elif kind == "op" and v in (")", "]"):
                depth -= 1
            elif kind == "op" and v == "=" and depth == 0:
                return True
            elif (kind == "op" and v in (";", "{", "}")) or kind == "eof":
                return False
            elif kind == "op" and depth == 0 and v != ".":
                return False
            k += 1
        return False

    def stmt(self, allow_tail=False):
        if self.opt("id", "let") or self.at("id", "const"):
            const = self.opt("id", "const")
            pat = self.pattern()
            ty = self.ty() if self.opt("op", ":") else None
            self.eat("op", "=")
            e = self.expr()
            self.eat("op", ";")
            return ("let", pat, ty, e, const)
        if self.opt("id", "return"):
            e = None if self.at("op", ";") else self.expr()
            self.eat("op", ";")
            return ("return", e)
        if self.opt("id", "break"):
            self.eat("op", ";")
            return ("break",)
        if self.opt("id", "continue"):
            self.eat("op", ";")
            return ("continue",)
        if self.opt("id", "while"):
            c = self.expr(no_struct=True)
            return ("while", c, self.block())
        if self.opt("id", "loop"):
            return ("while", ("bool", True), self.block())
        if self.opt("id", "for"):
            var = self.eat("id")[1]
            self.eat("id", "in")
            lo = self.expr(no_struct=True, no_range=True)
            self.eat("op", "..")
            hi = self.expr(no_struct=True, no_range=True)
            return ("for", var, lo, hi, self.block())
        if self.at("id", "assert!"):
            self.i += 1
            e = self.expr()
            self.eat("op", ";")
            return ("assert", e)
        if self.at("id", "if"):
            e = self.expr()
            if self.opt("op", ";") or not allow_tail or not self.at("op", "}"):
                return ("expr", e)
            return ("tail", e)
        if self.at("id") and self.is_assignment():
            place = self.postfix()
            self.eat("op", "=")
            e = self.expr()
            self.eat("op", ";")
            return ("assign", place, e)
        e = self.expr()
        if self.opt("op", ";"):
            return ("expr", e)
        if allow_tail and self.at("op", "}"):
            return ("tail", e)
        raise Unsupported(f"expected ; got {self.t[self.i]}")

    def expr(self, level=0, no_struct=False, no_range=False):
        if level == len(self.PREC):
            e = self.unary(no_struct)
            while self.opt("id", "as"):
                e = ("cast", e, self.ty())
            return e
        left = self.expr(level + 1, no_struct)
        while self.at("op") and self.t[self.i][1] in self.PREC[level]:
            op = self.eat("op")[1]
            left = ("bin", op, left, self.expr(level + 1, no_struct))
        return left

    def unary(self, no_struct=False):
        if self.opt("op", "-"):
            return ("neg", self.unary(no_struct))
        if self.opt("op", "!"):
            return ("not", self.unary(no_struct))
        return self.postfix(no_struct)

    def postfix(self, no_struct=False):
        e = self.primary(no_struct)
        while True:
            if self.at("op", ".") and not self.at("op", "..", 0):
                self.i += 1
                if self.at("num"):
                    e = ("field", e, self.eat("num")[1])
                    continue
                name = self.eat("id")[1]
                if self.opt("op", "("):
                    args = []
                    while not self.at("op", ")"):
                        args.append(self.expr())
                        self.opt("op", ",")
                    self.eat("op", ")")
                    e = ("method", e, name, args)
                else:
                    e = ("field", e, name)
            elif self.opt("op", "["):
                ix = self.expr()
                self.eat("op", "]")
                e = ("index", e, ix)
            else:
                return e

    def primary(self, no_struct=False):
        if self.opt("op", "("):
            items, trailing = [], False
            while not self.at("op", ")"):
                items.append(self.expr())
                trailing = self.opt("op", ",")
            self.eat("op", ")")
            return items[0] if len(items) == 1 and not trailing else ("tuple", items)
        if self.opt("op", "["):
            items = []
            while not self.at("op", "]"):
                items.append(self.expr())
                self.opt("op", ",")
            self.eat("op", "]")
            return ("array", items)
        if self.at("num"):
            _, v, suf = self.eat("num")
            return ("num", v, suf)
        if self.at("str"):
            return ("str", self.eat("str")[1])
        if self.at("id", "println!"):   # output only: arguments are evaluated for their faults, nothing is returned
            self.i += 1
            self.eat("op", "(")
            args = []
            while not self.at("op", ")"):
                args.append(self.expr())
                self.opt("op", ",")
            self.eat("op", ")")
            return ("tuple", [a for a in args if a[0] != "str"])
        if self.opt("id", "if"):
            c = self.expr(no_struct=True)
            then = self.block()
            els = None
            if self.opt("id", "else"):
                els = ("block", [], self.primary()) if self.at("id", "if") else self.block()   # `else if`: the nested if is the value
            return ("if", c, then, els)
        name = self.eat("id")[1]
        if name in ("true", "false"):
            return ("bool", name == "true")
        if self.at("op", "::"):
            path = [name]
            while self.opt("op", "::"):
                path.append(self.eat("id")[1])
            name = "::".join(path)
        if self.opt("op", "("):
            args = []
            while not self.at("op", ")"):
                args.append(self.expr())
                self.opt("op", ",")
            self.eat("op", ")")
            return ("call", name, args)
        if name in self.structs and not no_struct and self.at("op", "{"):
            self.eat("op", "{")
            fields = []
            while not self.at("op", "}"):
                f = self.eat("id")[1]
                self.eat("op", ":")
                fields.append((f, self.expr()))
                self.opt("op", ",")
            self.eat("op", "}")
            return ("slit", name, fields)
        return ("var", name)


class Ret(Exception):
    def __init__(self, v):
        self.v = v


class Brk(Exception):
    pass


class Cont(Exception):
    pass


class Int:
    """an integer with its Rust type (None: an unsuffixed literal / a value inferred from one)"""
    __slots__ = ("v", "t")

    def __init__(self, v, t):
        self.v, self.t = v, t

    def __repr__(self):
        return f"{self.v}{self.t or ''}"


def fit(v, t):
    if t is not None and t in RANGE and not RANGE[t][0] <= v < RANGE[t][1]:
        raise Fault(f"{v} overflows {t}")
    return Int(v, t)


class Interp:
    def __init__(self, src):
        fns, self.structs, consts = Parser(lex(src)).program()
        self.fns = {n: (p, r, b) for n, p, r, b in fns}
        self.order = [f[0] for f in fns]
        self.steps = 0
        self.globals = {"u32::MAX": Int(2**32 - 1, "u32"), "i32::MAX": Int(2**31 - 1, "i32"), "u64::MAX": Int(2**64 - 1, "u64")}
        for c in consts:
            self.stmt(c, [self.globals], None)

    def call(self, name, args):
        if name not in self.fns:
            raise Unsupported(f"call of {name}")
        params, ret, body = self.fns[name]
        if len(params) != len(args):
            raise Unsupported("argument count")
        env = [self.globals, {pn: self.typed(self.copy(a), pt) for (pn, pt), a in zip(params, args)}]
        try:
            v = self.block(body, env, ret)
        except Ret as r:
            v = r.v
        return self.typed(v, ret) if ret is not None and v is not None else v

    def typed(self, v, ty):
        """value as the declared type: untyped integers adopt it (range-checked), aggregates element-wise"""
        if isinstance(v, Int):
            if isinstance(ty, str) and ty in RANGE:
                if v.t is not None and v.t != ty:
                    raise Unsupported(f"{v.t} where {ty} is declared")
                return fit(v.v, ty)
            return v
        if isinstance(v, tuple) and v and v[0] == "tuple" and isinstance(ty, tuple) and ty[0] == "tuple":
            return ("tuple", [self.typed(x, t) for x, t in zip(v[1], ty[1])])
        if isinstance(v, list) and isinstance(ty, tuple) and ty[0] in ("array", "vec"):
            v[:] = [self.typed(x, ty[1]) for x in v]
        return v

    def copy(self, v):
        if isinstance(v, list):
            return [self.copy(x) for x in v]
        if isinstance(v, tuple) and v and v[0] == "tuple":
            return ("tuple", [self.copy(x) for x in v[1]])
        if isinstance(v, dict):
            return {k: self.copy(x) for k, x in v.items()}
        return v

    def lookup(self, env, name):
        for scope in reversed(env):
            if name in scope:
                return scope
        raise Unsupported(f"unknown variable {name}")

    def block(self, b, env, ret):
        env.append({})
        try:
            for s in b[1]:
                self.stmt(s, env, ret)
            return self.ev(b[2], env, ret) if b[2] is not None else None
        finally:
            env.pop()

    def bind(self, pat, v, env):
        if pat[0] == "pvar":
            env[-1][pat[1]] = v
            return
        if not (isinstance(v, tuple) and v[0] == "tuple" and len(v[1]) == len(pat[1])):
            raise Unsupported("tuple pattern against a non-tuple")
        for p_, x in zip(pat[1], v[1]):
            self.bind(p_, x, env)

    def place(self, e, env):
        if e[0] == "var":
            return self.lookup(env, e[1]), e[1]
        if e[0] == "field":
            c, k = self.place(e[1], env)
            base = c[k]
            if isinstance(base, tuple) and base[0] == "tuple":
                return base[1], e[2]
            if isinstance(base, dict):
                return base, e[2]
        if e[0] == "index":
            base = self.ev(e[1], env, None)
            ix = self.ev(e[2], env, "usize")
            if not isinstance(base, list) or not 0 <= ix.v < len(base):
                raise Fault("index out of bounds")
            return base, ix.v
        raise Unsupported(f"assignment to {e[0]}")

    def stmt(self, s, env, ret):
        self.steps += 1
        if self.steps > 2_000_000:
            raise Unsupported("too many steps")
        k = s[0]
        if k == "let":
            self.bind(s[1], self.copy(self.typed(self.ev(s[3], env, s[2]), s[2])), env)
        elif k == "assign":
            c, key = self.place(s[1], env)
            old = c[key]
            ty = old.t if isinstance(old, Int) else None
            c[key] = self.copy(self.typed(self.ev(s[2], env, ty), ty))
        elif k == "return":
            raise Ret(None if s[1] is None else self.ev(s[1], env, ret))
        elif k == "break":
            raise Brk()
        elif k == "continue":
            raise Cont()
        elif k == "assert":
            if not self.truth(self.ev(s[1], env, None)):
                raise Fault("assertion fails")
        elif k == "while":
            while self.truth(self.ev(s[1], env, None)):
                try:
                    self.block(s[2], env, ret)
                except Brk:
                    break
                except Cont:
                    continue
        elif k == "for":
            lo, hi = self.ev(s[2], env, None), self.ev(s[3], env, None)
            t = lo.t or hi.t
            for i in range(lo.v, hi.v):
                env.append({s[1]: Int(i, t)})
                try:
                    self.block(s[4], env, ret)
                except Brk:
                    env.pop()
                    break
                except Cont:
                    pass
                env.pop()
        elif k == "expr":
            self.ev(s[1], env, None)
        else:
            raise Unsupported(k)

    @staticmethod
    def truth(v):
        if isinstance(v, bool):
            return v
        raise Unsupported("a condition that is not a bool")

    def ev(self, e, env, want):
        k = e[0]
        if k == "num":
            return fit(e[1], e[2] or (want if isinstance(want, str) and want in RANGE else None))
        if k == "bool":
            return e[1]
        if k == "var":
            return self.lookup(env, e[1])[e[1]]
        if k == "tuple":
            wt = want[1] if isinstance(want, tuple) and want[0] == "tuple" and len(want[1]) == len(e[1]) else [None] * len(e[1])
            return ("tuple", [self.ev(x, env, t) for x, t in zip(e[1], wt)])
        if k == "array":
            el = want[1] if isinstance(want, tuple) and want[0] in ("array", "vec") else None
            return [self.copy(self.ev(x, env, el)) for x in e[1]]
        if k == "slit":
            decl = dict(self.structs[e[1]])
            return {f: self.copy(self.typed(self.ev(x, env, decl[f]), decl[f])) for f, x in e[2]}
        if k == "field":
            base = self.ev(e[1], env, None)
            if isinstance(base, tuple) and base[0] == "tuple":
                return base[1][e[2]]
            if isinstance(base, dict):
                return base[e[2]]
            raise Unsupported("field of a scalar")
        if k == "index":
            base, ix = self.ev(e[1], env, None), self.ev(e[2], env, "usize")
            if not isinstance(base, list) or not 0 <= ix.v < len(base):
                raise Fault("index out of bounds")
            return base[ix.v]
        if k == "if":
            if self.truth(self.ev(e[1], env, None)):
                return self.block(e[2], env, want)
            return self.block(e[3], env, want) if e[3] is not None else None
        if k == "cast":
            v = self.ev(e[1], env, None)
            if not isinstance(v, Int) or e[2] not in RANGE:
                raise Unsupported(f"cast to {e[2]}")
            lo, hi = RANGE[e[2]]
            return Int((v.v - lo) % (hi - lo) + lo, e[2])   # `as` truncates / reinterprets
        if k == "call":
            if e[1] == "Vec::with_capacity" or e[1] == "Vec::new":
                return []
            if e[1] == "M31::from":   # stwo's field element: + - * / are the field operations
                return Int(self.ev(e[2][0], env, None).v % P31, "M31")
            params = self.fns[e[1]][0] if e[1] in self.fns else []
            return self.call(e[1], [self.ev(a, env, params[i][1] if i < len(params) else None) for i, a in enumerate(e[2])])
        if k == "method":
            r = self.ev(e[1], env, None)
            if e[2] == "push" and isinstance(r, list):
                r.append(self.copy(self.ev(e[3][0], env, None)))
                return None
            if e[2] in ("wrapping_add", "wrapping_sub", "wrapping_mul", "wrapping_div") and isinstance(r, Int):
                o = self.ev(e[3][0], env, r.t)
                t = r.t or o.t or "i32"
                lo, hi = RANGE[t]
                if e[2] == "wrapping_div":
                    if o.v == 0:
                        raise Fault("division by zero")
                    return Int(abs(r.v) // abs(o.v) * (1 if (r.v >= 0) == (o.v >= 0) else -1), t)
                x = {"wrapping_add": r.v + o.v, "wrapping_sub": r.v - o.v, "wrapping_mul": r.v * o.v}[e[2]]
                return Int((x - lo) % (hi - lo) + lo, t)
            raise Unsupported(f"method {e[2]}")
        if k == "neg":
            v = self.ev(e[1], env, want)
            return fit(-v.v, v.t)
        if k == "not":
            v = self.ev(e[1], env, None)
            if isinstance(v, bool):
                return not v
            raise Unsupported("! of a non-bool")
        if k == "bin":
            op = e[1]
            if op in ("&&", "||"):
                a = self.truth(self.ev(e[2], env, None))
                if op == "&&":
                    return a and self.truth(self.ev(e[3], env, None))
                return a or self.truth(self.ev(e[3], env, None))
            cmp_op = op in ("==", "!=", "<", ">", "<=", ">=")
            wt = None if cmp_op else (want if isinstance(want, str) else None)
            a, b = self.ev(e[2], env, wt), self.ev(e[3], env, wt)
            if isinstance(a, bool) and isinstance(b, bool) and op in ("==", "!="):
                return (a == b) == (op == "==")
            if not (isinstance(a, Int) and isinstance(b, Int)):
                raise Unsupported(f"operands of {op}")
            if op in ("<<", ">>"):
                t = a.t
            else:
                if a.t is not None and b.t is not None and a.t != b.t:
                    raise Unsupported(f"operands of {op}: {a.t} and {b.t}")
                t = a.t or b.t
            x, y = a.v, b.v
            if t == "M31" and not cmp_op:
                if op == "/":
                    if y % P31 == 0:
                        raise Fault("division by zero")
                    return Int(x * pow(y, P31 - 2, P31) % P31, "M31")
                if op in ("+", "-", "*"):
                    return Int((x + y if op == "+" else x - y if op == "-" else x * y) % P31, "M31")
                raise Unsupported(f"M31 {op}")
            if cmp_op:
                return {"==": x == y, "!=": x != y, "<": x < y, ">": x > y, "<=": x <= y, ">=": x >= y}[op]
            if op in ("/", "%"):
                if y == 0:
                    raise Fault("division by zero")
                q = abs(x) // abs(y) * (1 if (x >= 0) == (y >= 0) else -1)   # Rust truncates towards zero
                return fit(q if op == "/" else x - q * y, t)
            if op in ("<<", ">>") and not 0 <= y < 64:
                raise Fault("shift amount out of range")
            r = (x + y if op == "+" else x - y if op == "-" else x * y if op == "*" else x & y if op == "&" else x | y if op == "|"
                 else x ^ y if op == "^" else x << y if op == "<<" else x >> y)
            return fit(r, t)
        raise Unsupported(k)


def flat_ints(v):
    """the integers of a returned value in order (tuples / structs / arrays flattened): the words the runner compares"""
    if v is None:
        return []
    if isinstance(v, Int):
        return [v.v]
    if isinstance(v, bool):
        return [1 if v else 0]
    if isinstance(v, tuple) and v[0] == "tuple":
        return [x for it in v[1] for x in flat_ints(it)]
    if isinstance(v, dict):
        return [x for it in v.values() for x in flat_ints(it)]
    if isinstance(v, list):
        return [x for it in v for x in flat_ints(it)]
    raise Unsupported(type(v).__name__)


def mdtest_pairs(root="/root/reference/mdtest"):
    """-> [(markdown file, cairo-m source, rust source)]: every Cairo-M block of the reference's mdtests that is IMMEDIATELY followed
    by a Rust block (mdtest/README.md: "an optional Rust equivalent")"""
    import glob
    out = []
    for md in sorted(glob.glob(root + "/*/*.md")):
        blocks = re.findall(r"^```(\S*)[^\n]*\n(.*?)^```\s*$", open(md).read(), re.S | re.M)
        for (l0, c0), (l1, c1) in zip(blocks, blocks[1:]):
            if l0 == "cairo-m" and l1 == "rust":
                out.append((md, c0, c1))
    return out
