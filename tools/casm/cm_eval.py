"""A small evaluator of the Cairo-M SOURCE subset the compiler's codegen snapshots are written in (build container only: it reads
`/root/reference/crates/compiler/codegen/tests/snapshots/*.snap`).  It exists to compute, from the source text of a snapshot, the
value its entry function returns — independently of the CASM listing, of the VM and of every line of the prover: the expected
side of tests/golden/casm/*.json.

Subset: `fn name(p: T, ..) -> T { .. }`, `let x[: T] = e;`, assignment, `if / else if / else`, `while`, `loop`, C-style
`for (let i = a; cond; i = e) { .. }`, `break`, `continue`, `return e;`, calls, literals (`10`, `10u32`, `true`, `false`), unary
`-` `!`, binary `* / %  + -  & ^ |  == != < > <= >=  && ||`.  Types: `felt` (arithmetic mod 2^31 - 1, `/` = multiplication by the
inverse), `u32` (wrapping; `/` and `%` integer), `bool`.  An untyped integer literal takes the type of the other operand, of the
declared variable, of the parameter or of the return type; alone it is a felt.  Anything else raises Unsupported: the fixture
generator then leaves that snapshot out."""
import re

P = 2**31 - 1


class Unsupported(Exception):
    pass


TOKEN = re.compile(r"\s+|//[^\n]*|(?P<num>\d+)(?P<suf>u32|felt)?|(?P<id>[A-Za-z_]\w*)|(?P<op>->|==|!=|<=|>=|&&|\|\||[-+*/%&|^!<>=(){};:,])")


def lex(src):
    out, i = [], 0
    while i < len(src):
        m = TOKEN.match(src, i)
        if not m:
            raise Unsupported(f"cannot lex {src[i:i + 20]!r}")
        i = m.end()
        if m.group("num") is not None:
            out.append(("num", int(m.group("num")), m.group("suf")))
        elif m.group("id"):
            out.append(("id", m.group("id"), None))
        elif m.group("op"):
            out.append(("op", m.group("op"), None))
    out.append(("eof", None, None))
    return out


class Parser:
    PREC = [["||"], ["&&"], ["==", "!="], ["<", ">", "<=", ">="], ["|"], ["^"], ["&"], ["+", "-"], ["*", "/", "%"]]

    def __init__(self, toks):
        self.t, self.i = toks, 0

    def at(self, kind, val=None):
        k, v, _ = self.t[self.i]
        return k == kind and (val is None or v == val)

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
        _, name, _ = self.eat("id")
        if name not in ("felt", "u32", "bool"):
            raise Unsupported(f"type {name}")
        return name

    def program(self):
        fns = []
        while not self.at("eof"):
            self.eat("id", "fn")
            name = self.eat("id")[1]
            self.eat("op", "(")
            params = []
            while not self.at("op", ")"):
                pn = self.eat("id")[1]
                self.eat("op", ":")
                params.append((pn, self.ty()))
                self.opt("op", ",")
            self.eat("op", ")")
            ret = None
            if self.opt("op", "->"):
                ret = self.ty()
            fns.append((name, params, ret, self.block()))
        return fns

    def block(self):
        self.eat("op", "{")
        out = []
        while not self.at("op", "}"):
            out.append(self.stmt())
        self.eat("op", "}")
        return out

    def simple(self):
        """`let x[: T] = e` or `x = e` (no trailing `;`): statement and for-header form"""
        if self.opt("id", "let"):
            name = self.eat("id")[1]
            ty = self.ty() if self.opt("op", ":") else None
            self.eat("op", "=")
            return ("let", name, ty, self.expr())
        name = self.eat("id")[1]
        self.eat("op", "=")
        return ("assign", name, self.expr())

    def stmt(self):
        if self.at("id", "let"):
            s = self.simple()
            self.eat("op", ";")
            return s
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
        if self.opt("id", "if"):
            return self.if_rest()
        if self.opt("id", "while"):
            c = self.expr()
            return ("while", c, self.block())
        if self.opt("id", "loop"):
            return ("while", ("bool", True), self.block())
        if self.opt("id", "for"):
            self.eat("op", "(")
            init = self.simple()
            self.eat("op", ";")
            cond = self.expr()
            self.eat("op", ";")
            step = self.simple()
            self.eat("op", ")")
            return ("for", init, cond, step, self.block())
        if self.at("id") and self.t[self.i + 1][:2] == ("op", "="):
            s = self.simple()
            self.eat("op", ";")
            return s
        e = self.expr()
        self.eat("op", ";")
        return ("expr", e)

    def if_rest(self):
        c = self.expr()
        then = self.block()
        els = None
        if self.opt("id", "else"):
            els = [self.if_rest()] if self.opt("id", "if") else self.block()
        return ("if", c, then, els)

    def expr(self, level=0):
        if level == len(self.PREC):
            return self.unary()
        left = self.expr(level + 1)
        while self.at("op") and self.t[self.i][1] in self.PREC[level]:
            op = self.eat("op")[1]
            left = ("bin", op, left, self.expr(level + 1))
        return left

    def unary(self):
        if self.opt("op", "-"):
            return ("neg", self.unary())
        if self.opt("op", "!"):
            return ("not", self.unary())
        if self.opt("op", "("):
            e = self.expr()
            self.eat("op", ")")
            return e
        if self.at("num"):
            _, v, suf = self.eat("num")
            return ("num", v, suf)
        name = self.eat("id")[1]
        if name in ("true", "false"):
            return ("bool", name == "true")
        if self.opt("op", "("):
            args = []
            while not self.at("op", ")"):
                args.append(self.expr())
                self.opt("op", ",")
            self.eat("op", ")")
            return ("call", name, args)
        return ("var", name)


class Ret(Exception):
    def __init__(self, v):
        self.v = v


class Brk(Exception):
    pass


class Cont(Exception):
    pass


def coerce(v, ty):
    """(type, value) -> value of type `ty`; 'lit' adopts it"""
    t, x = v
    if t == "lit":
        if ty == "u32":
            if not 0 <= x < 2**32:
                raise Unsupported("u32 literal out of range")
            return ("u32", x)
        if ty == "bool":
            raise Unsupported("integer literal used as bool")
        return ("felt", x % P)
    if ty is None or t == ty:
        return v
    raise Unsupported(f"type mismatch {t} vs {ty}")


class Interp:
    def __init__(self, src):
        self.fns = {name: (params, ret, body) for name, params, ret, body in Parser(lex(src)).program()}
        self.order = [f[0] for f in Parser(lex(src)).program()]
        self.steps = 0

    def call(self, name, args):
        if name not in self.fns:
            raise Unsupported(f"call of {name}")
        params, ret, body = self.fns[name]
        if len(params) != len(args):
            raise Unsupported("argument count")
        env = [{pn: coerce(a, pt) for (pn, pt), a in zip(params, args)}]
        try:
            self.block(body, env, ret)
        except Ret as r:
            return coerce(r.v, ret) if ret else None
        if ret:
            raise Unsupported("function falls off its end")
        return None

    def lookup(self, env, name):
        for scope in reversed(env):
            if name in scope:
                return scope
        raise Unsupported(f"unknown variable {name}")

    def block(self, stmts, env, ret):
        env.append({})
        try:
            for s in stmts:
                self.stmt(s, env, ret)
        finally:
            env.pop()

    def stmt(self, s, env, ret):
        self.steps += 1
        if self.steps > 2_000_000:
            raise Unsupported("too many steps")
        k = s[0]
        if k == "let":
            v = self.ev(s[3], env, s[2])
            env[-1][s[1]] = coerce(v, s[2] or (None if v[0] != "lit" else "felt"))
        elif k == "assign":
            scope = self.lookup(env, s[1])
            scope[s[1]] = coerce(self.ev(s[2], env, scope[s[1]][0]), scope[s[1]][0])
        elif k == "return":
            raise Ret(self.ev(s[1], env, ret))
        elif k == "break":
            raise Brk()
        elif k == "continue":
            raise Cont()
        elif k == "if":
            if self.truth(self.ev(s[1], env, None)):
                self.block(s[2], env, ret)
            elif s[3] is not None:
                self.block(s[3], env, ret)
        elif k == "while":
            while self.truth(self.ev(s[1], env, None)):
                try:
                    self.block(s[2], env, ret)
                except Brk:
                    break
                except Cont:
                    continue
        elif k == "for":
            env.append({})
            try:
                self.stmt(s[1], env, ret)
                while self.truth(self.ev(s[2], env, None)):
                    try:
                        self.block(s[4], env, ret)
                    except Brk:
                        break
                    except Cont:
                        pass
                    self.stmt(s[3], env, ret)
            finally:
                env.pop()
        elif k == "expr":
            self.ev(s[1], env, None)
        else:
            raise Unsupported(k)

    @staticmethod
    def truth(v):
        t, x = v
        if t == "bool":
            return x
        if t in ("felt", "lit"):
            return x % P != 0
        raise Unsupported("u32 used as a condition")

    def ev(self, e, env, want):
        k = e[0]
        if k == "num":
            if e[2] == "u32":
                return ("u32", e[1])
            if e[2] == "felt":
                return ("felt", e[1] % P)
            return coerce(("lit", e[1]), want) if want in ("felt", "u32") else ("lit", e[1])
        if k == "bool":
            return ("bool", e[1])
        if k == "var":
            return self.lookup(env, e[1])[e[1]]
        if k == "call":
            params = self.fns[e[1]][0] if e[1] in self.fns else []
            args = [self.ev(a, env, params[i][1] if i < len(params) else None) for i, a in enumerate(e[2])]
            return self.call(e[1], args)
        if k == "neg":
            v = self.ev(e[1], env, want)
            if v[0] == "lit":
                v = coerce(v, want or "felt")
            if v[0] == "felt":
                return ("felt", (-v[1]) % P)
            raise Unsupported("negation of a non-felt")
        if k == "not":
            v = self.ev(e[1], env, None)
            return ("bool", not self.truth(v))
        if k == "bin":
            op = e[1]
            if op in ("&&", "||"):
                a = self.truth(self.ev(e[2], env, None))
                if op == "&&":
                    return ("bool", a and self.truth(self.ev(e[3], env, None)))
                return ("bool", a or self.truth(self.ev(e[3], env, None)))
            cmp_op = op in ("==", "!=", "<", ">", "<=", ">=")
            a = self.ev(e[2], env, None if cmp_op else want)
            b = self.ev(e[3], env, None if cmp_op else want)
            if a[0] == "lit" and b[0] != "lit":
                a = coerce(a, b[0])
            elif b[0] == "lit" and a[0] != "lit":
                b = coerce(b, a[0])
            elif a[0] == "lit" and b[0] == "lit":
                t = want if want in ("felt", "u32") and not cmp_op else "felt"
                a, b = coerce(a, t), coerce(b, t)
            if a[0] != b[0]:
                raise Unsupported(f"operands of {op}: {a[0]} and {b[0]}")
            t, x, y = a[0], a[1], b[1]
            if cmp_op:
                if t == "bool" and op in ("==", "!="):
                    return ("bool", (x == y) == (op == "=="))
                if t == "felt" and op not in ("==", "!="):
                    raise Unsupported("ordering of felts")
                return ("bool", {"==": x == y, "!=": x != y, "<": x < y, ">": x > y, "<=": x <= y, ">=": x >= y}[op])
            if t == "felt":
                if op == "+": return ("felt", (x + y) % P)
                if op == "-": return ("felt", (x - y) % P)
                if op == "*": return ("felt", (x * y) % P)
                if op == "/":
                    if y % P == 0:
                        raise Unsupported("felt division by zero")
                    return ("felt", x * pow(y, P - 2, P) % P)
                raise Unsupported(f"felt {op}")
            if t == "u32":
                M = 2**32
                if op == "+": return ("u32", (x + y) % M)
                if op == "-": return ("u32", (x - y) % M)
                if op == "*": return ("u32", (x * y) % M)
                if op in ("/", "%"):
                    if y == 0:
                        raise Unsupported("u32 division by zero")
                    return ("u32", x // y if op == "/" else x % y)
                if op == "&": return ("u32", x & y)
                if op == "|": return ("u32", x | y)
                if op == "^": return ("u32", x ^ y)
            raise Unsupported(f"{t} {op}")
        raise Unsupported(k)


def to_words(v):
    """ABI slots of a returned / passed value: felt and bool one word, u32 two 16-bit limbs (low first)"""
    t, x = v
    if t == "u32":
        return [x & 0xFFFF, x >> 16]
    if t == "bool":
        return [1 if x else 0]
    return [x % P]
