"""A small evaluator of the Cairo-M SOURCE subset the compiler's codegen snapshots are written in (build container only: it reads
`/root/reference/crates/compiler/codegen/tests/snapshots/*.snap`).  It exists to compute, from the source text of a snapshot, the
value its entry function returns — independently of the CASM listing, of the VM and of every line of the prover: the expected
side of tests/golden/casm/*.json.

Subset: `fn name(p: T, ..) -> T { .. }`, `let x[: T] = e;`, assignment, `if / else if / else`, `while`, `loop`, C-style
`for (let i = a; cond; i = e) { .. }`, `break`, `continue`, `return e;`, calls, literals (`10`, `10u32`, `true`, `false`), unary
`-` `!`, binary `* / %  + -  & ^ |  == != < > <= >=  && ||`.  Types: `felt` (arithmetic mod 2^31 - 1, `/` = multiplication by the
inverse), `u32` (wrapping; `/` and `%` integer), `bool`.  An untyped integer literal takes the type of the other operand, of the
declared variable, of the parameter or of the return type; alone it is a felt.
Aggregates: `struct S { f: T, .. }` with literals `S { f: e, .. }`, tuples `(a, b)` / `(T, U)` with `let (a, (b, c)) = e;`,
fixed arrays `[T; n]` with `[a, b, c]` / `[e; n]`, `const NAME: T = e;`, heap arrays `new T[n]` behind `T*`; places
`x.f`, `x.0`, `x[i]` nest freely on both sides of `=`; `e as felt` on a u32 (the value must fit below 2^31 - 1), `assert(e);`.
Structs and tuples are values (copied by `let`, `=` and calls); arrays and pointers are references to their storage, as in the
compiled code.  Anything else raises Unsupported: the fixture generator then leaves that snapshot out."""
import re

P = 2**31 - 1


class Unsupported(Exception):
    pass


class Fault(Unsupported):
    """the PROGRAM fails on these inputs (division by zero, a cast that does not fit, a failing assertion): not an expected value"""


TOKEN = re.compile(r"\s+|//[^\n]*|(?P<num>\d+)(?P<suf>u32|felt)?|(?P<id>[A-Za-z_]\w*)|(?P<op>->|==|!=|<=|>=|&&|\|\||[-+*/%&|^!<>=(){};:,.\[\]])")


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
        self.struct_names = {self.t[k + 1][1] for k in range(len(self.t) - 1) if self.t[k][:2] == ("id", "struct")}

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
        """felt | u32 | bool | Struct | (T, ..) | [T; n], each optionally followed by `*`"""
        if self.opt("op", "("):
            items = []
            while not self.at("op", ")"):
                items.append(self.ty())
                self.opt("op", ",")
            self.eat("op", ")")
            t = items[0] if len(items) == 1 else ("tuple", tuple(items))
        elif self.opt("op", "["):
            el = self.ty()
            self.eat("op", ";")
            n = self.eat("num")[1]
            self.eat("op", "]")
            t = ("array", el, n)
        else:
            _, name, _ = self.eat("id")
            if name in ("felt", "u32", "bool"):
                t = name
            elif name in self.struct_names:
                t = ("struct", name)
            else:
                raise Unsupported(f"type {name}")
        while self.opt("op", "*"):
            t = ("ptr", t)
        return t

    def program(self):
        fns, structs, consts = [], {}, []
        while not self.at("eof"):
            if self.opt("id", "struct"):
                name = self.eat("id")[1]
                self.eat("op", "{")
                fields = []
                while not self.at("op", "}"):
                    fn_ = self.eat("id")[1]
                    self.eat("op", ":")
                    fields.append((fn_, self.ty()))
                    self.opt("op", ",")
                self.eat("op", "}")
                structs[name] = fields
                continue
            if self.opt("id", "const"):
                name = self.eat("id")[1]
                ty = self.ty() if self.opt("op", ":") else None
                self.eat("op", "=")
                consts.append((name, ty, self.expr()))
                self.eat("op", ";")
                continue
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
        return fns, structs, consts

    def block(self):
        self.eat("op", "{")
        out = []
        while not self.at("op", "}"):
            out.append(self.stmt())
        self.eat("op", "}")
        return out

    def pattern(self):
        """x | (p, p, ..)"""
        if self.opt("op", "("):
            items = []
            while not self.at("op", ")"):
                items.append(self.pattern())
                self.opt("op", ",")
            self.eat("op", ")")
            return ("ptuple", items)
        return ("pvar", self.eat("id")[1])

    def simple(self):
        """`let pat[: T] = e` or `place = e` (no trailing `;`): statement and for-header form"""
        if self.opt("id", "let"):
            pat = self.pattern()
            ty = self.ty() if self.opt("op", ":") else None
            self.eat("op", "=")
            return ("let", pat, ty, self.expr())
        place = self.postfix()
        self.eat("op", "=")
        return ("assign", place, self.expr())

    def is_assignment(self):
        """a place followed by `=` (not `==`) before the `;` of the statement"""
        k, depth = self.i, 0
        while k < len(self.t):
            kind, v, _ = self.t[k]
            if kind == "op" and v in "([":
                depth += 1
            elif kind == "op" and v in ")]":
                depth -= 1
            elif kind == "op" and v == "=" and depth == 0:
                return True
            elif (kind == "op" and v in (";", "{", "}")) or kind == "eof":
                return False
            elif kind == "op" and depth == 0 and v != ".":
                return False
            k += 1
        return False

    def stmt(self):
        if self.at("id", "let"):
            s = self.simple()
            self.eat("op", ";")
            return s
        if self.opt("id", "return"):
            e = None
            if self.at("op", "(") and self.at("op", ")", 1):   # `return();`
                self.i += 2
            elif not self.at("op", ";"):
                e = self.expr()
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
        if self.at("id", "assert") and self.at("op", "(", 1):
            self.i += 1
            e = self.expr()
            self.eat("op", ";")
            return ("assert", e)
        if self.at("id") and self.is_assignment():
            s = self.simple()
            self.eat("op", ";")
            return s
        e = self.expr()
        self.eat("op", ";")
        return ("expr", e)

    def if_rest(self):
        c = self.expr(no_struct=True)
        then = self.block()
        els = None
        if self.opt("id", "else"):
            els = [self.if_rest()] if self.opt("id", "if") else self.block()
        return ("if", c, then, els)

    def expr(self, level=0, no_struct=False):
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
            if self.opt("op", "."):
                if self.at("num"):
                    e = ("field", e, self.eat("num")[1])
                else:
                    e = ("field", e, self.eat("id")[1])
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
            if len(items) == 1 and not trailing:
                return items[0]
            return ("tuple", items)
        if self.opt("op", "["):
            first = self.expr()
            if self.opt("op", ";"):
                n = self.eat("num")[1]
                self.eat("op", "]")
                return ("repeat", first, n)
            items = [first]
            while self.opt("op", ","):
                if self.at("op", "]"):
                    break
                items.append(self.expr())
            self.eat("op", "]")
            return ("array", items)
        if self.at("num"):
            _, v, suf = self.eat("num")
            return ("num", v, suf)
        name = self.eat("id")[1]
        if name in ("true", "false"):
            return ("bool", name == "true")
        if name == "new":
            el = self.ty_no_ptr()
            self.eat("op", "[")
            n = self.expr()
            self.eat("op", "]")
            return ("new", el, n)
        if self.opt("op", "("):
            args = []
            while not self.at("op", ")"):
                args.append(self.expr())
                self.opt("op", ",")
            self.eat("op", ")")
            return ("call", name, args)
        if name in self.struct_names and not no_struct and self.at("op", "{"):
            self.eat("op", "{")
            fields = []
            while not self.at("op", "}"):
                fn_ = self.eat("id")[1]
                self.eat("op", ":")
                fields.append((fn_, self.expr()))
                self.opt("op", ",")
            self.eat("op", "}")
            return ("slit", name, fields)
        return ("var", name)

    def ty_no_ptr(self):
        _, name, _ = self.eat("id")
        if name in ("felt", "u32", "bool"):
            return name
        if name in self.struct_names:
            return ("struct", name)
        raise Unsupported(f"type {name}")


class Ret(Exception):
    def __init__(self, v):
        self.v = v


class Brk(Exception):
    pass


class Cont(Exception):
    pass


class Ref:
    """storage of an array or of a heap allocation: shared by every copy of the array value / pointer"""
    def __init__(self, items, el, grow):
        self.items, self.el, self.grow = items, el, grow


def copy_val(v):
    """structs and tuples are values; arrays / pointers keep pointing at their storage"""
    if v[0] == "tuple":
        return ("tuple", [copy_val(x) for x in v[1]])
    if v[0] == "struct":
        return ("struct", v[1], {k: copy_val(x) for k, x in v[2].items()})
    return v


class Interp:
    def __init__(self, src):
        fns, self.structs, consts = Parser(lex(src)).program()
        self.fns = {name: (params, ret, body) for name, params, ret, body in fns}
        self.order = [f[0] for f in fns]
        self.steps = 0
        self.globals = {}
        for name, ty, e in consts:
            self.globals[name] = self.coerce(self.ev(e, [self.globals], ty), ty)

    # ---- types -------------------------------------------------------------------------------------------------------------
    def coerce(self, v, ty):
        """value -> value of type `ty` (None: leave; untyped literals become felts); literals inside aggregates adopt the
        element / field types"""
        t = v[0]
        if t == "lit":
            x = v[1]
            if ty == "u32":
                if not 0 <= x < 2**32:
                    raise Unsupported("u32 literal out of range")
                return ("u32", x)
            if ty in (None, "felt"):
                return ("felt", x % P)
            raise Unsupported(f"integer literal used as {ty}")
        if t == "tuple":
            if ty is None:
                return ("tuple", [self.coerce(x, None) for x in v[1]])
            if not (isinstance(ty, tuple) and ty[0] == "tuple" and len(ty[1]) == len(v[1])):
                raise Unsupported(f"tuple against {ty}")
            return ("tuple", [self.coerce(x, et) for x, et in zip(v[1], ty[1])])
        if t == "array":
            ref = v[1]
            if isinstance(ty, tuple) and ty[0] == "ptr":
                raise Unsupported("array used as a pointer")
            if ty is not None and not (isinstance(ty, tuple) and ty[0] == "array" and ty[2] == len(ref.items)):
                raise Unsupported(f"array against {ty}")
            el = ty[1] if ty is not None else ref.el
            ref.items[:] = [self.coerce(x, el) for x in ref.items]
            ref.el = el
            return v
        if t == "ptr":
            if ty is not None and not (isinstance(ty, tuple) and ty[0] == "ptr"):
                raise Unsupported(f"pointer against {ty}")
            return v
        if t == "struct":
            if ty is not None and ty != ("struct", v[1]):
                raise Unsupported(f"struct {v[1]} against {ty}")
            return v
        if ty is None or t == ty:
            return v
        raise Unsupported(f"type mismatch {t} vs {ty}")

    def type_of(self, v):
        t = v[0]
        if t in ("felt", "u32", "bool"):
            return t
        if t == "lit":
            return None
        if t == "tuple":
            return ("tuple", tuple(self.type_of(x) for x in v[1]))
        if t == "struct":
            return ("struct", v[1])
        if t == "array":
            return ("array", v[1].el, len(v[1].items))
        if t == "ptr":
            return ("ptr", v[1].el)
        raise Unsupported(t)

    def zero(self, ty):
        if ty == "felt":
            return ("felt", 0)
        if ty == "u32":
            return ("u32", 0)
        if ty == "bool":
            return ("bool", False)
        if ty[0] == "struct":
            return ("struct", ty[1], {f: self.zero(ft) for f, ft in self.structs[ty[1]]})
        if ty[0] == "tuple":
            return ("tuple", [self.zero(t) for t in ty[1]])
        raise Unsupported(f"zero of {ty}")

    # ---- calls / statements ------------------------------------------------------------------------------------------------
    def call(self, name, args):
        if name not in self.fns:
            raise Unsupported(f"call of {name}")
        params, ret, body = self.fns[name]
        if len(params) != len(args):
            raise Unsupported("argument count")
        env = [self.globals, {pn: copy_val(self.coerce(a, pt)) for (pn, pt), a in zip(params, args)}]
        try:
            self.block(body, env, ret)
        except Ret as r:
            if ret is None:
                if r.v is not None:
                    raise Unsupported("value returned from a function without a return type")
                return None
            return copy_val(self.coerce(r.v, ret))
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

    def bind(self, pat, v, env):
        if pat[0] == "pvar":
            env[-1][pat[1]] = v
            return
        if v[0] != "tuple" or len(v[1]) != len(pat[1]):
            raise Unsupported("tuple pattern against a non-tuple")
        for p_, x in zip(pat[1], v[1]):
            self.bind(p_, x, env)

    def place(self, e, env):
        """-> (container, key) such that container[key] is the value the place names"""
        if e[0] == "var":
            scope = self.lookup(env, e[1])
            if scope is self.globals:
                raise Unsupported("assignment to a constant")
            return scope, e[1]
        if e[0] == "field":
            c, k = self.place(e[1], env)
            base = c[k]
            if base[0] == "tuple" and isinstance(e[2], int):
                return base[1], e[2]
            if base[0] == "struct" and e[2] in base[2]:
                return base[2], e[2]
            raise Unsupported(f"field {e[2]} of {base[0]}")
        if e[0] == "index":
            base = self.ev(e[1], env, None)
            return self.slot(base, self.ev(e[2], env, "felt")), None
        raise Unsupported(f"assignment to {e[0]}")

    def slot(self, base, ix):
        if base[0] not in ("array", "ptr"):
            raise Unsupported(f"index into {base[0]}")
        ix = self.coerce(ix, "felt") if ix[0] == "lit" else ix
        if ix[0] != "felt":
            raise Unsupported("index that is not a felt")
        ref, i = base[1], ix[1]
        if i >= len(ref.items):
            if not ref.grow or i >= 1 << 16:
                raise Unsupported("index out of bounds")
            ref.items.extend(None for _ in range(i + 1 - len(ref.items)))
        return _Slot(ref, i)

    def stmt(self, s, env, ret):
        self.steps += 1
        if self.steps > 2_000_000:
            raise Unsupported("too many steps")
        k = s[0]
        if k == "let":
            v = self.ev(s[3], env, s[2])
            self.bind(s[1], copy_val(self.coerce(v, s[2])), env)
        elif k == "assign":
            c, key = self.place(s[1], env)
            if isinstance(c, _Slot):
                v = copy_val(self.coerce(self.ev(s[2], env, c.ref.el), c.ref.el))
                c.ref.items[c.i] = v
            else:
                ty = self.type_of(c[key])
                c[key] = copy_val(self.coerce(self.ev(s[2], env, ty), ty))
        elif k == "return":
            raise Ret(None if s[1] is None else self.ev(s[1], env, ret))
        elif k == "break":
            raise Brk()
        elif k == "continue":
            raise Cont()
        elif k == "assert":
            if not self.truth(self.ev(s[1], env, None)):
                raise Fault("assertion fails")
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
        t, x = v[0], v[1]
        if t == "bool":
            return x
        if t in ("felt", "lit"):
            return x % P != 0
        raise Unsupported(f"{t} used as a condition")

    def ev(self, e, env, want):
        k = e[0]
        if k == "num":
            if e[2] == "u32":
                return ("u32", e[1])
            if e[2] == "felt":
                return ("felt", e[1] % P)
            return self.coerce(("lit", e[1]), want) if want in ("felt", "u32") else ("lit", e[1])
        if k == "bool":
            return ("bool", e[1])
        if k == "var":
            return self.lookup(env, e[1])[e[1]]
        if k == "tuple":
            wt = want[1] if isinstance(want, tuple) and want[0] == "tuple" and len(want[1]) == len(e[1]) else [None] * len(e[1])
            return ("tuple", [self.ev(x, env, t) for x, t in zip(e[1], wt)])
        if k == "array":
            el = want[1] if isinstance(want, tuple) and want[0] == "array" else None
            items = [copy_val(self.ev(x, env, el)) for x in e[1]]
            if el is None:
                items = [self.coerce(x, None) for x in items]
                el = self.type_of(items[0])
            return ("array", Ref([self.coerce(x, el) for x in items], el, False))
        if k == "repeat":
            el = want[1] if isinstance(want, tuple) and want[0] == "array" else None
            v = self.coerce(self.ev(e[1], env, el), el)
            return ("array", Ref([copy_val(v) for _ in range(e[2])], self.type_of(v), False))
        if k == "new":
            n = self.ev(e[2], env, "felt")
            return ("ptr", Ref([], e[1], True))
        if k == "slit":
            if e[1] not in self.structs:
                raise Unsupported(f"struct {e[1]}")
            decl = dict(self.structs[e[1]])
            if set(decl) != {f for f, _ in e[2]}:
                raise Unsupported(f"fields of {e[1]}")
            given = {f: copy_val(self.coerce(self.ev(x, env, decl[f]), decl[f])) for f, x in e[2]}   # evaluated in literal order
            return ("struct", e[1], {f: given[f] for f, _ in self.structs[e[1]]})                    # kept in declared order
        if k == "field":
            base = self.ev(e[1], env, None)
            if base[0] == "tuple" and isinstance(e[2], int) and e[2] < len(base[1]):
                return base[1][e[2]]
            if base[0] == "struct" and e[2] in base[2]:
                return base[2][e[2]]
            raise Unsupported(f"field {e[2]} of {base[0]}")
        if k == "index":
            sl = self.slot(self.ev(e[1], env, None), self.ev(e[2], env, "felt"))
            v = sl.ref.items[sl.i]
            if v is None:
                raise Unsupported("read of a heap cell that was never written")
            return v
        if k == "cast":
            v = self.ev(e[1], env, None)
            if e[2] == "felt" and v[0] == "u32":
                if v[1] >= P:
                    raise Fault("cast: the u32 does not fit in a felt")
                return ("felt", v[1])
            if e[2] == "felt" and v[0] in ("felt", "lit"):
                return self.coerce(v, "felt")
            raise Unsupported(f"cast of {v[0]} to {e[2]}")
        if k == "call":
            params = self.fns[e[1]][0] if e[1] in self.fns else []
            args = [self.ev(a, env, params[i][1] if i < len(params) else None) for i, a in enumerate(e[2])]
            return self.call(e[1], args)
        if k == "neg":
            v = self.ev(e[1], env, want)
            if v[0] == "lit":
                v = self.coerce(v, want if want in ("felt", "u32") else "felt")
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
            want = want if want in ("felt", "u32") else None
            a = self.ev(e[2], env, None if cmp_op else want)
            b = self.ev(e[3], env, None if cmp_op else want)
            if a[0] == "lit" and b[0] != "lit":
                a = self.coerce(a, b[0])
            elif b[0] == "lit" and a[0] != "lit":
                b = self.coerce(b, a[0])
            elif a[0] == "lit" and b[0] == "lit":
                t = want if want in ("felt", "u32") and not cmp_op else "felt"
                a, b = self.coerce(a, t), self.coerce(b, t)
            if a[0] != b[0]:
                raise Unsupported(f"operands of {op}: {a[0]} and {b[0]}")
            t, x, y = a[0], a[1], b[1]
            if cmp_op:
                if t == "bool" and op in ("==", "!="):
                    return ("bool", (x == y) == (op == "=="))
                if t == "felt" and op not in ("==", "!="):
                    raise Unsupported("ordering of felts")
                if t not in ("felt", "u32"):
                    raise Unsupported(f"comparison of {t}")
                return ("bool", {"==": x == y, "!=": x != y, "<": x < y, ">": x > y, "<=": x <= y, ">=": x >= y}[op])
            if t == "felt":
                if op == "+": return ("felt", (x + y) % P)
                if op == "-": return ("felt", (x - y) % P)
                if op == "*": return ("felt", (x * y) % P)
                if op == "/":
                    if y % P == 0:
                        raise Fault("felt division by zero")
                    return ("felt", x * pow(y, P - 2, P) % P)
                raise Unsupported(f"felt {op}")
            if t == "u32":
                M = 2**32
                if op == "+": return ("u32", (x + y) % M)
                if op == "-": return ("u32", (x - y) % M)
                if op == "*": return ("u32", (x * y) % M)
                if op in ("/", "%"):
                    if y == 0:
                        raise Fault("u32 division by zero")
                    return ("u32", x // y if op == "/" else x % y)
                if op == "&": return ("u32", x & y)
                if op == "|": return ("u32", x | y)
                if op == "^": return ("u32", x ^ y)
            raise Unsupported(f"{t} {op}")
        raise Unsupported(k)


class _Slot:
    def __init__(self, ref, i):
        self.ref, self.i = ref, i

    def __getitem__(self, key):   # Interp.place returns (container, key): a slot is its own container under key None
        return self.ref.items[self.i]

    def __setitem__(self, key, v):
        self.ref.items[self.i] = v


def coerce(v, ty):
    """(module-level form kept for the fixture script) scalar coercion"""
    return Interp.coerce(None, v, ty)


def to_words(v):
    """ABI slots of a returned / passed value: felt and bool one word, u32 two 16-bit limbs (low first); tuples and structs are
    their members' slots in order"""
    if v is None:
        return []
    t = v[0]
    if t == "u32":
        return [v[1] & 0xFFFF, v[1] >> 16]
    if t == "bool":
        return [1 if v[1] else 0]
    if t == "felt":
        return [v[1] % P]
    if t == "tuple":
        return [w for x in v[1] for w in to_words(x)]
    if t == "struct":
        return [w for x in v[2].values() for w in to_words(x)]
    raise Unsupported(f"{t} in the entry function's signature")
