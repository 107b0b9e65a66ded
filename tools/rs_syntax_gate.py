#!/usr/bin/env python3
"""A syntax gate for the Rust sources shipped without a compiler (integration/prover-hip/: no rustc / cargo in the image).
Not a Rust parser: a faithful LEXER (nested block comments, strings, raw / byte strings, chars vs lifetimes, numeric suffixes,
every punctuation token) plus the token-tree and item-level checks that catch what hand-edited, never-compiled code accumulates:
  * unterminated strings / comments, characters no Rust token starts with;
  * unbalanced or crossed delimiters (the token-tree pass rustc runs before parsing);
  * items: only `use / mod / fn / struct / enum / union / impl / trait / const / static / type / extern / macro_rules! / name!`
    (behind attributes, visibility and `unsafe / async / const / extern "C"` qualifiers) may start an item, at file level and
    inside `mod / impl / trait / extern` blocks;  `fn` needs a name, a parameter list and a body or `;`;  `struct` fields are
    `name: Type`;  `use`, `const`, `static`, `type` end in `;`;
  * a doc comment (`///`, `/** */`) must be followed by an item, a field, a variant or a statement — never by `}` or the end of
    the file (rustc: "expected item after doc comment");
  * inside bodies: `let` ends in `;` at its own nesting depth, `else` follows `}`, no `;;`-free `) {`-less `fn`.
usage: tools/rs_syntax_gate.py file.rs ...   (exit status 1 and one line per finding)"""
import re
import sys

OPEN, CLOSE = "([{", ")]}"
PAIR = {")": "(", "]": "[", "}": "{"}
PUNCT = ["<<=", ">>=", "...", "..=", "::", "->", "=>", "==", "!=", "<=", ">=", "&&", "||", "+=", "-=", "*=", "/=", "%=", "^=", "&=", "|=",
         "<<", ">>", "..", "+", "-", "*", "/", "%", "^", "!", "&", "|", "=", "<", ">", "@", ".", ",", ";", ":", "#", "$", "?", "~", "(", ")",
         "[", "]", "{", "}"]
ITEM_KW = {"use", "mod", "fn", "struct", "enum", "union", "impl", "trait", "const", "static", "type", "extern", "macro_rules"}
QUALIFIERS = {"pub", "unsafe", "async", "default"}


class GateError(Exception):
    pass


def lex(src, name):
    """-> [(kind, text, line)]; kinds: id, lifetime, num, str, char, doc, punct"""
    toks, i, line, n = [], 0, 1, len(src)

    def err(msg):
        raise GateError(f"{name}:{line}: {msg}")
    while i < n:
        c = src[i]
        if c == "\n":
            line += 1; i += 1; continue
        if c in " \t\r":
            i += 1; continue
        if src.startswith("//", i):
            j = src.find("\n", i)
            j = n if j < 0 else j
            text = src[i:j]
            if (text.startswith("///") and not text.startswith("////")) or text.startswith("//!"):
                toks.append(("doc", text[:3], line))
            i = j
            continue
        if src.startswith("/*", i):
            depth, j, start = 1, i + 2, line
            while j < n and depth:
                if src.startswith("/*", j): depth += 1; j += 2
                elif src.startswith("*/", j): depth -= 1; j += 2
                else:
                    line += src[j] == "\n"; j += 1
            if depth:
                line = start; err("unterminated block comment")
            if src.startswith("/**", i) and not src.startswith("/***", i) and j - i > 4:
                toks.append(("doc", "/**", start))
            i = j
            continue
        m = re.match(r"(b?r)(#*)\"", src[i:])
        if m:   # raw (byte) string
            close = '"' + m.group(2)
            j = src.find(close, i + m.end())
            if j < 0: err("unterminated raw string")
            line_add = src.count("\n", i, j)
            toks.append(("str", src[i:j + len(close)], line)); line += line_add; i = j + len(close)
            continue
        if c == '"' or src.startswith('b"', i):
            j = i + (2 if c == "b" else 1)
            start = line
            while j < n and src[j] != '"':
                if src[j] == "\\": j += 1
                line += j < n and src[j] == "\n"
                j += 1
            if j >= n:
                line = start; err("unterminated string literal")
            toks.append(("str", src[i:j + 1], start)); i = j + 1
            continue
        if c == "'" or src.startswith("b'", i):
            k = i + (1 if c == "b" else 0)
            m = re.match(r"'(\\(x[0-9a-fA-F]{2}|u\{[0-9a-fA-F_]{1,6}\}|.)|[^\\'\n])'", src[k:])
            if m:
                toks.append(("char", src[i:k + m.end()], line)); i = k + m.end(); continue
            m = re.match(r"'[A-Za-z_][A-Za-z0-9_]*", src[i:])
            if m and c == "'":
                toks.append(("lifetime", m.group(0), line)); i += m.end(); continue
            err("stray quote: neither a character literal nor a lifetime")
        m = re.match(r"(0x[0-9a-fA-F_]+|0b[01_]+|0o[0-7_]+|\d[\d_]*(\.\d[\d_]*)?([eE][+-]?\d+)?)([A-Za-z_][A-Za-z0-9_]*)?", src[i:])
        if m and c.isdigit():
            suf = m.group(4)
            if suf and suf not in ("u8", "u16", "u32", "u64", "u128", "usize", "i8", "i16", "i32", "i64", "i128", "isize", "f32", "f64"):
                err(f"invalid numeric suffix `{suf}`")
            text = m.group(0)
            if m.group(2) is None and src.startswith("..", i + len(m.group(1))) is False and text.endswith("."):
                text = text[:-1]
            toks.append(("num", text, line)); i += len(text)
            continue
        m = re.match(r"(r#)?[A-Za-z_][A-Za-z0-9_]*", src[i:])
        if m:
            toks.append(("id", m.group(0), line)); i += m.end(); continue
        for p in PUNCT:
            if src.startswith(p, i):
                toks.append(("punct", p, line)); i += len(p); break
        else:
            err(f"no Rust token starts with {c!r}")
    return toks


def check_delimiters(toks, name):
    stack = []
    for kind, text, line in toks:
        if kind != "punct":
            continue
        if text in OPEN:
            stack.append((text, line))
        elif text in CLOSE:
            if not stack:
                raise GateError(f"{name}:{line}: unmatched `{text}`")
            o, ol = stack.pop()
            if o != PAIR[text]:
                raise GateError(f"{name}:{line}: `{text}` closes the `{o}` opened on line {ol}")
    if stack:
        o, ol = stack[-1]
        raise GateError(f"{name}:{ol}: `{o}` is never closed")


class Items:
    """item-level grammar over the token list (bodies of functions are checked by `body`)"""
    def __init__(self, toks, name):
        self.t, self.i, self.name = toks, 0, name

    def peek(self, k=0):
        j = self.i + k
        return self.t[j] if j < len(self.t) else ("eof", "", self.t[-1][2] if self.t else 0)

    def err(self, msg, tok=None):
        tok = tok or self.peek()
        raise GateError(f"{self.name}:{tok[2]}: {msg}")

    def is_p(self, text, k=0):
        t = self.peek(k)
        return t[0] == "punct" and t[1] == text

    def is_id(self, text=None, k=0):
        t = self.peek(k)
        return t[0] == "id" and (text is None or t[1] == text)

    def eat_p(self, text):
        if not self.is_p(text):
            self.err(f"expected `{text}`, found `{self.peek()[1] or 'end of file'}`")
        self.i += 1

    def skip_group(self):
        """the current token opens a delimiter: skip to behind its partner (delimiters are balanced: checked before)"""
        depth = 0
        while True:
            k, text, _ = self.peek()
            if k == "eof":
                self.err("unexpected end of file inside a delimited group")
            if k == "punct" and text in OPEN: depth += 1
            if k == "punct" and text in CLOSE: depth -= 1
            self.i += 1
            if depth == 0:
                return

    def skip_until(self, stops):
        """advance over whole groups to the first of `stops` at depth 0; angle brackets of generics are not delimiters"""
        while True:
            k, text, _ = self.peek()
            if k == "eof":
                self.err(f"expected one of {' '.join(stops)} before the end of the file")
            if k == "punct" and text in stops:
                return text
            if k == "punct" and text in OPEN:
                self.skip_group()
            elif k == "punct" and text in CLOSE:
                self.err(f"expected one of {' '.join(stops)} before `{text}`")
            else:
                self.i += 1

    def skip_type(self, stops):
        """like skip_until, inside a TYPE: `<` / `>` nest (`HashMap<u32, Vec<T>>` holds a comma that ends no field)"""
        angle = 0
        while True:
            k, text, _ = self.peek()
            if k == "eof":
                self.err(f"expected one of {' '.join(stops)} before the end of the file")
            if k == "punct" and angle <= 0 and text in stops:
                return text
            if k == "punct" and text in OPEN:
                self.skip_group(); continue
            if k == "punct" and text in CLOSE:
                self.err(f"expected one of {' '.join(stops)} before `{text}`")
            if k == "punct" and text == "<": angle += 1
            elif k == "punct" and text == ">": angle -= 1
            elif k == "punct" and text == ">>": angle -= 2
            self.i += 1

    def attributes_and_docs(self):
        seen_doc = None
        while True:
            if self.peek()[0] == "doc":
                seen_doc = self.peek(); self.i += 1
            elif self.is_p("#"):
                self.i += 1
                if self.is_p("!"): self.i += 1
                if not self.is_p("["): self.err("`#` must start an attribute `#[...]`")
                self.skip_group()
            else:
                return seen_doc

    def items(self, closer):
        """items until `closer` ('}' or eof)"""
        while True:
            doc = self.attributes_and_docs()
            t = self.peek()
            if (closer == "eof" and t[0] == "eof") or (closer == "}" and self.is_p("}")):
                if doc and doc[1] != "//!":
                    self.err("expected an item after this doc comment", doc)
                return
            if t[0] == "eof":
                self.err("unexpected end of file: a block is not closed")
            self.item()

    def item(self):
        start = self.peek()
        while self.is_id() and self.peek()[1] in QUALIFIERS:
            self.i += 1
            if self.is_p("("):   # pub(crate)
                self.skip_group()
        if self.is_id("const") and self.is_id("fn", 1): self.i += 1
        if self.is_id("extern") and self.peek(1)[0] == "str" and (self.is_id("fn", 2)): self.i += 2
        t = self.peek()
        if t[0] != "id":
            self.err(f"expected an item, found `{t[1]}`", t)
        kw = t[1]
        if kw == "fn":
            self.i += 1
            if not self.is_id(): self.err("`fn` needs a name")
            self.i += 1
            if self.is_p("<"): self.skip_generics()
            if not self.is_p("("): self.err("`fn` needs a parameter list")
            self.skip_group()
            stop = self.skip_until(["{", ";"])
            if stop == ";": self.i += 1
            else: self.body()
        elif kw in ("struct", "union"):
            self.i += 1
            if not self.is_id(): self.err(f"`{kw}` needs a name")
            self.i += 1
            if self.is_p("<"): self.skip_generics()
            stop = self.skip_until(["{", "(", ";"])
            if stop == "{": self.fields()
            elif stop == "(":
                self.skip_group(); self.skip_until([";"]); self.i += 1
            else: self.i += 1
        elif kw == "enum":
            self.i += 1
            if not self.is_id(): self.err("`enum` needs a name")
            self.skip_until(["{"])
            self.skip_group()
        elif kw in ("impl", "trait", "mod"):
            self.i += 1
            stop = self.skip_until(["{", ";"])
            if stop == ";":
                if kw != "mod": self.err(f"`{kw}` needs a block")
                self.i += 1
            else:
                self.i += 1
                self.items("}")
                self.eat_p("}")
        elif kw == "extern":
            self.i += 1
            if self.peek()[0] == "str": self.i += 1
            if self.is_id("crate"):
                self.skip_until([";"]); self.i += 1
            else:
                self.eat_p("{"); self.items("}"); self.eat_p("}")
        elif kw in ("use", "const", "static", "type"):
            self.i += 1
            self.skip_until([";"]); self.i += 1
        elif kw == "macro_rules" or self.is_p("!", 1) or (self.is_p("::", 1)):
            # macro invocation in item position: path ! (...) ; | path ! { ... }
            while not self.is_p("!"):
                if self.peek()[0] == "eof": self.err("expected an item", start)
                self.i += 1
            self.i += 1
            if self.is_id(): self.i += 1
            if not (self.is_p("(") or self.is_p("[") or self.is_p("{")): self.err("a macro invocation needs a delimited argument")
            brace = self.is_p("{")
            self.skip_group()
            if not brace: self.eat_p(";")
        else:
            self.err(f"`{kw}` cannot start an item", t)

    def skip_generics(self):
        depth = 0
        while True:
            k, text, _ = self.peek()
            if k == "eof": self.err("unterminated generic parameter list")
            if k == "punct" and text == "<": depth += 1
            elif k == "punct" and text == ">": depth -= 1
            elif k == "punct" and text == ">>": depth -= 2
            elif k == "punct" and text in OPEN:
                self.skip_group(); continue
            self.i += 1
            if depth <= 0:
                return

    def fields(self):
        self.eat_p("{")
        while True:
            doc = self.attributes_and_docs()
            if self.is_p("}"):
                if doc: self.err("expected a field after this doc comment", doc)
                self.i += 1
                return
            if self.is_id("pub"):
                self.i += 1
                if self.is_p("("): self.skip_group()
            if not self.is_id(): self.err("expected a field name")
            self.i += 1
            self.eat_p(":")
            stop = self.skip_type([",", "}"])
            if stop == ",": self.i += 1

    def body(self):
        """a `{ ... }` block of statements: token-tree level checks only"""
        start = self.i
        self.skip_group()
        toks = self.t[start:self.i]
        depth, let_depth = 0, []
        for j, (k, text, line) in enumerate(toks):
            if k == "punct" and text in OPEN: depth += 1
            if k == "punct" and text in CLOSE:
                if let_depth and let_depth[-1][0] == depth:
                    # `let ... else { ... };` and block-valued lets close their own groups at a deeper level; a `let` whose own
                    # block closes without `;` is an error
                    raise GateError(f"{self.name}:{let_depth[-1][1]}: this `let` statement is never terminated by `;`")
                depth -= 1
                # a block inside an open `let` has just closed: what follows continues the expression (`;`, `.method()`, `?`,
                # `else`, `as`, an operator) — an identifier or another `let` starts a new statement: the `;` is missing
                if text == "}" and let_depth and let_depth[-1][0] == depth and j + 1 < len(toks):
                    nk, nt, nl = toks[j + 1]
                    if (nk == "id" and nt not in ("else", "as")) or nk in ("num", "str", "lifetime"):
                        raise GateError(f"{self.name}:{let_depth[-1][1]}: this `let` statement is not terminated by `;` (a new statement starts on line {nl})")
            if k == "id" and text == "let":
                prev = toks[j - 1] if j else None
                cond = (prev and prev[0] == "id" and prev[1] in ("if", "while")) or (prev and prev[0] == "punct" and prev[1] == "&&")
                if not cond:   # (`if let` / `while let` / let chains are expressions, not statements)
                    if let_depth and let_depth[-1][0] == depth:
                        raise GateError(f"{self.name}:{let_depth[-1][1]}: this `let` statement is not terminated by `;` before the `let` on line {line}")
                    let_depth.append((depth, line))
            if k == "punct" and text == ";" and let_depth and let_depth[-1][0] == depth:
                let_depth.pop()
            if k == "id" and text == "else":
                prev = toks[j - 1] if j else None
                if not (prev and prev[0] == "punct" and prev[1] == "}") and not any(d == depth for d, _ in let_depth):
                    raise GateError(f"{self.name}:{line}: `else` without a preceding block")
            if k == "doc" and j + 1 < len(toks) and toks[j + 1][0] == "punct" and toks[j + 1][1] == "}":
                raise GateError(f"{self.name}:{line}: expected a statement after this doc comment")


def check_file(path):
    src = open(path, encoding="utf-8").read()
    toks = lex(src, path)
    check_delimiters(toks, path)
    Items(toks, path).items("eof")
    return len(toks)


def main(argv):
    bad = 0
    for p in argv:
        try:
            n = check_file(p)
            print(f"{p}: ok ({n} tokens)")
        except GateError as e:
            print(e)
            bad = 1
    return bad


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
