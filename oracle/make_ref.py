#!/usr/bin/env python3
"""oracle/make_ref.py -- builds oracle/_ref: THE REFERENCE ITSELF, compiled here.

Test infrastructure (never imported by the product package).  The reference engine
(`/root/reference/src/K4os.Compression.LZ4/Engine/**`, `Internal/Mem*.cs`) is `unsafe` C#
pointer code -- C in all but spelling.  This script reads those files WHERE THEY LIE (nothing
is copied into the repository: oracle/Makefile has the generated C++ written into a scratch directory that
lasts as long as the g++ run that reads it; only the compiled libraries and the report stay, under the git-ignored `oracle/_ref/`),
rewrites the spelling with purely syntactic, table-driven rules, and leaves every statement
of every engine function exactly as the reference has it.  What the rules cannot express
(calls into the .NET runtime) is supplied by `oracle/ref_prelude.hpp` -- a fixed list of
one-to-three-line members, each citing the reference member it stands in for -- and named in
EXCLUDED below; the generated header repeats the list.

Rules (R1..R20), all token-level:
  R1  `#define` / `#if` / `#elif` / `#else` / `#endif` evaluated (BIT32 per file, NET5_0_OR_GREATER
      and LZ4_FAST_DEC_LOOP from the command line); `#region`, `#pragma`, `#nullable` dropped
  R2  `using X;` dropped; `using A = B.C.D;` -> `using A = <D or the C++ name of System.*>;`
      emitted at the top of the class the file contributes to
  R3  `namespace N;` / `namespace N { }` dropped (everything lands in `namespace k4ref`)
  R4  `partial class C[: B]` bodies of all files concatenated into ONE `struct C [: B]`
  R5  attributes `[...]` in front of members and parameters dropped
  R6  access / CLR modifiers dropped (`public protected private internal unsafe new readonly
      sealed`); `static` kept; member `const T x = e` -> `static constexpr T x = e`;
      `static [readonly] T x = e` -> `static inline T x = e`; `T[] x = {..}` -> `T x[] = {..}`
  R7  `enum E { }` -> `enum class E { };`   `struct S { }` -> `struct S { };` (members by the same
      rules; `fixed T a[N]` -> `T a[N]`; a struct with a constructor also gets `S() = default;`)
  R8  `T f(args) => e;` -> `T f(args) { return e; }` (`{ e; }` when T is void)
  R9  `ref T` return / `ref|out T p` parameter -> `T&`; `ref e` in an expression -> `e`
  R10 `var` -> `auto`;  `null` -> `nullptr`;  `@name` -> `name_`;  `this.` -> `this->`
  R11 `unchecked(e)` -> `(e)`
  R12 `X.y` -> `X::y` whenever X is a class / struct / enum / alias name declared in the inputs
  R13 `new S(args)` -> `S(args)` for a struct S declared in the inputs
  R14 `stackalloc T[n]` -> `(T*) alloca(sizeof(T) * (n))`
  R15 `sizeof(T)` -> `((int) sizeof(T))` (C# `sizeof` is an `int`)
  R16 `try { A } finally { B }` -> `{ K4RefFinally _fin([&]() { B }); A }`
  R17 `f(out var x, ...)` -> `<OUT_TYPES[f]> x; f(x, ...)`
  R18 `p -= q` with both sides the two pointer FIELDS named in PTR_DIFF_ASSIGN -> `p = (byte*) (p - q)`
      (C# lets `byte* -= byte*` through its compound-assignment conversion; C++ does not)
  R19 an identifier that is a C++ keyword gets a trailing underscore
  R20 an instance method of a class with only static state -> `static` is NOT added; left as is
  R21 the nested `enum`s of a class are emitted before its other members (declaration order only)
  R22 a switch EXPRESSION `s switch { p1 => e1, p2 => e2, ... }` (s an identifier or a parenthesised expression) ->
      `([&](auto _v) { if (c1) return e1; if (c2) return e2; ... })(s)`; patterns: a constant (`_v == k`), a relational
      pattern (`> k`: `_v > k`), `or` / `and` of those, the discard `_`, `var x` (`auto x = _v;`); an arm `=> throw e`
      stays a `throw`
  R23 an interpolated string `$"...{x}..."` -> the plain literal `"...{x}..."` (exception messages: text only, no behaviour)
  R24 of a class listed in ONLY just the named members are taken (the others are managed-array code: Span, ArrayPool,
      IBufferWriter); a top-level attribute is dropped and a top-level struct named in TOPLEVEL_SUPPLIED is left to
      oracle/ref_prelude.hpp

What is NOT C# any more after this: integer promotion.  C# computes `uint (+|-|*) int` in
`long`; C++ computes it in `unsigned`.  The original C (lz4 1.9.2) these files were ported
from has the C++ behaviour; every site is listed by `make -C oracle ref-signcheck`
(`-Wsign-compare -Wsign-conversion`) and the audit is in DESIGN.md section 3.  Signed
overflow wraps in both (`-fwrapv`).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import re
import sys

SRC_ROOT = "/root/reference/src/K4os.Compression.LZ4"

# (class, files in concatenation order).  Types first: a nested struct's array bound must have
# seen its constant.
INPUTS = [
    ("Mem", ["Internal/Mem.cs"]),
    ("Mem64", ["Internal/x64/Mem64.cs"]),
    ("Mem32", ["Internal/x32/Mem32.cs"]),
    ("LL", ["Engine/LL.types.cs", "Engine/LL.types.high.cs", "Engine/LL.tools.cs", "Engine/LL.high.cs"]),
    ("LL64", ["Engine/x64/LL64.tools.cs", "Engine/x64/LL64.fast.cs", "Engine/x64/LL64.dec.cs", "Engine/x64/LL64.high.cs"]),
    ("LL32", ["Engine/x32/LL32.tools.cs", "Engine/x32/LL32.fast.cs", "Engine/x32/LL32.dec.cs", "Engine/x32/LL32.high.cs"]),
    # the envelope's header arithmetic (R24: the named members only)
    ("LZ4Pickler", ["LZ4Pickler.pickle.cs", "LZ4Pickler.unpickle.cs"]),
]

# R24: classes of which only these members are taken -- LZ4Pickler's header helpers (LZ4Pickler.pickle.cs:161-229,
# LZ4Pickler.unpickle.cs:131-148); everything else in those files is Span / ArrayPool / IBufferWriter code
ONLY = {
    "LZ4Pickler": {"MAX_STACKALLOC", "VersionMask", "GetPessimisticHeaderSize", "GetUncompressedHeaderSize", "GetCompressedHeaderSize",
                   "EncodeUncompressedHeader", "EncodeUncompressedHeaderV0", "EncodeCompressedHeader", "EncodeCompressedHeaderV0",
                   "EncodeHeaderByteV0", "EffectiveSizeOf", "EncodeSizeOf", "DecodeHeader", "DecodeHeaderV0"},
}
TOPLEVEL_SUPPLIED = {"PickleHeader"}       # LZ4Pickler.unpickle.cs:163-181: auto-properties; three fields and the constructor in the prelude

# members the rules cannot express: bodies are .NET runtime calls (Unsafe.*, Marshal.*, Buffer.*,
# properties, generics with managed arrays).  oracle/ref_prelude.hpp supplies each one.
EXCLUDED = {
    ("Mem", "Empty"): "managed byte[] (Array.Empty<byte>())",
    ("Mem", "System32"): "property syntax; `sizeof(void*) < sizeof(ulong)`",
    ("Mem", "CpBlk"): "Unsafe.CopyBlockUnaligned",
    ("Mem", "ZBlk"): "Unsafe.InitBlockUnaligned",
    ("Mem", "Move"): "Buffer.MemoryCopy",
    ("Mem", "Alloc"): "Marshal.AllocHGlobal",
    ("Mem", "Free"): "Marshal.FreeHGlobal",
    ("Mem", "CloneArray"): "generic over a managed T[] with `fixed` / `ref`",
    ("LL", "Assert"): "[Conditional(\"DEBUG\")] + CallerArgumentExpression + string",
    ("LL", "Enforce32"): "auto-property (process-wide switch; the entry points take the engine explicitly)",
    ("LL", "Algorithm"): "property returning the managed enum Engine/Algorithm.cs",
    ("LZ4Pickler", "PokeN"): "Unsafe.CopyBlockUnaligned on `ref target[0]`",
    ("LZ4Pickler", "PeekN"): "`fixed` + Unsafe.CopyBlockUnaligned",
    ("LZ4Pickler", "UnexpectedVersion"): "new ArgumentException (managed exception object)",
    ("LZ4Pickler", "CorruptedPickle"): "new InvalidDataException (managed exception object)",
}

OUT_TYPES = {"PinnedMemory.Alloc": "PinnedMemory"}          # R17
PTR_DIFF_ASSIGN = {("end", "base_")}                          # R18: LL.high.cs `LZ4_streamHCPtr->end -= LZ4_streamHCPtr->@base`
SYSTEM_TYPES = {"System.UInt32": "uint32_t", "System.UInt64": "uint64_t", "System.Int32": "int32_t", "System.Int64": "int64_t"}
MODIFIERS_DROP = {"public", "protected", "private", "internal", "unsafe", "new", "readonly", "sealed", "partial", "override", "virtual"}
CPP_KEYWORDS = {
    "register", "delete", "template", "typename", "union", "and", "or", "not", "xor", "signed", "unsigned", "inline",
    "friend", "mutable", "export", "asm", "bitand", "bitor", "compl", "and_eq", "or_eq", "xor_eq", "not_eq",
    "typedef", "extern", "wchar_t", "near", "far", "errno", "min", "max"}
EXTERNAL_TYPES = {"PinnedMemory", "BitOperations", "Debug"}  # prelude structs reached with `.`

TOKEN_RE = re.compile(r"""
  (?P<ws>\s+)
 |(?P<lc>//[^\n]*)
 |(?P<bc>/\*.*?\*/)
 |(?P<str>\$?@?"(?:\\.|[^"\\])*")
 |(?P<chr>'(?:\\.|[^'\\])+')
 |(?P<num>0[xX][0-9a-fA-F_]+[uUlL]*|\d[\d_]*(?:\.\d+)?(?:[eE][+-]?\d+)?[uUlLfFdDmM]*)
 |(?P<id>@?[A-Za-z_]\w*)
 |(?P<op>=>|->|\+\+|--|<<=|>>=|<<|>>|<=|>=|==|!=|&&|\|\||\+=|-=|\*=|/=|%=|&=|\|=|\^=|\?\?|::|[{}()\[\];,.<>+\-*/%&|^!~?:=])
""", re.S | re.X)


class Tok:
    __slots__ = ("kind", "text", "line")

    def __init__(self, kind, text, line):
        self.kind, self.text, self.line = kind, text, line

    @property
    def sig(self):
        return self.kind not in ("ws", "lc", "bc")

    def __repr__(self):
        return f"{self.kind}:{self.text!r}@{self.line}"


class RuleCount(dict):
    def hit(self, rule, n=1):
        self[rule] = self.get(rule, 0) + n


def preprocess(text: str, defines: set[str], counts: RuleCount) -> str:
    """R1.  Line numbers are preserved (dropped lines become empty)."""
    if text.startswith("﻿"):
        text = text[1:]
    out = []
    stack = []      # (parent_active, this_branch_taken_already, currently_active)
    active = True
    defines = set(defines)

    def ev(expr: str) -> bool:
        expr = re.sub(r"//.*", "", expr)
        py = re.sub(r"[A-Za-z_]\w*", lambda m: "True" if m.group(0) in defines or m.group(0) == "true" else "False", expr)
        py = py.replace("&&", " and ").replace("||", " or ").replace("!", " not ")
        return bool(eval(py, {}, {}))

    for line in text.split("\n"):
        s = line.strip()
        if s.startswith("#"):
            counts.hit("R1")
            d = s[1:].strip()
            if d.startswith("define"):
                if active:
                    defines.add(d.split()[1])
            elif d.startswith("if"):
                stack.append((active, False, active))
                taken = active and ev(d[2:])
                stack[-1] = (stack[-1][0], taken, taken)
                active = taken
            elif d.startswith("elif"):
                parent, taken, _ = stack[-1]
                now = parent and not taken and ev(d[4:])
                stack[-1] = (parent, taken or now, now)
                active = now
            elif d.startswith("else"):
                parent, taken, _ = stack[-1]
                now = parent and not taken
                stack[-1] = (parent, True, now)
                active = now
            elif d.startswith("endif"):
                parent, _, _ = stack.pop()
                active = parent
            elif d.split()[0] in ("region", "endregion", "pragma", "nullable"):
                pass
            else:
                raise SystemExit(f"unknown directive: {s}")
            out.append("")
        else:
            out.append(line if active else "")
    assert not stack
    return "\n".join(out)


def tokenize(text: str) -> list[Tok]:
    toks, pos, line = [], 0, 1
    while pos < len(text):
        m = TOKEN_RE.match(text, pos)
        if not m:
            raise SystemExit(f"cannot tokenize at line {line}: {text[pos:pos + 40]!r}")
        kind = m.lastgroup
        toks.append(Tok(kind, m.group(0), line))
        line += m.group(0).count("\n")
        pos = m.end()
    return toks


def match_close(toks, i, open_, close):
    """index of the token closing the bracket opened at i"""
    depth = 0
    for j in range(i, len(toks)):
        t = toks[j]
        if t.kind == "op":
            if t.text == open_:
                depth += 1
            elif t.text == close:
                depth -= 1
                if depth == 0:
                    return j
    raise SystemExit(f"unbalanced {open_} at line {toks[i].line}")


def next_sig(toks, i):
    while i < len(toks) and not toks[i].sig:
        i += 1
    return i


def prev_sig(toks, i):
    while i >= 0 and not toks[i].sig:
        i -= 1
    return i


class Translator:
    def __init__(self, defines):
        self.defines = set(defines)
        self.counts = RuleCount()
        self.classes = {}       # name -> dict(base, aliases(ordered), members[text], files)
        self.type_names = set(EXTERNAL_TYPES)
        self.struct_names = set()
        self.excluded_seen = []
        self.supplied_seen = []
        self.not_taken = []
        self.parsed = []        # (cls, path, tokens of the class body, aliases)

    # ---------------------------------------------------------------- file level
    def load(self, cls, rel):
        path = os.path.join(SRC_ROOT, rel)
        text = preprocess(open(path, encoding="utf-8").read(), self.defines, self.counts)
        toks = tokenize(text)
        aliases = []
        i = 0
        n = len(toks)
        body = None
        while i < n:
            i = next_sig(toks, i)
            if i >= n:
                break
            t = toks[i]
            if t.text == "using":
                j = i
                while toks[j].text != ";":
                    j += 1
                stmt = "".join(x.text for x in toks[i + 1:j] if x.sig)
                if "=" in stmt:
                    a, b = stmt.split("=", 1)
                    aliases.append((a, SYSTEM_TYPES.get(b, b.split(".")[-1])))
                self.counts.hit("R2")
                i = j + 1
            elif t.text == "namespace":
                j = i
                while toks[j].text not in (";", "{"):
                    j += 1
                self.counts.hit("R3")
                i = j + 1      # a block namespace's closing brace is dropped below (trailing `}`)
            elif t.text in MODIFIERS_DROP or t.text in ("static", "abstract"):
                i += 1
            elif t.text == "class":
                name = toks[next_sig(toks, i + 1)].text
                j = i
                while toks[j].text != "{":
                    j += 1
                hdr = [x.text for x in toks[i:j] if x.sig]
                base = hdr[hdr.index(":") + 1] if ":" in hdr else None
                k = match_close(toks, j, "{", "}")
                assert name == cls, (name, cls)
                body = toks[j + 1:k]
                c = self.classes.setdefault(cls, dict(base=None, aliases=[], members=[], files=[]))
                if base:
                    c["base"] = base
                c["files"].append(rel)
                for a in aliases:
                    if a not in c["aliases"]:
                        c["aliases"].append(a)
                    self.type_names.add(a[0])
                self.type_names.add(cls)
                self.counts.hit("R4")
                i = k + 1
            elif t.text == "}":
                i += 1
            elif t.text == "[":                        # R24: a top-level attribute
                i = match_close(toks, i, "[", "]") + 1
                self.counts.hit("R5")
            elif t.text == "struct" and toks[next_sig(toks, i + 1)].text in TOPLEVEL_SUPPLIED:
                nm = toks[next_sig(toks, i + 1)].text
                j = i
                while toks[j].text != "{":
                    j += 1
                i = match_close(toks, j, "{", "}") + 1
                self.type_names.add(nm)
                self.struct_names.add(nm)
                self.supplied_seen.append((nm, rel, t.line))
                self.counts.hit("R24")
            else:
                raise SystemExit(f"{rel}:{t.line}: unexpected top-level token {t.text!r}")
        assert body is not None, rel
        self.parsed.append((cls, rel, body))
        # collect nested type names before any body is translated (R12 needs the full set)
        for k, t in enumerate(body):
            if t.kind == "id" and t.text in ("enum", "struct"):
                nm = body[next_sig(body, k + 1)].text
                self.type_names.add(nm)
                if t.text == "struct":
                    self.struct_names.add(nm)

    # ---------------------------------------------------------------- member level
    def split_members(self, toks):
        """yields token slices, one per member of a class / struct body"""
        i, n = 0, len(toks)
        while True:
            i = next_sig(toks, i)
            if i >= n:
                return
            start = i
            seen_eq = False
            j = i
            while j < n:
                t = toks[j]
                if t.kind == "op":
                    if t.text == "(":
                        j = match_close(toks, j, "(", ")")
                    elif t.text == "[":
                        j = match_close(toks, j, "[", "]")
                    elif t.text in ("=", "=>"):
                        seen_eq = True
                    elif t.text == "{":
                        j = match_close(toks, j, "{", "}")
                        if not seen_eq:
                            k = next_sig(toks, j + 1)
                            if k < n and toks[k].text == "=":      # `{ get; set; } = v;`
                                seen_eq = True
                            else:
                                if k < n and toks[k].text == ";":
                                    j = k
                                break
                    elif t.text == ";":
                        break
                j += 1
            yield toks[start:j + 1]
            i = j + 1

    def member(self, cls, toks, rel, enclosing_struct=None):
        """one member -> C++ text ('' when excluded)"""
        # R5: leading attributes
        i = next_sig(toks, 0)
        while toks[i].text == "[":
            i = next_sig(toks, match_close(toks, i, "[", "]") + 1)
            self.counts.hit("R5")
        toks = toks[i:]
        line = toks[0].line
        mods = []
        i = 0
        while toks[i].kind == "id" and (toks[i].text in MODIFIERS_DROP or toks[i].text in ("static", "const", "fixed")):
            mods.append(toks[i].text)
            i = next_sig(toks, i + 1)
        self.counts.hit("R6", sum(1 for m in mods if m in MODIFIERS_DROP))
        rest = toks[i:]
        head = rest[0].text
        where = f"/* {rel}:{line} */ "
        if head == "enum":
            name = rest[next_sig(rest, 1)].text
            b = next(k for k, t in enumerate(rest) if t.text == "{")
            e = match_close(rest, b, "{", "}")
            self.counts.hit("R7")
            hdr = "".join(t.text for t in rest[1:b])
            return f"{where}enum class{hdr}{{{self.body(rest[b + 1:e])}}};\n"
        if head == "struct":
            name = rest[next_sig(rest, 1)].text
            b = next(k for k, t in enumerate(rest) if t.text == "{")
            e = match_close(rest, b, "{", "}")
            inner = []
            has_ctor = False
            for m in self.split_members(rest[b + 1:e]):
                txt = self.member(cls, m, rel, enclosing_struct=name)
                has_ctor |= txt.lstrip().startswith(f"/* ctor */")
                inner.append("\t" + txt)
            if has_ctor:
                inner.insert(0, f"\t{name}() = default;\n")
            self.counts.hit("R7")
            return f"{where}struct {name} {{\n{''.join(inner)}}};\n"
        # method / ctor / property / field: find the first of `(`, `=`, `=>`, `{`, `;` at depth 0
        k = 0
        generic_at = None
        while rest[k].text not in ("(", "=", "=>", "{", ";"):
            if rest[k].text == "[":
                k = match_close(rest, k, "[", "]")
            if rest[k].text == "<":
                generic_at = k
                k = match_close(rest, k, "<", ">")
            k += 1
        name_idx = prev_sig(rest, (generic_at if generic_at is not None else k) - 1)
        name = rest[name_idx].text
        if cls in ONLY and enclosing_struct is None and name not in ONLY[cls] and (cls, name) not in EXCLUDED:
            self.not_taken.append((cls, name, rel, line))
            self.counts.hit("R24")
            return f"{where}// not taken: {name} (R24)\n"
        if (cls, name) in EXCLUDED and enclosing_struct is None:
            self.excluded_seen.append((cls, name, rel, line))
            return f"{where}// excluded: {name} -- {EXCLUDED[(cls, name)]}; supplied by oracle/ref_prelude.hpp\n"
        if rest[k].text == "(":
            close = match_close(rest, k, "(", ")")
            rtype = rest[:name_idx]
            params = self.params(rest[k + 1:close])
            after = next_sig(rest, close + 1)
            is_ctor = enclosing_struct is not None and name == enclosing_struct and not any(t.sig for t in rtype)
            rt = self.rtype(rtype)
            static = "static " if "static" in mods else ""
            if rest[after].text == "=>":
                end = len(rest) - 1
                assert rest[end].text == ";"
                expr = self.body(rest[after + 1:end])
                self.counts.hit("R8")
                ret = "" if rt.strip() == "void" else "return "
                return f"{where}{static}{rt}{self.ident(name)}({params}) {{ {ret}{expr.strip()}; }}\n"
            assert rest[after].text == "{", (rel, line, rest[after])
            end = match_close(rest, after, "{", "}")
            bodytxt = self.body(rest[after + 1:end])
            if is_ctor:
                return f"/* ctor */ {where}{name}({params}) {{{bodytxt}}}\n"
            return f"{where}{static}{rt}{self.ident(name)}({params})\n\t{{{bodytxt}}}\n"
        if rest[k].text in ("{", "=>"):
            raise SystemExit(f"{rel}:{line}: property `{name}` is not in EXCLUDED")
        # field
        ftype = rest[:name_idx]
        ttxt = self.body(ftype).strip()
        arr = ""
        if ttxt.endswith("[]"):
            ttxt, arr = ttxt[:-2].rstrip(), "[]"
        tail = self.body(rest[name_idx + 1:])        # `= init;`, `[N];` or `;`
        if "const" in mods:
            q = "static constexpr "
        elif "static" in mods:
            q = "static inline "
        else:
            q = ""
        if "fixed" in mods:
            self.counts.hit("R7")
        self.counts.hit("R6")
        return f"{where}{q}{ttxt} {self.ident(name)}{arr}{tail}\n"

    def ident(self, name):
        if name.startswith("@"):
            self.counts.hit("R10")
            return name[1:] + "_"
        if name in CPP_KEYWORDS:
            self.counts.hit("R19")
            return name + "_"
        return name

    def rtype(self, toks):
        sig = [t for t in toks if t.sig]
        if sig and sig[0].text == "ref":
            self.counts.hit("R9")
            return self.body(sig[1:]).strip() + "& "
        return self.body(toks).strip() + " " if sig else ""

    def params(self, toks):
        out, cur, depth = [], [], 0
        for t in toks:
            if t.kind == "op" and t.text in "([{":
                depth += 1
            elif t.kind == "op" and t.text in ")]}":
                depth -= 1
            if t.kind == "op" and t.text == "," and depth == 0:
                out.append(cur)
                cur = []
            else:
                cur.append(t)
        if any(t.sig for t in cur):
            out.append(cur)
        res = []
        for p in out:
            i = next_sig(p, 0)
            while p[i].text == "[":
                i = next_sig(p, match_close(p, i, "[", "]") + 1)
                self.counts.hit("R5")
            p = p[i:]
            byref = p[0].text in ("ref", "out")
            if byref:
                p = p[next_sig(p, 1):]
                self.counts.hit("R9")
            # split `type name [= default]`
            eq = next((k for k, t in enumerate(p) if t.text == "="), len(p))
            nm = prev_sig(p, eq - 1)
            ty = self.body(p[:nm]).strip()
            txt = f"{ty}{'&' if byref else ''} {self.ident(p[nm].text)}"
            if eq < len(p):
                txt += " =" + self.body(p[eq + 1:])
            res.append(txt)
        return ", ".join(res)

    # ---------------------------------------------------------------- statement / expression level
    def switch_exprs(self, toks):
        """R22: every `s switch { arms }` of the token list replaced by one raw token holding its C++ text"""
        while True:
            at = next((i for i, t in enumerate(toks) if t.kind == "id" and t.text == "switch" and
                       next_sig(toks, i + 1) < len(toks) and toks[next_sig(toks, i + 1)].text == "{"), None)
            if at is None:
                return toks
            pv = prev_sig(toks, at - 1)
            if toks[pv].text == ")":
                depth, start = 0, pv
                while True:
                    if toks[start].text == ")":
                        depth += 1
                    elif toks[start].text == "(":
                        depth -= 1
                        if depth == 0:
                            break
                    start -= 1
            else:
                assert toks[pv].kind == "id", f"switch expression over {toks[pv]} at line {toks[at].line}"
                start = pv
            scrut = self.body(toks[start:pv + 1]).strip()
            b = next_sig(toks, at + 1)
            e = match_close(toks, b, "{", "}")
            arms, cur, depth = [], [], 0
            for t in toks[b + 1:e]:
                if t.kind == "op" and t.text in "([{":
                    depth += 1
                elif t.kind == "op" and t.text in ")]}":
                    depth -= 1
                if t.kind == "op" and t.text == "," and depth == 0:
                    arms.append(cur); cur = []
                else:
                    cur.append(t)
            if any(t.sig for t in cur):
                arms.append(cur)
            pieces = []
            for arm in arms:
                k = next(i for i, t in enumerate(arm) if t.kind == "op" and t.text == "=>")
                pat = [t for t in arm[:k] if t.sig]
                expr = self.body(arm[k + 1:]).strip()
                bind, conds, i2 = "", [], 0
                if len(pat) == 1 and pat[0].text == "_":
                    cond = None
                elif len(pat) == 2 and pat[0].text == "var":
                    cond, bind = None, f"auto {self.ident(pat[1].text)} = _v; "
                else:
                    txt, i2 = [], 0
                    while i2 < len(pat):
                        t = pat[i2]
                        if t.kind == "id" and t.text in ("or", "and"):
                            txt.append(" || " if t.text == "or" else " && ")
                            i2 += 1
                            continue
                        rel = "=="
                        if t.kind == "op" and t.text in ("<", ">", "<=", ">="):
                            rel = t.text
                            i2 += 1
                        j2 = i2
                        while j2 < len(pat) and not (pat[j2].kind == "id" and pat[j2].text in ("or", "and")):
                            j2 += 1
                        txt.append(f"(_v {rel} {self.body(pat[i2:j2]).strip()})")
                        i2 = j2
                    cond = "".join(txt)
                stmt = expr + ";" if expr.startswith("throw ") else f"return {expr};"
                pieces.append(f"{{ {bind}{stmt} }}" if cond is None else f"if ({cond}) {{ {stmt} }}")
            self.counts.hit("R22")
            raw = Tok("raw", f"([&](auto _v) {{ {' '.join(pieces)} }})({scrut})", toks[at].line)
            toks = toks[:start] + [raw] + toks[e + 1:]

    def body(self, toks):
        toks = self.switch_exprs(list(toks))
        out = []            # text pieces
        stmt_start = 0      # index in `out` where the current statement began (R17)
        i, n = 0, len(toks)
        while i < n:
            t = toks[i]
            if not t.sig:
                out.append(t.text)
                i += 1
                continue
            x = t.text
            if t.kind == "id":
                nx = next_sig(toks, i + 1)
                nxt = toks[nx].text if nx < n else ""
                pv = prev_sig(toks, i - 1)
                prv = toks[pv].text if pv >= 0 else ""
                if x == "var":
                    out.append("auto"); self.counts.hit("R10")
                elif x == "null":
                    out.append("nullptr"); self.counts.hit("R10")
                elif x == "this" and nxt == ".":
                    out.append("this->"); self.counts.hit("R10")
                    i = nx + 1
                    continue
                elif x == "unchecked" and nxt == "(":
                    self.counts.hit("R11")          # the parenthesis that follows stays
                elif x == "ref" and prv in ("", "(", ",", "=", "return", "=>", "{", ";"):
                    self.counts.hit("R9")
                elif x == "new" and toks[nx].kind == "id" and toks[nx].text in self.struct_names:
                    self.counts.hit("R13")
                elif x == "stackalloc":
                    b = next(k for k in range(i, n) if toks[k].text == "[")
                    e = match_close(toks, b, "[", "]")
                    ty = self.body(toks[nx:b]).strip()
                    out.append(f"({ty}*) alloca(sizeof({ty}) * ({self.body(toks[b + 1:e]).strip()}))")
                    self.counts.hit("R14")
                    i = e + 1
                    continue
                elif x == "sizeof" and nxt == "(":
                    e = match_close(toks, nx, "(", ")")
                    out.append(f"((int) sizeof({self.body(toks[nx + 1:e]).strip()}))")
                    self.counts.hit("R15")
                    i = e + 1
                    continue
                elif x == "try" and nxt == "{":
                    ae = match_close(toks, nx, "{", "}")
                    f = next_sig(toks, ae + 1)
                    assert toks[f].text == "finally", f"try without finally at line {t.line}"
                    fb = next_sig(toks, f + 1)
                    fe = match_close(toks, fb, "{", "}")
                    out.append("{ K4RefFinally _fin([&]() {" + self.body(toks[fb + 1:fe]) + "}); " + self.body(toks[nx + 1:ae]) + "}")
                    self.counts.hit("R16")
                    i = fe + 1
                    stmt_start = len(out)
                    continue
                elif x == "out" and nxt == "var" and prv in ("(", ","):
                    v = toks[next_sig(toks, nx + 1)]
                    # the callee: dotted name in front of the innermost unmatched `(`
                    depth, k = 0, i - 1
                    while k >= 0:
                        if toks[k].text == ")":
                            depth += 1
                        elif toks[k].text == "(":
                            if depth == 0:
                                break
                            depth -= 1
                        k -= 1
                    callee, k = [], prev_sig(toks, k - 1)
                    while k >= 0 and (toks[k].kind == "id" or toks[k].text == "."):
                        callee.insert(0, toks[k].text)
                        k = prev_sig(toks, k - 1)
                    ty = OUT_TYPES["".join(callee)]
                    out.insert(stmt_start, f"{ty} {v.text}; ")
                    out.append(v.text)
                    self.counts.hit("R17")
                    i = next_sig(toks, nx + 1) + 1
                    continue
                elif nxt == "." and x in self.type_names and prv not in (".", "->"):
                    out.append(self.ident(x) + "::"); self.counts.hit("R12")
                    i = nx + 1
                    continue
                else:
                    out.append(self.ident(x))
                i += 1
                continue
            if t.kind == "op":
                if x == "-=":
                    # R18: `a->end -= a->@base`
                    pv = prev_sig(toks, i - 1)
                    e = i
                    while toks[e].text != ";":
                        e += 1
                    rhs = [u for u in toks[i + 1:e] if u.sig]
                    if toks[pv].kind == "id" and rhs and (self.ident(toks[pv].text), self.ident(rhs[-1].text)) in PTR_DIFF_ASSIGN \
                            and toks[prev_sig(toks, pv - 1)].text == "->":
                        lhs = "".join(out[stmt_start:]).strip()
                        out.append(f"= (byte*) ({lhs} - {self.body(toks[i + 1:e]).strip()})")
                        self.counts.hit("R18")
                        i = e
                        continue
                out.append(x)
                if x in (";", "{", "}"):
                    stmt_start = len(out)
                i += 1
                continue
            if t.kind == "str" and x.startswith("$"):
                x = x.lstrip("$@")                      # R23
                self.counts.hit("R23")
            out.append(x)
            i += 1
        return "".join(out)

    # ---------------------------------------------------------------- output
    def emit(self):
        for cls, rel, body in self.parsed:
            for m in self.split_members(body):
                self.classes[cls]["members"].append(self.member(cls, m, rel))
        o = []
        for cls, c in self.classes.items():
            base = f" : {c['base']}" if c["base"] else ""
            o.append(f"// ===== {cls}: {', '.join(c['files'])}\nstruct {cls}{base} {{\n")
            for a, b in c["aliases"]:
                o.append(f"\tusing {a} = {'::k4ref::' + b if b in self.classes else b};\n")
            o.append(f"#ifdef K4REF_MEMBERS_{cls}\n\tK4REF_MEMBERS_{cls}\n#endif\n")
            # R21: nested enums first (C++ wants a type declared before a member declaration names it; C# does not care)
            enums = [m for m in c["members"] if "enum class" in m.split("*/", 1)[-1][:14]]
            self.counts.hit("R21", len(enums))
            for m in enums + [m for m in c["members"] if m not in enums]:
                o.append("\t" + m)
            o.append("};\n\n")
        return "".join(o)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True, help="directory for k4ref_engine.hpp + make_ref_report.json (oracle/Makefile passes a scratch directory)")
    ap.add_argument("-D", dest="defines", action="append", default=[],
                    help="C# conditional symbols (e.g. NET5_0_OR_GREATER); BIT32 comes from the x32 files themselves")
    args = ap.parse_args()
    if not os.path.isdir(SRC_ROOT):
        raise SystemExit(f"{SRC_ROOT} not present: oracle/_ref can only be generated where the reference is")
    tr = Translator(args.defines)
    digest = hashlib.sha256()
    for cls, files in INPUTS:
        for rel in files:
            digest.update(open(os.path.join(SRC_ROOT, rel), "rb").read())
            tr.load(cls, rel)
    text = tr.emit()
    missing = [k for k in EXCLUDED if k not in {(c, n) for c, n, _, _ in tr.excluded_seen}]
    if missing:
        raise SystemExit(f"EXCLUDED names never met in the inputs: {missing}")
    os.makedirs(args.out, exist_ok=True)
    hdr = ["// GENERATED by oracle/make_ref.py from /root/reference (read where it lies); lives in a scratch directory while g++ compiles it.",
           "// Every function body below is the reference's own statement sequence; only the spelling rules",
           "// R1..R20 documented in oracle/make_ref.py were applied.  Do not edit.",
           f"// inputs sha256: {digest.hexdigest()}",
           f"// conditional symbols: {sorted(args.defines)}",
           "// excluded members (supplied by oracle/ref_prelude.hpp):"]
    for c, nme, rel, line in tr.excluded_seen:
        hdr.append(f"//   {c}.{nme}  ({rel}:{line})  {EXCLUDED[(c, nme)]}")
    hdr.append(f'#pragma once\n#define K4REF_INPUTS_SHA256 "{digest.hexdigest()}"\nnamespace k4ref {{\n')
    with open(os.path.join(args.out, "k4ref_engine.hpp"), "w") as f:
        f.write("\n".join(hdr) + text + "} // namespace k4ref\n")
    report = dict(inputs_sha256=digest.hexdigest(), defines=sorted(args.defines), rules=dict(sorted(tr.counts.items())),
                  excluded=[dict(cls=c, member=n, file=rel, line=line, why=EXCLUDED[(c, n)]) for c, n, rel, line in tr.excluded_seen],
                  not_taken=[dict(cls=c, member=n, file=rel, line=line) for c, n, rel, line in tr.not_taken],
                  supplied_structs=[dict(name=n, file=rel, line=line) for n, rel, line in tr.supplied_seen],
                  files=[f for _, fs in INPUTS for f in fs], generated_lines=text.count("\n"))
    with open(os.path.join(args.out, "make_ref_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(f"[make_ref] {len(report['files'])} files -> {args.out}/k4ref_engine.hpp ({report['generated_lines']} lines), "
          f"rules applied: {sum(tr.counts.values())}, excluded members: {len(tr.excluded_seen)}")


if __name__ == "__main__":
    main()
