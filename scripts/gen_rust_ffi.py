#!/usr/bin/env python3
"""Generate the Rust `extern "C"` binding of include/diee.h (the block INTEGRATION.md section 2 shows a die-e maintainer).

    python scripts/gen_rust_ffi.py            # print src/diee_ffi.rs
    python scripts/gen_rust_ffi.py --update   # rewrite the generated block inside INTEGRATION.md

Every `#define` constant, `typedef struct`, `typedef enum` and function prototype of the header is translated, so the
binding cannot fall behind the boundary: tests/test_abi_cpu.py re-generates it and compares it with INTEGRATION.md, and
checks name and arity of every prototype against the header."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "diee.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- BEGIN GENERATED diee_ffi.rs (scripts/gen_rust_ffi.py) -->", "<!-- END GENERATED diee_ffi.rs -->"

SCALARS = {"int": "c_int", "unsigned": "c_uint", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "float": "f32",
           "double": "f64", "int8_t": "i8", "uint8_t": "u8", "uint16_t": "u16", "int32_t": "i32", "char": "c_char",
           "void": "c_void", "diee_status": "c_int"}


def camel(name):
    return "".join(p.capitalize() for p in name.split("_"))


def strip_comments(src):
    return re.sub(r"/\*.*?\*/", " ", src, flags=re.S)


def rust_type(ctype):
    """'const diee_bg_state*' -> '*const DieeBgState' ..."""
    t = ctype.strip()
    depth = []
    while t.endswith("*") or t.endswith("* const") or t.endswith("*const"):
        t = re.sub(r"\*\s*(const)?$", "", t).strip()
        # constness of THIS pointer level is that of what it points to (the qualifier to its left)
        depth.append(None)
    const = False
    toks = t.split()
    if "const" in toks:
        const = True
        toks = [x for x in toks if x != "const"]
    base = " ".join(toks)
    r = SCALARS.get(base) or (camel(base) if base.startswith("diee_") else None)
    if r is None:
        raise ValueError(f"unknown C type {ctype!r}")
    n = len(depth)
    if n == 0:
        return r
    # innermost pointer carries the base type's constness; outer levels of `T* const*` are const, of `T**` mut
    inner = ("*const " if const else "*mut ") + r
    for _ in range(n - 1):
        inner = ("*const " if re.search(r"\*\s*const\s*\*", ctype) else "*mut ") + inner
    return inner


def split_decl(arg):
    """'const float* blob' -> ('const float*', 'blob'); 'float out[144]' is not used by the header"""
    arg = arg.strip()
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", arg)
    ctype, name = m.group(1).strip(), m.group(2)
    if name in ("type", "in", "ref", "box", "move", "fn", "use", "mod"):
        name += "_"
    return ctype, name


def parse(src):
    src = strip_comments(src)
    consts = [(m.group(1), m.group(2)) for m in re.finditer(r"^#define\s+(DIEE_[A-Z0-9_]+)\s+\(?(-?[0-9]+)u?\)?\s*$", src, flags=re.M)]
    enums = []
    for m in re.finditer(r"typedef\s+enum\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        enums.append((m.group(2), [(a.strip(), int(b)) for a, b in re.findall(r"(\w+)\s*=\s*(\d+)", m.group(1))]))
    structs = []
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for stmt in m.group(1).split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            first, *rest = [x.strip() for x in stmt.split(",")]
            ctype, name = split_decl(re.sub(r"\[\d+\]", "", first))
            base = ctype.rstrip("*").strip()
            for decl in [first] + [ctype_part for ctype_part in rest]:
                mm = re.match(r"^(?:.*?)(\**)\s*([A-Za-z_]\w*)\s*(?:\[(\d+)\])?$", decl if decl is first else base + " " + decl)
                stars, fname, arr = mm.group(1), mm.group(2), mm.group(3)
                t = rust_type(base + stars)
                fields.append((fname, f"[{t}; {arr}]" if arr else t))
        structs.append((m.group(2), fields))
    opaque = re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", src)
    body = re.sub(r"typedef\s+(struct|enum)\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    body = re.sub(r"^\s*#.*$", " ", body, flags=re.M)
    body = body.replace('extern "C" {', " ")
    funcs = []
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(diee_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", body):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = [] if args in ("", "void") else [split_decl(a) if re.search(r"[A-Za-z_]\w*$", a.strip()) and not a.strip().endswith("*") else (a.strip(), None)
                                                  for a in args.split(",")]
        funcs.append((ret, name, params))
    return consts, enums, structs, opaque, funcs


def generate(src=None):
    consts, enums, structs, opaque, funcs = parse(src if src is not None else open(HEADER).read())
    out = ["// src/diee_ffi.rs -- GENERATED from include/diee.h by scripts/gen_rust_ffi.py; do not edit",
           "#![allow(non_camel_case_types, dead_code)]",
           "use std::os::raw::{c_char, c_int, c_void};", ""]
    for name, val in consts:
        ty = "i32" if val.startswith("-") or name.startswith("DIEE_GAME") or name == "DIEE_NO_MOVE" else "u32"
        out.append(f"pub const {name}: {ty} = {val};")
    for ename, items in enums:
        out.append(f"// {ename} (returned as c_int)")
        for a, b in items:
            out.append(f"pub const {a}: c_int = {b};")
    out.append("")
    for sname, fields in structs:
        derive = "#[repr(C)] #[derive(Clone, Copy)]"
        out.append(f"{derive}\npub struct {camel(sname)} {{   // {sname}")
        for fname, t in fields:
            out.append(f"    pub {fname}: {t},")
        out.append("}")
    for o in opaque:
        out.append(f"pub enum {camel(o)} {{}}   // opaque {o}")
    out.append("")
    out.append('extern "C" {')
    for ret, name, params in funcs:
        ps = []
        for i, (ctype, pname) in enumerate(params):
            rt = rust_type(ctype)
            ps.append(f"{pname or ('ctx' if rt.endswith('DieeCtx') else 'f' if rt.endswith('DieeFragments') else f'arg{i}')}: {rt}")
        r = "" if ret == "void" else f" -> {rust_type(ret)}"
        out.append(f"    pub fn {name}({', '.join(ps)}){r};")
    out.append("}")
    return "\n".join(out) + "\n"


def prototypes(src=None):
    """[(name, arity)] of the header, for the test"""
    return [(name, len(params)) for _, name, params in parse(src if src is not None else open(HEADER).read())[4]]


def update_doc():
    doc = open(DOC).read()
    a, b = doc.index(BEGIN), doc.index(END)
    doc = doc[:a] + BEGIN + "\n```rust\n" + generate() + "```\n" + doc[b:]
    open(DOC, "w").write(doc)


if __name__ == "__main__":
    if "--update" in sys.argv:
        update_doc()
    else:
        sys.stdout.write(generate())
