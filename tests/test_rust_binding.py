"""bindings/rust/ cannot be compiled here (no rustc / cargo in the image), so it is kept honest mechanically:

  * src/ffi.rs declares EXACTLY the functions include/fastlanes_amd.h declares -- same names, same arity, and every argument /
    return type the Rust spelling of the C one (this file's own C-to-Rust table, independent of tools/gen_rust_ffi.py);
  * every `ffi::fl_*` call in src/gpu_impl.rs and src/device.rs (after expanding their macro_rules! textually) names a declared
    function and passes as many arguments as it takes;
  * every ```rust block of INTEGRATION.md only uses ffi functions / binding items that exist.

The reference-side boundary is a Rust `impl` over this FFI; a header change that is not mirrored there fails here."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fastlanes_amd.h")
RUST = os.path.join(ROOT, "bindings", "rust", "src")


# ---- the C side: prototypes after the preprocessor (the per-type macro expanded) ---------------------------------------------
def c_prototypes():
    text = subprocess.run(["gcc", "-E", "-P", HEADER], check=True, capture_output=True, text=True).stdout
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(fl_\w+)\s*\(([^()]*)\)\s*;", text):
        ret, name, params = " ".join(m.group(1).split()), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        args = []
        if params and params != "void":
            for p in params.split(","):
                p = " ".join(p.replace("*", " * ").split())
                args.append(re.sub(r"\s*\w+$", "", p).strip())      # drop the parameter name
        assert name not in protos, f"{name} declared twice"
        protos[name] = (ret, args)
    return protos


C_BASE = {"unsigned": "u32", "unsigned int": "u32", "int": "i32", "size_t": "usize", "uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32",
          "uint64_t": "u64", "char": "c_char", "void": "c_void", "fl_mixed_plan": "fl_mixed_plan"}


def expected_rust(ctype):
    """The Rust FFI spelling of a C type: scalars by the table; `const X *` -> `*const X`, `X *` -> `*mut X`, applied inside-out."""
    toks = ctype.replace("*", " * ").split()
    quals, i = [], 0
    while i < len(toks) and toks[i] != "*":
        quals.append(toks[i])
        i += 1
    const = "const" in quals
    base = " ".join(q for q in quals if q != "const")
    rust = C_BASE[base]
    while i < len(toks):
        assert toks[i] == "*"
        i += 1
        nxt_const = i < len(toks) and toks[i] == "const"
        if nxt_const:
            i += 1
        rust = ("*const " if const else "*mut ") + rust
        const = nxt_const
    return rust


# ---- the Rust side -----------------------------------------------------------------------------------------------------------
def rust_decls(path=os.path.join(RUST, "ffi.rs")):
    src = re.sub(r"//[^\n]*", "", open(path).read())
    decls = {}
    for block in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\}', src, re.S):
        for m in re.finditer(r"pub\s+fn\s+(\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+?))?\s*;", block.group(1), re.S):
            name, params, ret = m.group(1), m.group(2), (m.group(3) or "()").strip()
            args = []
            for p in filter(None, (x.strip() for x in params.split(","))):
                pname, ptype = p.split(":", 1)
                args.append(" ".join(ptype.split()))
            assert name not in decls, f"{name} declared twice in ffi.rs"
            decls[name] = (" ".join(ret.split()), args)
    return decls


def test_ffi_rs_declares_exactly_what_the_header_declares():
    c, r = c_prototypes(), rust_decls()
    assert len(c) > 140, "the header parse lost its functions"
    assert sorted(set(c) - set(r)) == [], "declared in fastlanes_amd.h, missing from ffi.rs"
    assert sorted(set(r) - set(c)) == [], "declared in ffi.rs, not in fastlanes_amd.h"
    for name, (cret, cargs) in c.items():
        rret, rargs = r[name]
        assert len(rargs) == len(cargs), (name, cargs, rargs)
        assert rret == ("()" if cret == "void" else expected_rust(cret)), (name, cret, rret)
        for i, (ca, ra) in enumerate(zip(cargs, rargs)):
            assert ra == expected_rust(ca), (name, i, ca, ra)
    # per element type the same surface (a method added for one type only is a header bug)
    per_type = {ty: sorted(n[len(f"fl_{ty}_"):] for n in c if n.startswith(f"fl_{ty}_")) for ty in ("u8", "u16", "u32", "u64")}
    assert per_type["u8"] == per_type["u16"] == per_type["u32"] == per_type["u64"] and len(per_type["u8"]) >= 34
    # the measurement hooks of fastlanes_amd_internal.h are NOT part of the binding
    assert not any(n.startswith("fl_internal_") for n in r)


def test_ffi_rs_is_what_the_generator_writes():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_ffi_constants_match_the_header():
    h = open(HEADER).read()
    rs = open(os.path.join(RUST, "ffi.rs")).read()
    consts = dict(re.findall(r"\b(FL_(?:OK|ERR_\w+|DEVERR_\w+|CMP_\w+))\s*=\s*(\d+)", h))
    assert len(consts) >= 18
    for name, value in consts.items():
        m = re.search(rf"pub const {name}: [iu]32 = (\d+);", rs)
        assert m and m.group(1) == value, name


# ---- the hand-written Rust on top of ffi.rs ------------------------------------------------------------------------------------
def expand_macros(src):
    """macro_rules! name { (params) => { body } }  +  name!(args);  ->  the bodies with $param replaced by the arguments (textual;
    enough to see which ffi function each call site reaches and with how many arguments)."""
    out = [src]
    for m in re.finditer(r"macro_rules!\s*(\w+)\s*\{\s*\(([^)]*)\)\s*=>\s*\{(.*?)\n    \};\n\}", src, re.S):
        name, params, body = m.group(1), m.group(2), m.group(3)
        pnames = re.findall(r"\$(\w+)\s*:", params)
        for inv in re.finditer(rf"\b{name}!\s*\(([^;]*?)\)\s*;", src, re.S):
            args = [a.strip() for a in inv.group(1).split(",")]
            assert len(args) == len(pnames), (name, args, pnames)
            text = body
            for p, a in sorted(zip(pnames, args), key=lambda pa: -len(pa[0])):
                text = re.sub(rf"\${p}\b", a, text)
            out.append(text)
    return "\n".join(out)


def call_sites(text):
    """(function, number of arguments) of every `ffi::fl_xxx(...)` call"""
    sites = []
    for m in re.finditer(r"ffi::(fl_\w+)\s*\(", text):
        depth, i, n_args, seen = 1, m.end(), 0, False
        while depth:
            ch = text[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == "," and depth == 1:
                n_args += 1
            if depth and not ch.isspace():
                seen = True
            i += 1
        sites.append((m.group(1), n_args + 1 if seen else 0))
    return sites


@pytest.mark.parametrize("fname", ["gpu_impl.rs", "device.rs"])
def test_binding_sources_call_declared_functions_with_the_right_arity(fname):
    decls = rust_decls()
    text = expand_macros(re.sub(r"//[^\n]*", "", open(os.path.join(RUST, fname)).read()))
    sites = [s for s in call_sites(text) if "$" not in s[0]]
    assert len(sites) >= 20, (fname, len(sites))
    for fn, n in sites:
        assert fn in decls, f"{fname}: ffi::{fn} is not declared in ffi.rs"
        assert n == len(decls[fn][1]), f"{fname}: ffi::{fn} called with {n} arguments, takes {len(decls[fn][1])}"
    # all four element types are wired
    for ty in ("u8", "u16", "u32", "u64"):
        assert any(fn.startswith(f"fl_{ty}_") for fn, _ in sites), (fname, ty)


def test_integration_md_rust_snippets_resolve():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```rust\n(.*?)```", md, re.S)
    assert len(blocks) >= 4
    decls = rust_decls()
    device_rs = open(os.path.join(RUST, "device.rs")).read()
    ffi_rs = open(os.path.join(RUST, "ffi.rs")).read()
    device_items = set(re.findall(r"\bfn (\w+)", device_rs)) | set(re.findall(r"pub (?:struct|trait) (\w+)", device_rs))
    checked = 0
    for b in blocks:
        code = re.sub(r"//[^\n]*", "", b)
        for fn in re.findall(r"(?<!core::)\bffi::(\w+)", code):
            if fn.startswith("$"):
                continue
            if fn.startswith("fl_"):
                assert fn in decls, f"INTEGRATION.md uses ffi::{fn}, which ffi.rs does not declare"
            else:
                assert re.search(rf"\b(?:fn|const) {fn}\b", ffi_rs), f"INTEGRATION.md uses ffi::{fn}, which ffi.rs does not define"
            checked += 1
        for fn, n in call_sites(code):
            if "$" not in fn and fn in decls:
                assert n == len(decls[fn][1]), f"INTEGRATION.md calls ffi::{fn} with {n} arguments, it takes {len(decls[fn][1])}"
        for fn in re.findall(r"\bpub fn (fl_\w+)", code):            # declarations quoted in the document
            assert fn in decls, f"INTEGRATION.md declares {fn}, which the header does not"
        for item in re.findall(r"\b(?:u8|u16|u32|u64)::(\w+_column|unpack_chunks)\b", code) + re.findall(r"\b(DeviceSlice(?:Mut)?|ChunkTable|DeviceCodec|Stream)\b", code):
            assert item in device_items, f"INTEGRATION.md uses {item}, which device.rs does not define"
            checked += 1
    assert checked >= 10
