"""Regenerate the machine-derived part of INTEGRATION.md's Rust binding from include/nuts_amd.h.

    python tools/gen_rust_binding.py            # rewrite the section between the GENERATED markers of INTEGRATION.md
    python tools/gen_rust_binding.py --check    # exit 1 if the section is stale

What a maintainer with a Rust toolchain would get from bindgen: the `#[repr(C)]` structs (field for field, in ABI order),
`NM_ABI_VERSION`, and one `extern "C"` declaration per exported function.  tests/test_integration_binding.py parses the document
independently of this script and compares it with the header."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nuts_amd.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN = "// ---- BEGIN GENERATED from include/nuts_amd.h by tools/gen_rust_binding.py (do not edit by hand) ----"
END = "// ---- END GENERATED ----"

STRUCT_NAMES = {"nm_settings": "NmSettings", "nm_logp_spec": "NmLogpSpec", "nm_engine_config": "NmEngineConfig",
                "nm_draw_stats": "NmDrawStats", "nm_draw_outputs": "NmDrawOutputs"}
OPAQUE = {"nm_engine": "NmEngine", "nm_math": "NmMath", "nm_vec": "NmVec"}
SCALARS = {"uint64_t": "u64", "int64_t": "i64", "double": "f64", "uint8_t": "u8", "char": "c_char", "void": "c_void", "int": "i32"}
FN_PTRS = {"nm_host_logp_fn": "Option<NmHostLogpFn>", "nm_lowrank_estimator_fn": "Option<NmLowrankEstimatorFn>"}
DERIVES = {"NmSettings": "#[repr(C)] #[derive(Clone, Copy)]", "NmDrawStats": "#[repr(C)] #[derive(Clone, Copy)]"}


def strip_comments(s):
    s = re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", s, flags=re.S))
    s = re.sub(r"^\s*#[^\n]*(\\\n[^\n]*)*", "", s, flags=re.M)          # preprocessor lines
    return re.sub(r'extern\s+"C"\s*\{', "", s)


def split_top(s):
    parts, depth, cur = [], 0, ""
    for ch in s:
        depth += ch in "(<[{"
        depth -= ch in ")>]}"
        if ch == "," and depth == 0:
            parts.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return [" ".join(p.split()) for p in parts if p.strip()]


def rust_type(ctype, array=None):
    """C declarator type (pointer stars attached) -> Rust."""
    t = ctype.strip().replace(" *", "*").replace("* ", "*")
    if t in FN_PTRS:
        return FN_PTRS[t]
    const = t.startswith("const ")
    if const:
        t = t[len("const "):]
    stars = len(t) - len(t.rstrip("*"))
    base = t.rstrip("*").strip()
    base = SCALARS.get(base) or STRUCT_NAMES.get(base) or OPAQUE.get(base)
    assert base, ctype
    if array is not None:                  # an array PARAMETER decays to a pointer; an array FIELD stays an array
        return ("*const " if const else "*mut ") + base if array == "param" else f"[{base}; {array}]"
    out = base
    for i in range(stars):
        out = ("*const " if (const and i == 0) else "*mut ") + out
    return out


def parse_header():
    raw = open(HEADER).read()
    version = int(raw.split("#define NM_ABI_VERSION")[1].split()[0])
    src = strip_comments(raw)
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fm = re.match(r"(.*?)(\w+)(\[(\d+)\])?$", decl)
            fields.append((fm.group(2), rust_type(fm.group(1), fm.group(4))))
        structs[m.group(1)] = fields
    fnptrs = {}
    for m in re.finditer(r"typedef\s+(\w+)\s*\(\s*\*\s*(\w+)\s*\)\s*\((.*?)\)\s*;", src, flags=re.S):
        fnptrs[m.group(2)] = (m.group(1), split_top(m.group(3)))
    nofp = re.sub(r"typedef[^;{]*\([^;]*;", "", src)
    funcs = []
    for m in re.finditer(r"([\w\s\*]+?)\b(nm_\w+)\s*\(([^)]*)\)\s*;", nofp):
        ret = " ".join(m.group(1).split())
        args = m.group(3).strip()
        params = []
        for a in ([] if args in ("", "void") else split_top(args)):
            fm = re.match(r"(.*?)(\w+)(\[\d*\])?$", a)
            params.append((fm.group(2), rust_type(fm.group(1), "param" if fm.group(3) else None)))
        funcs.append((m.group(2), ret, params))
    return version, structs, fnptrs, funcs


def wrap(items, indent, width=118, sep=" "):
    lines, cur = [], indent
    for it in items:
        if len(cur) + len(it) + 1 > width and cur.strip():
            lines.append(cur.rstrip()); cur = indent
        cur += it + sep
    if cur.strip():
        lines.append(cur.rstrip())
    return "\n".join(lines)


def generate():
    version, structs, fnptrs, funcs = parse_header()
    out = [BEGIN,
           f"pub const NM_ABI_VERSION: u64 = {version};     // include/nuts_amd.h; checked against nm_abi_version() in AmdChainBatch::new",
           ""]
    for cname, (ret, params) in fnptrs.items():
        rname = FN_PTRS[cname][len("Option<"):-1]
        ps = []
        for a in params:
            fm = re.match(r"(.*?)(\w+)$", a)
            ps.append(f"{fm.group(2)}: {rust_type(fm.group(1))}")
        out.append(f"pub type {rname} = unsafe extern \"C\" fn({', '.join(ps)}) -> {SCALARS[ret]};")
    out.append("")
    for cname, rname in STRUCT_NAMES.items():
        fields = structs[cname]
        out.append(f"{DERIVES.get(rname, '#[repr(C)]')}")
        words = sum(int(t.split("; ")[1].rstrip("]")) if t.startswith("[") else 1 for _, t in fields)
        out.append(f"pub struct {rname} {{           // == {cname}: {len(fields)} fields, {8 * words} bytes; field order is the ABI")
        out.append(wrap([f"pub {n}: {t}," for n, t in fields], "    "))
        out.append("}")
    for rname in OPAQUE.values():
        out.append(f"#[repr(C)] pub struct {rname} {{ _private: [u8; 0] }}")
    out.append("")
    out.append('#[link(name = "nuts_amd")]')
    out.append('extern "C" {')
    rets = {"nm_status": " -> i32", "void": "", "uint64_t": " -> u64", "const char*": " -> *const c_char", "const char *": " -> *const c_char", "int": " -> i32", "void*": " -> *mut c_void", "void *": " -> *mut c_void"}
    for name, ret, params in funcs:
        sig = f"pub fn {name}(" + ", ".join(f"{'r#' + n if n in ('fn', 'type', 'in', 'where') else n}: {t}" for n, t in params) + f"){rets[ret]};"
        if len(sig) + 4 <= 128:
            out.append("    " + sig)
        else:
            head = f"    pub fn {name}("
            out.append(head.rstrip())
            out.append(wrap([f"{n}: {t}," for n, t in params], "        "))
            out.append(f"    ){rets[ret]};")
    out.append("}")
    out.append(END)
    return "\n".join(out)


def main():
    doc = open(DOC).read()
    assert BEGIN in doc and END in doc, "markers missing in INTEGRATION.md"
    new = doc[:doc.index(BEGIN)] + generate() + doc[doc.index(END) + len(END):]
    if "--check" in sys.argv:
        sys.exit(0 if new == doc else 1)
    open(DOC, "w").write(new)


if __name__ == "__main__":
    main()
