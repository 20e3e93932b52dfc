"""INTEGRATION.md's Rust binding against include/nuts_amd.h.

The Rust source in INTEGRATION.md is what a nuts-rs maintainer would paste; no Rust toolchain exists in this image, so nothing
compiles it.  This test does what rustc + bindgen would: it parses the `#[repr(C)]` structs and the `extern "C"` blocks out of the
document and compares them with the header — field names, order, types and sizeof for every struct, name / parameter count /
return type for every function, and the ABI version the binding asserts.  A struct that lags the header (round 2: NmSettings 8
fields short, NmDrawStats 2 fields short = heap overflow in nm_engine_draw_to_host) fails here.  No GPU needed."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nuts_amd.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")

STRUCTS = {"nm_settings": "NmSettings", "nm_logp_spec": "NmLogpSpec", "nm_engine_config": "NmEngineConfig",
           "nm_draw_stats": "NmDrawStats", "nm_draw_outputs": "NmDrawOutputs"}
C_TO_RUST = {"uint64_t": "u64", "int64_t": "i64", "double": "f64", "double*": "*mut f64", "const double*": "*const f64",
             "const char*": "*const c_char", "void*": "*mut c_void", "nm_draw_stats*": "*mut NmDrawStats",
             "nm_host_logp_fn": "Option<NmHostLogpFn>"}


def _strip_c_comments(s):
    s = re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", s, flags=re.S))
    s = re.sub(r"^\s*#\s*(define|include|ifn?def|if|endif|else|pragma)\b[^\n]*(\\\n[^\n]*)*", "", s, flags=re.M)   # C preprocessor lines (not Rust's #[..])
    return re.sub(r'extern\s+"C"\s*\{\s*$', "", s, flags=re.M) if "typedef" in s else s


def header_structs():
    src = _strip_c_comments(open(HEADER).read())
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        name, body = m.group(1), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fm = re.match(r"(.*?)(\w+)(\[(\d+)\])?$", decl)
            ctype = fm.group(1).strip().replace(" *", "*").replace("* ", "*")
            rust = C_TO_RUST[ctype]
            if fm.group(4):
                rust = f"[{rust}; {fm.group(4)}]"
            fields.append((fm.group(2), rust))
        out[name] = fields
    return out


def _split_top(s):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[{":
            depth += 1
        elif ch in ")>]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return [p.strip() for p in parts if p.strip()]


def rust_source():
    doc = open(DOC).read()
    blocks = re.findall(r"```rust\n(.*?)```", doc, flags=re.S)
    return _strip_c_comments("\n".join(blocks))


def rust_structs():
    src = rust_source()
    out = {}
    for m in re.finditer(r"pub\s+struct\s+(Nm\w+)\s*\{(.*?)\}", src, flags=re.S):
        fields = []
        if "_private" in m.group(2):          # opaque handle types
            continue
        for f in _split_top(m.group(2)):
            fm = re.match(r"pub\s+(\w+)\s*:\s*(.+)$", f, flags=re.S)
            assert fm, f"cannot parse field {f!r} of {m.group(1)}"
            fields.append((fm.group(1), " ".join(fm.group(2).split())))
        out[m.group(1)] = fields
    return out


def test_rust_structs_match_the_header_field_for_field():
    H, R = header_structs(), rust_structs()
    for cname, rname in STRUCTS.items():
        assert cname in H, cname
        assert rname in R, f"INTEGRATION.md has no `pub struct {rname}`"
        hf, rf = H[cname], R[rname]
        assert [n for n, _ in rf] == [n for n, _ in hf], f"{rname}: field names / order differ from {cname}"
        for (n, ht), (_, rt) in zip(hf, rf):
            assert rt == ht, f"{rname}.{n}: {rt} in INTEGRATION.md, {ht} from the header"
        size = lambda fs: sum(8 * (int(re.search(r"; (\d+)\]", t).group(1)) if t.startswith("[") else 1) for _, t in fs)  # noqa: E731
        assert size(hf) == size(rf)
    # the sizes the ctypes mirror is checked against (tests/test_abi.py): one more tie between the three descriptions
    import ctypes as C
    from nuts_rs_amd import _lib
    assert 8 * len(H["nm_settings"]) == C.sizeof(_lib.NmSettings)
    assert 8 * len(H["nm_draw_stats"]) == _lib.STATS_DTYPE.itemsize
    assert [n for n, _ in H["nm_draw_stats"]] == list(_lib.STATS_DTYPE.names)
    assert [n for n, _ in H["nm_settings"]] == [n for n, _ in _lib.NmSettings._fields_]


def header_functions():
    src = _strip_c_comments(open(HEADER).read())
    src = re.sub(r"typedef[^;{]*\([^;]*;", "", src)          # function-pointer typedefs are not exports
    out = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(nm_\w+)\s*\(([^)]*)\)\s*;", src):
        ret = " ".join(m.group(1).split())
        args = m.group(3).strip()
        n = 0 if args in ("", "void") else len(_split_top(args))
        out[m.group(2)] = (ret, n)
    return out


def rust_functions():
    src = rust_source()
    out = {}
    for blk in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\}', src, flags=re.S):
        for m in re.finditer(r"pub\s+fn\s+(nm_\w+)\s*\((.*?)\)\s*(->\s*([^;]+))?;", blk.group(1), flags=re.S):
            out[m.group(1)] = ((m.group(4) or "()").strip(), len(_split_top(m.group(2))))
    return out


def test_rust_extern_block_matches_the_header():
    H, R = header_functions(), rust_functions()
    assert len(R) >= 40, "the binding should cover the engine's whole driver seam"
    ret_map = {"nm_status": "i32", "void": "()", "uint64_t": "u64", "const char*": "*const c_char", "const char *": "*const c_char",
               "void*": "*mut c_void", "void *": "*mut c_void", "int": "i32"}
    for name, (rret, rn) in R.items():
        assert name in H, f"INTEGRATION.md binds {name}, which include/nuts_amd.h does not declare"
        hret, hn = H[name]
        assert rn == hn, f"{name}: {rn} parameters in INTEGRATION.md, {hn} in the header"
        assert ret_map[hret] == rret, f"{name}: returns {rret} in INTEGRATION.md, {hret} in the header"
    # the driver seam must be complete: everything nm_engine_* / nm_settings_* / nm_init_* the header declares
    seam = [n for n in H if n.startswith(("nm_engine_", "nm_settings_", "nm_init_", "nm_host_", "nm_abi_", "nm_last_", "nm_chain_", "nm_pick_", "nm_lowrank_compute"))]
    missing = sorted(set(seam) - set(R))
    assert not missing, f"driver-seam functions missing from INTEGRATION.md's extern block: {missing}"


def test_binding_asserts_the_current_abi_version():
    want = int(open(HEADER).read().split("#define NM_ABI_VERSION")[1].split()[0])
    src = rust_source()
    m = re.search(r"pub\s+const\s+NM_ABI_VERSION\s*:\s*u64\s*=\s*(\d+)\s*;", src)
    assert m and int(m.group(1)) == want, "INTEGRATION.md must pin `pub const NM_ABI_VERSION: u64` to the header's value"
    assert re.search(r"nm_abi_version\(\)[\s}]*(!=|==)\s*NM_ABI_VERSION", src), "the binding must compare nm_abi_version() with NM_ABI_VERSION before creating an engine"
    doc = open(DOC).read()
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    n = 8 * len(header_structs()["nm_draw_stats"])
    assert f"nm_draw_stats, {n} B" in design, f"DESIGN.md section 2 must quote sizeof(nm_draw_stats) = {n} B"
    assert f"ABI v{want}" in doc


def test_generated_section_is_fresh():
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_binding.py"), "--check"])
    assert r.returncode == 0, "INTEGRATION.md's generated binding is stale: run python tools/gen_rust_binding.py"
