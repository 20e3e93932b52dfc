#!/usr/bin/env python3
"""A VGPR spill or reload placed BEFORE the exec restore of a join block (round 6; the root cause of DESIGN §22's incidents).

The defect, as found in the failing build of 8e9c172 (`nuts_draw_kernel<4,1,LrWrap<IidNormal>>`, block .LBB6_681 "Flow8074"):

        s_and_saveexec_b64 s[4:5], s[0:1]        ; if (!reuse_edge) ...
        s_cbranch_execz .LBB6_681                ; nobody enters: EXEC == 0 on this edge
        ...
    .LBB6_681:                                   ; the join block
        v_writelane_b32 v255, s64, 5             ; SGPR spills to VGPR lanes (exec-independent) at the block's top ...
        scratch_store_dwordx2 off, v[170:171], off offset:168 ; 8-byte Folded Spill     <- ... and the VGPR spills went in among them
        ...
        s_or_b64 exec, exec, s[4:5]              ; the exec restore that should have come first

When every lane skips the `if` the stores execute with EXEC == 0 and store nothing; the reloads (under the full mask, thousands of
instructions later) return what an EARLIER pass through the block left in the slot.  A scratch access is exec-dependent, v_writelane /
v_readlane are not: the compiler's spill placement took the SGPR spills for the block's prologue and put the VGPR spills after THEM
instead of after the exec restore.

Rule: in a block whose label is the target of an `s_cbranch_execz` (an edge that arrives with EXEC == 0), no exec-dependent spill access
(scratch_load / scratch_store; stack-relative buffer accesses) may sit between the label and the block's exec restore
(`s_or_b64 exec, exec, s[..]`, `s_or_saveexec_b64 s[..], s[..]`, `s_andn2_saveexec_b64`).
Input: the compiler's device assembly (-save-temps) or `llvm-objdump -d` text of a code object.  Exit 1 on a finding.
`--fix out.s`: write the assembly with each flagged exec restore moved up to its block's first instruction (only when everything it
jumps over is exec-independent or a spill access and does not write the restore's SGPR pair; otherwise the finding stays an error).
"""
import re
import sys

ASM_LABEL = re.compile(r"^(\.LBB\d+_\d+|[A-Za-z_$][\w.$]*):")
OBJ_SYM = re.compile(r"^[0-9a-f]+ <([^>]+)>:")
OBJ_ADDR = re.compile(r"//\s*([0-9A-Fa-f]+):")
OBJ_TARGET = re.compile(r"<[^>+]+\+0x([0-9A-Fa-f]+)>")
WIDEN = re.compile(r"^(s_or_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\]|vcc)|s_or_saveexec_b64\s+(s\[\d+:\d+\]|vcc),\s*(s\[\d+:\d+\]|vcc)|s_andn2_saveexec_b64\s)")
SPILL = re.compile(r"^(scratch_(load|store)_\w+|buffer_(load|store)_\w+\s[^;]*\bs\[0:3\],\s*(s32|s33|0)\b[^;]*\boffset:)")
EXEC_FREE = re.compile(r"^(v_writelane_b32|v_readlane_b32|s_nop|s_mov_b32\s+s|s_mov_b64\s+s|s_waitcnt)")


def parse(path):
    """-> list of dicts {kind: 'label'|'ins', text, key, line}; key = label name / instruction address (objdump) or None."""
    items = []
    objdump = None
    for ln, raw in enumerate(open(path, errors="replace"), 1):
        if objdump is None and ("file format elf64-amdgpu" in raw or OBJ_ADDR.search(raw)):
            objdump = True
        m = OBJ_SYM.match(raw)
        if m:
            items.append(dict(kind="func", text=m.group(1), key=None, line=ln)); continue
        m = ASM_LABEL.match(raw)
        if m:
            name = m.group(1)
            items.append(dict(kind="label" if name.startswith(".L") else "func", text=name, key=name, line=ln)); continue
        s = raw.strip()
        if not s or s.startswith(";") or s.startswith(".") or s.startswith("//"):
            continue
        addr = None
        ma = OBJ_ADDR.search(raw)
        if ma:
            addr = int(ma.group(1), 16)
        text = s.split("//")[0].split(";")[0].strip()
        if not text:
            continue
        tgt = None
        if text.startswith("s_cbranch") or text.startswith("s_branch"):
            mt = OBJ_TARGET.search(raw)
            if mt and ma:
                tgt = ("sym", raw[raw.rfind("<") + 1:raw.rfind("+0x")], int(mt.group(1), 16))
            else:
                parts = text.split()
                tgt = parts[1] if len(parts) > 1 else None
        items.append(dict(kind="ins", text=text, key=addr, line=ln, raw=s, target=tgt))
    return items


def scan(path):
    items = parse(path)
    # function start addresses (objdump) to resolve <sym+0xoff>
    sym_addr = {}
    last_func = None
    for k, it in enumerate(items):
        if it["kind"] == "func":
            last_func = it["text"]
            for j in range(k + 1, len(items)):
                if items[j]["kind"] == "ins":
                    if items[j]["key"] is not None:
                        sym_addr[last_func] = items[j]["key"]
                    break
    execz_targets = set()
    for it in items:
        if it["kind"] == "ins" and it["text"].startswith("s_cbranch_execz"):
            t = it["target"]
            if isinstance(t, tuple):
                base = sym_addr.get(t[1])
                if base is not None:
                    execz_targets.add(base + t[2])
            elif t:
                execz_targets.add(t)
    findings = []
    func = "?"
    k = 0
    n = len(items)
    while k < n:
        it = items[k]
        if it["kind"] == "func":
            func = it["text"]; k += 1; continue
        start = None
        if it["kind"] == "label" and it["key"] in execz_targets:
            start = k + 1; name = it["text"]
        elif it["kind"] == "ins" and it["key"] is not None and it["key"] in execz_targets:
            start = k; name = hex(it["key"])
        if start is None:
            k += 1; continue
        acc, between = [], []
        j = start
        while j < n and items[j]["kind"] == "ins":
            t = items[j]["text"]
            if j > start and items[j]["key"] is not None and items[j]["key"] in execz_targets:
                break
            if WIDEN.match(t):
                if acc:
                    findings.append(dict(func=func, block=name, widen=items[j], accesses=acc, between=between, first=items[start]))
                break
            if t.startswith("s_cbranch") or t.startswith("s_branch") or t.startswith("s_endpgm") or t.startswith("s_setpc") or t.startswith("s_swappc") \
               or re.match(r"^(s_\w+\s+exec\b|v_cmpx)", t):
                break
            if SPILL.match(t):
                acc.append(items[j])
            between.append(items[j])
            j += 1
        k = max(j, k + 1)
    return findings


def fixable(f):
    """The exec restore may move to the block's first instruction when everything it jumps over is exec-independent or a spill access,
    and nothing there writes the SGPR pair it reads."""
    m = re.search(r"(s\[(\d+):(\d+)\]|vcc)\s*$", f["widen"]["text"])
    if not m:
        return False
    src = m.group(1)
    lo, hi = (int(m.group(2)), int(m.group(3))) if m.group(2) else (None, None)
    for it in f["between"]:
        t = it["text"]
        if not (EXEC_FREE.match(t) or SPILL.match(t)):
            return False
        d = t.split(None, 1)[1].split(",")[0].strip() if " " in t else ""
        md = re.fullmatch(r"s(\d+)", d)
        if md and lo is not None and lo <= int(md.group(1)) <= hi:
            return False
        md = re.fullmatch(r"s\[(\d+):(\d+)\]", d)
        if md and lo is not None and not (int(md.group(2)) < lo or int(md.group(1)) > hi):
            return False
        if src == "vcc" and d.startswith("vcc"):
            return False
    return f["widen"]["text"].startswith("s_or_b64")


def main():
    argv = sys.argv[1:]
    fix_out = None
    if "--fix" in argv:
        i = argv.index("--fix"); fix_out = argv[i + 1]; del argv[i:i + 2]
    verbose = "-v" in argv
    paths = [a for a in argv if not a.startswith("-")]
    bad = 0
    for p in paths:
        fs = scan(p)
        for f in fs:
            print(f"{p}: {f['func']} block {f['block']}: {len(f['accesses'])} exec-dependent spill access(es) between the label (reached with EXEC == 0) and "
                  f"the exec restore at line {f['widen']['line']} ({f['widen']['text']}){' [fixable]' if fixable(f) else ''}")
            for it in (f["accesses"] if verbose else f["accesses"][:3]):
                print(f"    {it['line']}: {it['raw']}")
        if fix_out and len(paths) == 1:
            lines = open(p).read().split("\n")
            left = 0
            for f in sorted(fs, key=lambda f: -f["widen"]["line"]):
                if not fixable(f):
                    left += 1; continue
                w = f["widen"]["line"] - 1
                first = f["first"]["line"] - 1
                text = lines.pop(w)
                lines.insert(first, text + "        ; (moved up: tools/check_exec_spill.py --fix)")
            open(fix_out, "w").write("\n".join(lines))
            bad += left
        else:
            bad += len(fs)
    print(f"check_exec_spill: {bad} finding(s) in {len(paths)} file(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
