#!/usr/bin/env python3
"""Scan gfx950 device assembly for software-managed VALU hazards (the wait states the ISA leaves to the compiler) that are NOT honoured:

  A  VALU writes VGPR  -> DPP instruction reads it            : 2 wait states
  B  VALU writes VGPR  -> v_readlane / v_readfirstlane reads  : 1
  C  VALU writes VGPR  -> v_permlane16/32_swap reads / swaps  : 2
  D  VALU writes SGPR / VCC -> VALU reads it as an operand    : 2   (gfx90a / gfx940 family)
  E  VALU writes SGPR  -> v_readlane / v_writelane lane select: 4
  F  transcendental VALU result -> non-transcendental VALU    : 1
  G  VALU writes SGPR  -> vector memory instruction reads it  : 5   (descriptor, soffset, scalar base)

The compiler's hazard recogniser inserts s_nop for all of these when it knows the producer is a VALU instruction; an inline-asm VALU instruction is
opaque to it.  The window runs along the TEXT order: a label does not reset it (the fall-through path is a path), a branch target's other
predecessors are not followed; a function label does reset it."""
import re
import sys

TRANS = re.compile(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_")
TWO_DST = re.compile(r"v_(mad_u64_u32|mad_i64_i32|add_co_u32|sub_co_u32|subrev_co_u32|addc_co_u32|subb_co_u32|subbrev_co_u32|div_scale_f(32|64))")


def regs(tok, kind):
    out = set()
    for m in re.finditer(r"\b%s\[(\d+):(\d+)\]|\b%s(\d+)\b" % (kind, kind), tok):
        if m.group(1):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def scan(path):
    findings = []
    window = []      # (wait states since, vgpr defs, sgpr defs, is_trans, text, line)
    fn = "?"
    for ln, raw in enumerate(open(path), 1):
        t = raw.split(";")[0].strip()
        if not t:
            continue
        if t.endswith(":"):
            if not t.startswith(".L"):
                fn = t[:-1]
                window = []
            continue
        if t.startswith("."):
            continue
        op = t.split(None, 1)
        name = op[0]
        args = op[1] if len(op) > 1 else ""
        if name in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            # an unconditional transfer: the next line of TEXT is not what executes next (round 6: a v_readlane in front of an `s_branch` was paired
            # with the first load of the block that merely follows it in the file); a taken branch or a call outlasts every wait-state window
            window = []
            continue
        if name == "s_nop":
            n = int(args.strip() or 0) + 1
            window = [(w + n, a, b, c, d, e) for (w, a, b, c, d, e) in window]
            continue
        parts = [p.strip() for p in args.split(",")]
        is_valu = name.startswith("v_")
        if name.startswith(("buffer_", "global_", "scratch_", "flat_", "tbuffer_")):
            use_s = regs(args, "s")
            for (w, dv, ds, tr, txt, l0) in window:
                if w < 5 and (ds & use_s):
                    findings.append(("G", fn, l0, txt, ln, t))
        # a second (scalar) destination: carry out / scale flag
        two_dst = is_valu and (TWO_DST.match(name) is not None) and len(parts) >= 2 and (parts[1].startswith("s") or parts[1] == "vcc")
        if is_valu:
            srcs = ",".join(parts[2:] if two_dst else parts[1:])
            dpp = "dpp" in name or "quad_perm" in args or "row_" in args
            swap = "permlane" in name and "swap" in name
            rdlane = name.startswith("v_readlane") or name.startswith("v_readfirstlane")
            lanesel = (name.startswith("v_readlane") or name.startswith("v_writelane")) and len(parts) >= 3 and parts[2].startswith("s")
            use_v = regs(args if swap else srcs, "v")
            use_s = regs(srcs, "s")
            if "vcc" in srcs or name.endswith("_e32") and ("cndmask" in name or "addc" in name or "subb" in name or "div_fmas" in name):
                use_s.add(-1)
            for (w, dv, ds, tr, txt, l0) in window:
                if dpp and w < 2 and (dv & use_v):
                    findings.append(("A", fn, l0, txt, ln, t))
                if rdlane and w < 1 and (dv & use_v):
                    findings.append(("B", fn, l0, txt, ln, t))
                if swap and w < 2 and (dv & use_v):
                    findings.append(("C", fn, l0, txt, ln, t))
                if w < 2 and (ds & use_s):
                    findings.append(("D", fn, l0, txt, ln, t))
                if lanesel and w < 4 and (ds & regs(parts[2], "s")):
                    findings.append(("E", fn, l0, txt, ln, t))
                if tr and not TRANS.match(name) and w < 1 and (dv & use_v):
                    findings.append(("F", fn, l0, txt, ln, t))
        window = [(w + 1, a, b, c, d, e) for (w, a, b, c, d, e) in window if w + 1 < 6]
        if is_valu:
            dst = parts[0] if parts else ""
            dv = regs(dst, "v")
            ds = regs(dst, "s")
            if two_dst:
                ds |= regs(parts[1], "s")
                if parts[1] == "vcc":
                    ds.add(-1)
            if name.startswith("v_cmp") and name.endswith("_e32") or "vcc" in dst:
                ds.add(-1)
            if "permlane" in name and "swap" in name:
                dv |= regs(parts[1], "v")
            if name.startswith("v_readlane") or name.startswith("v_readfirstlane"):
                dv = set()
            window.append((0, dv, ds, bool(TRANS.match(name)), t, ln))
    return findings


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        f = scan(p)
        bad += len(f)
        kinds = {}
        for x in f:
            kinds[x[0]] = kinds.get(x[0], 0) + 1
        print(p, "findings:", kinds)
        for x in f[:40]:
            print("  %s in %s: line %d `%s` -> line %d `%s`" % (x[0], x[1][:40], x[2], x[3], x[4], x[5]))
    sys.exit(1 if bad else 0)
