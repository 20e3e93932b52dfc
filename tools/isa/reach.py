"""Reaching definitions on a gfx950 kernel's assembly text (scalar CFG; EXEC is ignored): which writes of a VGPR can reach a given use.
python tools/isa/reach.py file.s <kernel-symbol-substring> <use-line> <vgpr-number>
Round 6, DESIGN §22: looking for the lost update of the main tree's log_size in the failing build of 8e9c172."""
import re, sys
from collections import defaultdict

def parse(path, sym):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and l.rstrip().endswith(":") or (l.startswith("_Z") and sym in l and ":" in l and "@" in l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines, start, end

def regs_of(tok):
    """v5 -> [5]; v[4:7] -> [4,5,6,7]"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m: return [int(m.group(1))]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m: return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []

STORE = re.compile(r"^(scratch_store|buffer_store|global_store|flat_store|ds_write|ds_store|s_|v_cmp_|v_cmpx|v_readlane|v_readfirstlane|buffer_atomic|global_atomic|ds_add|ds_max|ds_min|s_waitcnt|v_nop)")
def defs_uses(ins):
    """(defs, uses) VGPR sets of one instruction line (text level)."""
    ins = ins.split(";")[0].strip()
    if not ins or ins.endswith(":") or ins.startswith("."): return set(), set()
    parts = ins.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in re.split(r",(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
    toks = []
    for o in ops:
        o = o.split()[0] if o else o
        o = o.lstrip("-|").rstrip("|")
        toks.append(o)
    d, u = set(), set()
    if op == "s_swappc_b64":
        return set(range(0, 40)) | set(range(48, 56)), set(range(0, 32))   # caller-saved clobbers (approx.)
    if STORE.match(op) and not op.startswith("v_readlane") and not op.startswith("v_readfirstlane") and not op.startswith("v_cmp"):
        for t in toks: u |= set(regs_of(t))
        return d, u
    if op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        for t in toks: u |= set(regs_of(t))
        return d, u
    if op.startswith("ds_read") or op.startswith("ds_load") or op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle"):
        d |= set(regs_of(toks[0]))
        for t in toks[1:]: u |= set(regs_of(t))
        return d, u
    if toks:
        d |= set(regs_of(toks[0]))
        for t in toks[1:]: u |= set(regs_of(t))
        if op.startswith("v_permlane") and "swap" in op and len(toks) > 1:
            d |= set(regs_of(toks[1]))
        if op.startswith("v_swap") and len(toks) > 1: d |= set(regs_of(toks[1]))
        if op.startswith("v_writelane") or op.startswith("v_mac") or op.startswith("v_fmac") or "mfma" in op or op.startswith("v_pk_fmac"):
            u |= set(regs_of(toks[0]))
    return d, u

def build(lines, start, end):
    # blocks: list of (first_line, last_line); label -> block index
    leaders = {start + 1}
    label_at = {}
    for i in range(start, end):
        l = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: label_at[m.group(1)] = i; leaders.add(i)
        s = l.strip()
        if s.startswith("s_cbranch") or s.startswith("s_branch") or s.startswith("s_endpgm") or s.startswith("s_setpc"):
            leaders.add(i + 1)
    ls = sorted(x for x in leaders if x < end)
    blocks = [(ls[k], (ls[k + 1] if k + 1 < len(ls) else end) - 1) for k in range(len(ls))]
    idx_of = {b[0]: k for k, b in enumerate(blocks)}
    succ = defaultdict(list)
    for k, (a, b) in enumerate(blocks):
        last = None
        for i in range(b, a - 1, -1):
            s = lines[i].split(";")[0].strip()
            if s and not s.endswith(":") and not s.startswith("."): last = s; break
        ft = True
        if last:
            op = last.split()[0]
            if op == "s_branch": succ[k].append(idx_of[label_at[last.split()[1]]]); ft = False
            elif op.startswith("s_cbranch"): succ[k].append(idx_of[label_at[last.split()[1]]])
            elif op in ("s_endpgm", "s_setpc_b64"): ft = False
        if ft and k + 1 < len(blocks): succ[k].append(k + 1)
    return blocks, succ

def main():
    path, sym, use_line, reg = sys.argv[1], sys.argv[2], int(sys.argv[3]) - 1, int(sys.argv[4])
    lines, start, end = parse(path, sym)
    blocks, succ = build(lines, start, end)
    pred = defaultdict(list)
    for k, ss in succ.items():
        for s in ss: pred[s].append(k)
    blk = next(k for k, (a, b) in enumerate(blocks) if a <= use_line <= b)
    # walk backwards from the use
    found = {}
    seen = set()
    work = [(blk, use_line - 1)]
    while work:
        k, frm = work.pop()
        a, b = blocks[k]
        hit = False
        for i in range(frm, a - 1, -1):
            d, _ = defs_uses(lines[i])
            if reg in d:
                found[i + 1] = lines[i].strip(); hit = True; break
        if hit: continue
        for p in pred[k]:
            if p not in seen:
                seen.add(p); work.append((p, blocks[p][1]))
        if not pred[k]: found[f"entry@{a+1}"] = "(function entry: undefined)"
    for k in sorted(found, key=lambda x: (isinstance(x, str), x)): print(k, found[k])

if __name__ == "__main__":
    main()
