"""Scan device assembly (hipcc --cuda-device-only -S) for the gfx950 store-data hazard LLVM does not cover: a buffer store of more
than 64 bits whose soffset is an SGPR, followed within `--window` instructions by a VALU write to one of its data VGPRs
(nuts_kernels.hpp store_guard).  Prints every instance; exit code 1 if any.

  python tools/check_store_hazard.py file.s [--window 2] [--mubuf64]

--mubuf64 (round 3): additionally require every 64-bit MUBUF store (buffer_store_dwordx2 ... offen: the lane kernel's scratch
addressing, any soffset) to keep its data VGPRs unwritten for `--window` wait states — the guard of LCtx::bst."""
import re
import sys

path = sys.argv[1]
window = int(sys.argv[sys.argv.index("--window") + 1]) if "--window" in sys.argv else 2
m64 = "--mubuf64" in sys.argv
store = re.compile(r"^\s*buffer_store_dwordx(" + ("[234]" if m64 else "[34]") + r")\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(s\d+|\d+|0x[0-9a-f]+|off)")
wr = re.compile(r"^\s*(v_\w+)\s+(v\[(\d+):(\d+)\]|v(\d+))")
lines = [l.rstrip("\n") for l in open(path)]
instr = [(i, l) for i, l in enumerate(lines) if re.match(r"^\s+[a-z]", l) and not l.strip().startswith((".", ";"))]
bad = 0
kernel = "?"
names = {i: l.split(":")[0] for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)}
for k, (i, l) in enumerate(instr):
    m = store.match(l)
    if not m or not (m.group(5).startswith("s") or m.group(1) == "2"):      # (LLVM covers > 64 bits with an immediate soffset; nothing covers 64 bits)
        continue
    lo, hi = int(m.group(2)), int(m.group(3))
    states = 0
    for j in range(k + 1, min(k + 1 + 4, len(instr))):
        t = instr[j][1].strip()
        if t.startswith("s_nop"):
            states += int(t.split()[1]) + 1
            continue
        w = wr.match(instr[j][1])
        if w and not t.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            a = int(w.group(3)) if w.group(3) else int(w.group(5))
            b = int(w.group(4)) if w.group(4) else a
            if a <= hi and b >= lo and states < window:
                bad += 1
                kn = [n for n in names if n <= i]
                print(f"{path}:{i + 1}: {l.strip()}  -->  {t}   (kernel {names[max(kn)] if kn else '?'})")
        states += 1
        if states >= window:
            break
print(f"{path}: {bad} unguarded store-data hazards")
sys.exit(1 if bad else 0)
