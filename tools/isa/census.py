#!/usr/bin/env python3
"""Static instruction census of a line range of a kernel's assembly, per basic block and in total (round 6).
python tools/isa/census.py file.s first_line last_line [--blocks]"""
import re, sys
from collections import Counter, OrderedDict

def cat(op, line):
    if op.startswith("v_fma_f64") or op.startswith("v_add_f64") or op.startswith("v_mul_f64") or op.startswith("v_fmac_f64"): return "f64"
    if "_f64" in op and (op.startswith("v_cmp") or op.startswith("v_cmpx")): return "cmp"
    if "_f64" in op: return "f64other"
    if "dpp" in line or "row_" in line or "quad_perm" in line or op.startswith("v_permlane") or op.startswith("ds_swizzle") or op.startswith("ds_bpermute") or op.startswith("v_mov_b32_dpp"): return "dpp"
    if op.startswith("v_accvgpr"): return "acc_copy"
    if op.startswith("v_mov") or op.startswith("v_pk_mov"): return "v_mov"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith("v_cmp"): return "cmp"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"): return "lane_xfer"
    if op.startswith("v_"): return "v_int"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop") or op.startswith("s_sleep"): return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_swappc") or op.startswith("s_setpc") or op.startswith("s_getpc"): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_"): return "vmem"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("ds_"): return "lds"
    return "other"

def main():
    path, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    lines = open(path).read().split("\n")
    blocks = OrderedDict(); cur = f"@{a}"; blocks[cur] = Counter()
    tot = Counter()
    for i in range(a - 1, b):
        raw = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):", raw)
        if m: cur = f"{m.group(1)}@{i+1}"; blocks[cur] = Counter(); continue
        m = re.match(r"^; %bb\.(\d+):", raw)
        if m: cur = f"bb.{m.group(1)}@{i+1}"; blocks[cur] = Counter(); continue
        s = raw.split(";")[0].strip()
        if not s or s.startswith(".") or s.endswith(":"): continue
        op = s.split()[0]
        c = cat(op, s)
        blocks[cur][c] += 1; tot[c] += 1
    order = ["f64", "f64other", "cmp", "v_mov", "acc_copy", "cndmask", "dpp", "v_int", "lane_xfer", "salu", "smem", "branch", "waitcnt", "nop", "vmem", "scratch", "lds", "other"]
    if "--blocks" in sys.argv:
        print("block".ljust(22) + " ".join(o[:7].rjust(7) for o in order) + "   total")
        for k, c in blocks.items():
            n = sum(c.values())
            if n >= int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else n > 0:
                print(k.ljust(22) + " ".join(str(c[o]).rjust(7) for o in order) + f"   {n}")
    n = sum(tot.values())
    print("TOTAL".ljust(22) + " ".join(str(tot[o]).rjust(7) for o in order) + f"   {n}")
    valu = sum(tot[o] for o in ("f64", "f64other", "cmp", "v_mov", "acc_copy", "cndmask", "dpp", "v_int", "lane_xfer"))
    print(f"VALU {valu}  (f64 arithmetic {tot['f64']})  SALU {tot['salu'] + tot['smem']}  branch {tot['branch']}  mem {tot['vmem'] + tot['scratch'] + tot['lds']}")

if __name__ == "__main__":
    main()
