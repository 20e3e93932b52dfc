"""Aggregate a rocprofv3 PC-sampling CSV on the GPU box (the raw file is too large to travel): counts per instruction
address / text and per stall reason, for the kernels whose name contains --kernel.  python tools/pcsample_agg.py <dir> <out.json> [--kernel substr]"""
import csv, glob, json, os, sys, collections
d, out = sys.argv[1], sys.argv[2]
ksub = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else ""
res = {"files": []}
csv.field_size_limit(1 << 30)
for f in sorted(glob.glob(os.path.join(d, "**", "*"), recursive=True)):
    if os.path.isfile(f):
        res["files"].append([f, os.path.getsize(f)])
for f in sorted(glob.glob(os.path.join(d, "**", "*pc_sampling*.csv"), recursive=True)):
    with open(f, newline="") as fh:
        rd = csv.reader(fh)
        hdr = next(rd, None)
        if hdr is None:
            continue
        entry = {"file": f, "header": hdr, "head": [], "n": 0}
        cols = {c.lower(): i for i, c in enumerate(hdr)}
        def col(*names):
            for n in names:
                for c, i in cols.items():
                    if n in c:
                        return i
            return None
        i_inst, i_off, i_stall, i_kern = col("instruction_comment", "instruction"), col("offset", "pc"), col("stall", "reason"), col("kernel", "dispatch")
        i_inst2 = cols.get("instruction")
        by = collections.Counter(); by_stall = collections.Counter(); by_cols = collections.defaultdict(collections.Counter)
        for row in rd:
            entry["n"] += 1
            if len(entry["head"]) < 30:
                entry["head"].append(row)
            key = tuple(row[i] if i is not None and i < len(row) else "" for i in (i_off, i_inst2, i_inst))
            by[key] += 1
            for c, i in cols.items():
                if any(t in c for t in ("stall", "reason", "wave_issued", "inst_type", "type", "arb", "snapshot", "issued")) and i < len(row):
                    by_cols[c][row[i]] += 1
                    by_cols[c + "|" + "|".join(key[:2])][row[i]] += 0   # placeholder to keep the key list short
        entry["top"] = [[list(k), n] for k, n in by.most_common(1500)]
        entry["columns"] = {c: dict(v.most_common(40)) for c, v in by_cols.items() if "|" not in c}
        res.setdefault("csv", []).append(entry)
json.dump(res, open(out, "w"), indent=0)
print("pcsample_agg:", [(e["file"], e["n"]) for e in res.get("csv", [])])
