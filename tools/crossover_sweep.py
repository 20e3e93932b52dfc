"""Which kernel family is fastest for small chains, by density class, dim bucket and chain count: the measurements behind
nm_engine_create's selection rule (csrc/nuts_engine.hip, "measured crossovers").  Families: one chain per wavefront (wave), 8 / 4 / 2 chains per
wavefront (group), one chain per lane (lane, dim <= 10).  For every (density, dim, chains): post-warm-up leapfrogs/s of each family that applies and
what the automatic rule picked.  One JSON document on stdout / --out.

  python tools/crossover_sweep.py [--out profiles/r05_crossovers.json] [--draws 100] [--quick]"""
import argparse
import json
import os
import sys

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuts_rs_amd as N  # noqa: E402

DENS = {
    "iid": lambda d: N.LogpSpec.iid_normal(d, 3.0),
    "diag": lambda d: N.LogpSpec.diag_normal(np.exp(np.linspace(-1, 1, d))),
    "funnel": lambda d: N.LogpSpec.funnel(d),
    "schools": lambda d: N.LogpSpec.eight_schools(),
}
FAMILIES = {"wave": dict(lane_groups=1, lane_chains=1), "group": dict(lane_groups=2, lane_chains=1), "lane": dict(lane_chains=2), "auto": dict()}


def rate(logp, chains, kw, tune, draws):
    s = N.DiagNutsSettings(num_chains=chains, seed=20260930, num_tune=tune, num_draws=draws)
    b = N.ChainBatch(s, logp, chains, **kw)
    try:
        b.set_position(b.init_positions_uniform())
        b.draw_device(tune)
        b.reset_counters()
        b.draw_device(draws)
        c = b.counters()
        fam = "lane" if b.lane_launches() else "group" if b.group_launches() else "wave"
        return c["total_leapfrogs"] / (c["kernel_ms"] * 1e-3), fam
    finally:
        b.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--draws", type=int, default=100)
    ap.add_argument("--tune", type=int, default=150)
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    cases = [("iid", 4), ("iid", 10), ("diag", 8), ("funnel", 10), ("schools", 10), ("iid", 16), ("diag", 30), ("funnel", 50)]
    counts = [1024, 4096, 8192, 16384, 32768, 65536] if not a.quick else [4096, 32768]
    rows = []
    for dens, dim in cases:
        for chains in counts:
            row = {"density": dens, "dim": dim, "chains": chains, "leapfrogs_per_s": {}}
            for fam, kw in FAMILIES.items():
                if fam == "lane" and dim > 10:
                    continue
                try:
                    r, ran = rate(DENS[dens](dim), chains, kw, a.tune, a.draws)
                except Exception as e:  # noqa: BLE001
                    row["leapfrogs_per_s"][fam] = f"error: {e}"
                    continue
                if fam == "auto":
                    row["auto_picked"] = ran
                    row["auto_leapfrogs_per_s"] = r
                elif ran == fam:
                    row["leapfrogs_per_s"][fam] = r
            num = {k: v for k, v in row["leapfrogs_per_s"].items() if isinstance(v, float)}
            if num:
                row["best"] = max(num, key=num.get)
                row["auto_over_best"] = row.get("auto_leapfrogs_per_s", 0.0) / num[row["best"]]
            rows.append(row)
            print(json.dumps(row), file=sys.stderr, flush=True)
    doc = {"tool": "tools/crossover_sweep.py", "draws": a.draws, "tune": a.tune, "rows": rows,
           "worst_auto_over_best": min((r["auto_over_best"] for r in rows if "auto_over_best" in r), default=None)}
    text = json.dumps(doc, indent=1)
    if a.out:
        open(a.out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
