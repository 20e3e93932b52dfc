"""How many bytes MUST a NUTS draw move through HBM on this engine's mapping?  (a roofline that is a bound.)

The engine keeps the two live points of the leapfrog, sigma and mu on the compute unit and lets the tree's end points travel: the
bytes it moves are what its design chose to move, so "moved bytes / 8 TB/s" is a utilisation, not a fraction of a bound.  This tool
derives the bound: the NECESSARY HBM traffic of one draw = the unavoidable input / output of the chain (it is not resident between
draws at 4096 chains on 1024 block slots) + the tree end points that cannot stay on chip at the kernel's occupancy, under the BEST
possible choice of what to keep (Belady's MIN over the draw's exact access trace of chain vectors, with the on-chip capacity a
resident chain really has).

  capacity (vectors of D doubles a resident chain can hold on chip)
      registers: 512 VGPRs x 64 lanes x 4 B = 128 KiB per wavefront slot        = 128 KiB / (8 D)
      LDS:       160 KiB per CU / resident chains per CU                        = 160 KiB / occ / (8 D)
      minus the leapfrog's own working set: cur (z, v, g) + new (z, v, g) + sigma + mu = 8 vectors
  trace (vector granularity, one entry per read / write of a whole chain vector, exactly the tree of nuts_kernels.hpp / src/nuts.rs:94-388):
      leaf n of a doubling at depth d: the new point (z, v, g) is produced on chip;
      odd n: the U-turn tests of the levels k = 2 .. t it closes read A.first (F), A.last (L[k-1]) and B.first (F[k-1]; level 2: on chip),
             z and v each; F[.] (z, v) is written for the first leaf of every sub-tree of level >= 2, L[t] (z, v) for the pending
             sub-tree's last leaf, one candidate z per pending sub-tree;
      end of a doubling: the top-level tests read both main-tree edges (z, v) and the sub-tree's first leaf; the new edge (z, v, g)
             is written; the next doubling in the other direction reads its edge (z, v, g);
      end of the draw: the chosen candidate's z is read.
  input / output per draw (never avoidable): read z, g_z, sigma, mu (the block switches chains every draw), write z, g_z, write the
      position row (D doubles) and the statistics row (192 B).

A read of a vector that is not on chip costs 8 D bytes; a vector evicted while it will still be read costs 8 D bytes (written back once).
Vectors that are never read again are dropped for free.  The result is a LOWER bound for any kernel that evaluates this tree with this
much on-chip memory per chain: it can only be beaten by changing the occupancy (fewer resident chains) — which the latency-bound
kernel cannot afford — or the algorithm.

  python tools/necessary_traffic.py [--dim 1024] [--occ 4] [--out profiles/r04_necessary_traffic.json]
"""
import argparse
import json
import random


def draw_trace(depth, rng, turning_last=True):
    """Access trace of one draw whose tree reaches `depth` (all doublings complete; directions random).
    Entries: ('r', name) / ('w', name); names identify chain vectors; 'new' points live in the working set (not traced)."""
    tr = []
    edges = {"L": "E0", "R": "E0"}            # edge ids per side; E0 = the initial point (z, v, g on chip at the draw's start)
    for comp in "zvg":
        tr.append(("w", ("E0", comp)))          # the initial point becomes edge 0 (it is both the chain's state and the tree's root)
    next_edge = 1
    last_dir, last_is_edge = None, False
    for d in range(depth):
        side = "R" if rng.random() < 0.5 else "L"
        nleaf = 1 << d
        if d > 0 and not (last_is_edge and last_dir == side):
            for comp in "zvg":
                tr.append(("r", (edges[side], comp)))
        for n in range(nleaf):
            if n & 1:
                t = (~n & (n + 1)).bit_length() - 1          # trailing ones of n
                for k in range(2, t + 1):
                    a_first = n + 1 - (1 << k)
                    fa = d if a_first == 0 else (a_first & -a_first).bit_length() - 1
                    for nm in (("F", d, fa), ("Lk", d, k - 1)) + ((("F", d, k - 1),) if k > 2 else ()):
                        tr.append(("r", (nm, "z"))); tr.append(("r", (nm, "v")))
                ne = n - 1
                if (ne & 3) == 0 and d > 1:
                    nm = ("F", d, d if ne == 0 else (ne & -ne).bit_length() - 1)
                    tr.append(("w", (nm, "z"))); tr.append(("w", (nm, "v")))
                if n + 1 < nleaf:
                    nm = ("Lk", d, t)
                    tr.append(("w", (nm, "z"))); tr.append(("w", (nm, "v")))
                    tr.append(("w", (("C", d, t), "z")))     # the pending sub-tree's candidate
        if d >= 1:                                          # top-level tests: both edges, the sub-tree's first leaf
            for e in (edges["L"], edges["R"]):
                tr.append(("r", (e, "z"))); tr.append(("r", (e, "v")))
            if d > 1:
                tr.append(("r", (("F", d, d), "z"))); tr.append(("r", (("F", d, d), "v")))
        tr.append(("w", (("C", "main", d), "z")))            # the merged candidate of the main tree (half of the time a new one)
        if d + 1 < depth or not turning_last:
            e = "E%d" % next_edge
            next_edge += 1
            for comp in "zvg":
                tr.append(("w", (e, comp)))
            edges[side] = e
            last_dir, last_is_edge = side, True
    tr.append(("r", (("C", "main", depth - 1), "z")))        # the chosen point
    return tr


def belady(trace, capacity):
    """(vector reads from HBM, vector write-backs to HBM) under MIN replacement with `capacity` on-chip vectors."""
    nxt = {}
    next_use = [None] * len(trace)
    for i in range(len(trace) - 1, -1, -1):
        op, name = trace[i]
        next_use[i] = nxt.get(name)
        if op == "r":
            nxt[name] = i
        else:
            nxt.pop(name, None)                 # a write kills the old value: earlier copies need not survive it
    INF = 10 ** 9
    onchip = {}                                 # name -> next read index (INF: never read again)
    in_hbm = set()
    loads = stores = 0

    def make_room():
        nonlocal stores
        while len(onchip) > capacity:
            dead = [k for k, v in onchip.items() if v >= INF]
            victim = dead[0] if dead else max(onchip, key=onchip.get)
            if onchip[victim] < INF and victim not in in_hbm:
                stores += 1
                in_hbm.add(victim)
            del onchip[victim]

    for i, (op, name) in enumerate(trace):
        nu = next_use[i] if next_use[i] is not None else INF
        if op == "w":
            in_hbm.discard(name)
            onchip[name] = nu
        else:
            if name not in onchip:
                loads += 1
            onchip[name] = nu
        if onchip[name] >= INF:
            del onchip[name]
        make_room()
    return loads, stores


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--occ", type=int, default=4, help="resident chains per CU (K2: one 64-lane wavefront per SIMD)")
    ap.add_argument("--trials", type=int, default=400)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    vec = 8 * a.dim
    cap_regs = 128 * 1024 // vec
    cap_lds = 160 * 1024 // a.occ // vec
    capacity = max(0, cap_regs + cap_lds - 8)
    rng = random.Random(1)
    table = {}
    for depth in range(1, 11):
        lo = st = 0
        n = max(20, a.trials >> max(0, depth - 5))
        for _ in range(n):
            l_, s_ = belady(draw_trace(depth, rng), capacity)
            lo += l_; st += s_
        table[depth] = {"tree_vector_loads": lo / n, "tree_vector_stores": st / n, "leaves": (1 << depth) - 1}
    io_vectors = 4 + 2 + 1                      # read z, g_z, sigma, mu; write z, g_z; write the position row
    out = {"dim": a.dim, "vector_bytes": vec, "resident_chains_per_cu": a.occ,
           "capacity_vectors": {"registers": cap_regs, "lds": cap_lds, "working_set": 8, "for_tree_end_points": capacity},
           "io_bytes_per_draw": io_vectors * vec + 192, "io_vectors_per_draw": io_vectors,
           "per_depth": {str(k): dict(v, necessary_bytes_per_draw=io_vectors * vec + 192 + (v["tree_vector_loads"] + v["tree_vector_stores"]) * vec)
                         for k, v in table.items()},
           "note": "Belady MIN over the draw's vector access trace (tools/necessary_traffic.py): a lower bound for this tree at this on-chip capacity"}
    print(json.dumps(out, indent=1))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
