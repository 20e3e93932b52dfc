"""Phase breakdown of one draw on one block (needs a build with -DNM_PROF=1; see NM_MARK in nuts_kernels.hpp).

  python tools/prof_phases.py [chains] [dim] [tune] [draws]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch  # noqa: F401  (its HIP runtime must initialise first)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuts_rs_amd as N  # noqa: E402
from nuts_rs_amd import _lib  # noqa: E402

NAMES = {0: "between draws (store stats, loop)", 1: "momentum refresh", 2: "tree (all doublings)",
         3: "winner x / g_x / stores / fisher", 4: "adapt", 5: "stats + mass-matrix event",
         16: " tree: pair-loop head", 17: " tree: leapfrog (even leaf)", 18: " tree: account + F store (even)",
         19: " tree: leapfrog (odd leaf)", 20: " tree: account (odd)", 21: " tree: level-1 merge (regs)",
         22: " tree: level>=2 merges + U-turn loads", 23: " tree: pending sub-tree stores", 24: " tree: doubling head (rng bool)",
         25: " tree: sub-tree done", 26: " tree: top-level U-turn tests", 27: " tree: depth-0 leaf",
         28: " tree: top-level merge + edge store",
         13: " (before merge_weights)", 14: " merge: logaddexp (exp + log1p)", 15: " merge: exp for Bernoulli",
         29: " merge: Bernoulli draw",
         8: " refresh: ChaCha words -> LDS", 9: " refresh: fast-path tests", 10: " refresh: walk", 12: " refresh: parallel slow paths",
         11: " refresh: scatter + barrier"}


def main():
    chains = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    tune = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    draws = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    s = N.DiagNutsSettings(num_chains=chains, seed=20260928, num_tune=tune, num_draws=draws)
    # PROF_LOGP=funnel: Neal's funnel at `dim`; PROF_FIXED_STEP=<eps>: fixed step size without jitter (deep trees on demand);
    # PROF_MAXDEPTH=<d>
    if os.environ.get("PROF_FIXED_STEP"):
        st = s.adapt_options.step_size_settings
        st.method, st.fixed_step_size, st.jitter = N.sampler.STEP_FIXED, float(os.environ["PROF_FIXED_STEP"]), None
    if os.environ.get("PROF_MAXDEPTH"):
        s.maxdepth = int(os.environ["PROF_MAXDEPTH"])
    logp = N.LogpSpec.funnel(dim) if os.environ.get("PROF_LOGP") == "funnel" else N.LogpSpec.iid_normal(dim, 3.0)
    b = N.ChainBatch(s, logp, chains)
    b.set_position(b.init_positions_uniform())
    L = _lib.load()
    L.nm_debug_read_prof.argtypes = [C.c_void_p, C.c_void_p]
    buf = np.zeros(32, dtype=np.uint64)
    L.nm_debug_read_prof(b._h, buf.ctypes.data)      # discard what set_position's kernel left there
    b.draw_device(tune)
    c0 = b.counters()
    L.nm_debug_read_prof(b._h, buf.ctypes.data)
    report("warm-up", buf, tune * (-(-chains // 1024) if chains > 1024 else 1), c0["kernel_ms"], c0["total_leapfrogs"] / tune / chains)
    b.reset_counters()
    b.draw_device(draws)
    c = b.counters()
    L.nm_debug_read_prof(b._h, buf.ctypes.data)
    report("sampling", buf, draws * (-(-chains // 1024) if chains > 1024 else 1), c["kernel_ms"], c["total_leapfrogs"] / draws / chains)


def report(title, buf, ndraw, ms, steps_per_draw):
    total = buf.sum()
    cyc_per_us = total / (ms * 1e3)                                   # block 0 is busy for the whole launch
    print(f"== {title}: kernel {ms:.2f} ms, block 0: {ndraw} draws, {total} cycles -> {cyc_per_us:.0f} cycles/us, {steps_per_draw:.2f} steps/draw")
    for k in range(32):
        if buf[k]:
            print(f"  [{k}] {NAMES.get(k, '?'):40s} {buf[k] / ndraw / cyc_per_us:8.2f} us/draw  ({100.0 * buf[k] / total:5.1f} %)")


if __name__ == "__main__":
    main()
