"""BASELINE config 5 on one GPU: N(0, Sigma) with a FULL Sigma at dim 256 x 4096 chains, sampled through the exact dense
preconditioner — the nearest reference semantics of a "dense mass matrix" (SURVEY §8(d) K5): the low-rank transformation
of src/transform/low_rank.rs with rank = dim, F(y) = (I + U (L^1/2 - I) U') y, Sigma = U L U'.

  python tools/bench_k5.py [--mode per_chain|shared] [--chains 4096] [--dim 256] [--rank 256] [--tune 100] [--draws 100]

One JSON line: leapfrogs/s, M1 = leapfrog-steps*dims/s, the f64 flop rate of the dense products (per leapfrog: the two
applications of the transformation = 4 dim x rank products, plus the density's dim x dim one), HBM/L2 traffic model.
  per_chain : every chain owns its (U, lambda) — what per-chain adaptation produces (parity mode; GEMV, HBM-bound)
  shared    : one transformation for all chains (nm_engine_set_transform per_chain = 0)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (initialises the HIP runtime first)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuts_rs_amd as N  # noqa: E402


def target(dim, seed=55, rank=8, scale=100.0):
    rng = np.random.default_rng(seed)
    u = np.linalg.qr(rng.normal(size=(dim, rank)))[0]
    sigma = np.eye(dim) + u @ np.diag(rng.uniform(5.0, scale, rank)) @ u.T
    sc = np.exp(rng.normal(0, 0.5, dim))
    sigma = np.diag(sc) @ sigma @ np.diag(sc)
    prec = np.linalg.inv(sigma)
    return (prec + prec.T) / 2, sigma


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=4096)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--rank", type=int, default=0, help="eigenvectors kept (0 = dim: the dense case)")
    ap.add_argument("--tune", type=int, default=100)
    ap.add_argument("--draws", type=int, default=100)
    ap.add_argument("--mode", default="per_chain")
    a = ap.parse_args()
    D, C = a.dim, a.chains
    r = a.rank or D
    prec, sigma = target(D)
    w, u = np.linalg.eigh(sigma)
    keep = np.argsort(np.abs(np.log(w)))[::-1][:r]            # the r eigenvalues furthest from 1
    tr = (np.ones(D), np.zeros(D), w[keep], np.ascontiguousarray(u[:, keep].T), np.zeros(D))
    s = N.LowRankNutsSettings(num_chains=C, seed=20260928, num_tune=a.tune, num_draws=a.draws, freeze_transform=True)
    b = N.ChainBatch(s, N.LogpSpec.mvn_precision(prec), C, lowrank_max_rank=r, chain_tiles=0 if a.mode == "shared" else 1)
    b.set_position(b.init_positions_uniform())
    t = time.time()
    b.set_transform(*tr)
    t_up = time.time() - t
    t = time.time()
    b.draw_device(a.tune)
    t_tune = time.time() - t
    b.reset_counters()
    st_dev = torch.empty((a.draws, C, N.STATS_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    pos_dev = torch.empty((a.draws, C, D), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()       # (allocations / fills on torch's stream are done before the engine's stream writes)
    t = time.time()
    b.draw_device(a.draws, pos_dev.data_ptr(), st_dev.data_ptr())
    dt = time.time() - t
    c = b.counters()
    st = st_dev.cpu().numpy().view(N.STATS_DTYPE).reshape(a.draws, C)
    steps = int(st["n_steps"].sum())
    z = pos_dev[-20:].reshape(-1, D).cpu().numpy() @ (u / np.sqrt(w))
    kern_s = c["kernel_ms"] * 1e-3
    flop_per_leapfrog = 2.0 * (4 * D * r + D * D)            # 2 flop per fma: U'v and U s for x and for g_z, P x for the density
    out = {"config": f"K5: N(0, Sigma) full Sigma dim {D} x {C} chains, low-rank transformation rank {r} ({a.mode}), frozen; "
                     f"step size adapted over {a.tune} draws",
           "mode": a.mode, "matrix_core_launches": b.tile_launches(), "leapfrogs_per_s": steps / dt, "M1_steps_dims_per_s": steps * D / dt, "draws_per_s_per_chain": a.draws / dt,
           "leapfrogs_per_draw": steps / (a.draws * C), "mean_depth": float(st["depth"].mean()), "step_size_mean": float(st["step_size"][-1].mean()),
           "divergence_rate": float(st["diverging"].mean()), "kernel_ms": c["kernel_ms"], "wall_s": dt, "tune_s": t_tune, "upload_s": t_up,
           "f64_dense_TFLOPs": steps * flop_per_leapfrog / kern_s / 1e12,
           "matrix_bytes_per_leapfrog": 8.0 * (2 * D * r + D * D), "matrix_read_TBps": steps * 8.0 * (2 * D * r + D * D) / kern_s / 1e12,
           "whitened_draws": {"mean": float(z.mean()), "var": float(z.var())}}
    if os.environ.get("NM_TILE_PROF"):     # a -DNM_TILE_PROF=1 build: cycles per wavefront in chain code / barrier waits / matrix-core products
        import ctypes as Ct
        from nuts_rs_amd import _lib
        L = _lib.load()
        L.nm_debug_read_prof.argtypes = [Ct.c_void_p, Ct.c_void_p]
        buf = np.zeros(32, dtype=np.uint64)
        L.nm_debug_read_prof(b._h, buf.ctypes.data)
        chain_, wait_, mma_, rounds_, idle_ = (float(buf[i]) for i in (16, 17, 18, 19, 20))
        tot = chain_ + wait_ + mma_ + idle_
        out["tile_prof"] = {"note": "both launches (tune + timed); s_memtime cycles summed over all wavefronts",
                            "chain_code_frac": chain_ / tot, "barrier_wait_frac": wait_ / tot, "matrix_product_frac": mma_ / tot,
                            "idle_column_between_rounds_frac": idle_ / tot, "rendezvous_of_wave0": rounds_,
                            "cycles_per_wave_round": tot / max(rounds_, 1.0) / 16.0}
    if os.environ.get("NM_LOCK_PROF"):     # a -DNM_LOCK_PROF=1 build of kern_lockstep.hip: where does a round of wavefront 0, block 0 go?
        import ctypes as Ct
        from nuts_rs_amd import _lib
        L = _lib.load()
        L.nm_debug_read_prof.argtypes = [Ct.c_void_p, Ct.c_void_p]
        buf = np.zeros(32, dtype=np.uint64)
        L.nm_debug_read_prof(b._h, buf.ctypes.data)
        names = ["logic (P2) / refresh", "wait: barrier behind P2", "unit hand-out + barrier", "P3", "wait: barrier before the products", "five products", "P1 + stripe reductions"]
        tot = float(buf[:7].sum())
        rounds = float(buf[8]) or 1.0
        out["lock_prof"] = {"rounds_block0": rounds, "cycles_per_round": tot / rounds, "note": "s_memtime ticks = shader cycles (2.4 GHz) of wavefront 0 of block 0, both launches",
                            "us_per_round": tot / rounds / 2400.0,
                            "phases": {n: float(buf[i]) / tot for i, n in enumerate(names)}}
    print(json.dumps(out))
    b.close()


if __name__ == "__main__":
    main()
