#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on BASELINE.json's headline configuration (K2), one JSON line on rank 0.

  metric  : leapfrog-steps x dims / second (M1 = sum_chains sum_draws n_steps * D / wall seconds), post-warm-up
            steady state; draws/sec/chain (M2) is reported beside it.
  workload: K2 — iid N(3,1), dim 1024, 4096 chains PER GPU, maxdepth 10, per-chain diagonal mass matrix,
            DiagNutsSettings defaults with num_tune 400; x0 ~ U(-1,1) from each chain's generator, seed 20260928.
  step    : one NUTS draw of every chain (one pass of the hot path over the batch of 4096 chains).
            Setup (untimed, reported as `adaptation`): engine creation, set_position, the 400 tuning draws.
            Then W untimed post-warm-up steps, then EXACTLY K timed steps between barrier + device sync.
  N GPUs  : one process per GPU (torch.distributed, RCCL backend); chains shard with no data-path collective
            (global chain id = rank * 4096 + local id; results are invariant to the partition) => "weak" scaling;
            value = all ranks' steps*dims / max-over-ranks time.
The state is resident in HBM before the timed region (positions are uploaded in set_position; draws stay on the
device), so `value` contains no PCIe traffic.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E datasheet peak (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0          # measured float4 copy on MI355X per the same guide
ALGO_BYTES_PER_STEP_DIM = 64   # read z,v,g_z,sigma,mu + write z',v',g_z' in f64, per (step x dim)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--chains", type=int, default=4096, help="chains per GPU")
    p.add_argument("--dim", type=int, default=1024)
    p.add_argument("--num-tune", type=int, default=400)
    p.add_argument("--seed", type=int, default=20260928)
    p.add_argument("--dims-per-lane", type=int, default=0)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-chains", type=int, default=0, help="chains of the bounded CPU sample (0 = 2 per host core)")
    p.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one rank per GPU) | gloo (test rigs: ranks may share a GPU)")
    return p.parse_args()


def pmc_traffic(args):
    """HBM bytes per timed launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of
    this same command, tools/pmc_run.sh; gfx950 correction: FETCH_SIZE x2 for 16 B/lane streams, per
    MI355X_MICROARCH.md).  Counters cannot be collected from inside an un-profiled run, so the latest committed
    summary under profiles/ is reported when it was taken on the same workload; otherwise null."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_k2.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload", {"chains": 4096, "dim": 1024, "steps": 200}) == {"chains": args.chains, "dim": args.dim, "steps": args.steps}:
            best = (f, d)
    if not best:
        return None, None
    f, d = best
    return d.get("hbm_bytes_per_launch"), os.path.relpath(f, ROOT)


def cpu_baseline(args, cores):
    """The CPU restatement of nuts-rs (oracle/, reference arithmetic: libm + SIMD-order sums) on this box's host
    cores: one chain per task over `cores` threads (the reference's Rayon structure, src/sampler.rs:1116), same
    density / settings / seeds, a bounded sample of the same workload.  Labelled "port": it is NOT nuts-rs itself
    (no Rust toolchain here or on the GPU box)."""
    from oracle import oracle as O
    n = args.cpu_chains or 2 * cores
    draws = 25
    s = O.default_settings(seed=args.seed, num_tune=args.num_tune, num_chains=n)
    x0 = O.init_positions_uniform(args.seed, 0, n, args.dim)
    t0 = time.time()
    r = O.run_timed(s, O.LOGP_IID_NORMAL, args.dim, [3.0], O.ref_cfg(), n, x0, args.num_tune, draws, n_threads=cores)
    wall = time.time() - t0
    rate = r["steps"] * args.dim / (r["cpu_seconds"] / cores) if r["cpu_seconds"] > 0 else 0.0
    return {
        "value": rate, "unit": "leapfrog-steps*dims/s", "cores": cores, "kind": "port",
        "sample": f"CPU restatement of nuts-rs (oracle/, not nuts-rs): {n} chains x dim {args.dim}, same settings/seed, "
                  f"{args.num_tune} warm-up draws untimed then {draws} timed draws per chain, one chain per task on "
                  f"{cores} threads; {r['steps']} steps in {r['cpu_seconds']:.2f} cpu-s ({wall:.1f} s wall incl. warm-up)",
        "warmup_value": r["warm_steps"] * args.dim / (r["warm_cpu_seconds"] / cores) if r["warm_cpu_seconds"] > 0 else 0.0,
    }


def main():
    args = parse()
    import torch
    import nuts_rs_amd as N

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.dist_backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
    device_index = local_rank % ndev
    torch.cuda.set_device(device_index)
    red_dev = "cuda" if args.dist_backend == "nccl" else "cpu"
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.dist_backend == "nccl":
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist_mod.init_process_group(args.dist_backend)
        dist = dist_mod

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    C_, D = args.chains, args.dim
    settings = N.DiagNutsSettings(num_chains=C_ * world, seed=args.seed, num_tune=args.num_tune,
                                  num_draws=args.steps + args.warmup)
    batch = N.ChainBatch(settings, N.LogpSpec.iid_normal(D, 3.0), C_, chain_id_offset=rank * C_,
                         device=device_index, dims_per_lane=args.dims_per_lane)
    x0 = batch.init_positions_uniform()
    status = batch.set_position(x0)
    assert (status == 0).all()
    # ---- adaptation phase (untimed setup; reported)
    barrier()
    t0 = time.perf_counter()
    batch.draw_device(args.num_tune)
    barrier()
    t_tune = time.perf_counter() - t0
    c_tune = batch.counters()
    # ---- W untimed post-warm-up steps
    if args.warmup:
        batch.draw_device(args.warmup)
    batch.reset_counters()
    # ---- K timed steps
    barrier()
    t0 = time.perf_counter()
    batch.draw_device(args.steps, sync=False)
    batch.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    c = batch.counters()
    steps_local = c["total_leapfrogs"]

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_max = float(t.item())
        sv = torch.tensor([float(steps_local), float(c_tune["total_leapfrogs"]), t_tune], dtype=torch.float64, device=red_dev)
        sv_max = sv.clone()
        dist.all_reduce(sv, op=dist.ReduceOp.SUM)
        dist.all_reduce(sv_max, op=dist.ReduceOp.MAX)
        steps_total, tune_steps_total, t_tune_max = float(sv[0]), float(sv[1]), float(sv_max[2])
    else:
        elapsed_max, steps_total, tune_steps_total, t_tune_max = elapsed, float(steps_local), float(c_tune["total_leapfrogs"]), t_tune

    if rank == 0:
        value = steps_total * D / elapsed_max
        # roofline of the dominant kernel (nuts_draw_kernel), rank 0's launch, timed with HIP events on the
        # engine's own stream inside libnuts_amd (nm_engine_get_counters)
        kern_s = c["kernel_ms"] * 1e-3
        algo_bytes = steps_local * D * ALGO_BYTES_PER_STEP_DIM
        achieved = algo_bytes / kern_s / 1e9
        traffic, traffic_src = pmc_traffic(args)
        out = {
            "metric": "leapfrog-steps*dims/sec at 4096 chains x dim 1024 (post-warm-up NUTS draws)",
            "value": value, "unit": "leapfrog-steps*dims/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"K2: iid N(3,1) dim {D} x {C_} chains per GPU, maxdepth 10, per-chain diag mass matrix, "
                                   f"DiagNutsSettings defaults, num_tune {args.num_tune} (untimed), seed {args.seed}",
                       "chains_per_gpu": C_, "dim": D, "num_tune": args.num_tune, "parallelism": f"chains sharded x{world}, no collective",
                       "rng": "ChaCha8 stream + ziggurat normals (reference semantics)"},
            "draws_per_sec_per_chain": args.steps / elapsed_max,
            "leapfrogs_per_draw": steps_total / (args.steps * C_ * world),
            "adaptation": {"value": tune_steps_total * D / t_tune_max, "unit": "leapfrog-steps*dims/s",
                           "draws": args.num_tune, "seconds": t_tune_max},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "nuts_draw_kernel", "kernel_ms_per_launch": c["kernel_ms"] / max(1, c["kernel_launches"]),
                         "launches": c["kernel_launches"], "algorithmic_bytes_per_launch": algo_bytes / max(1, c["kernel_launches"]),
                         "frac_of_measured_copy": achieved / HBM_COPY_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            try:
                out["cpu_baseline"] = cpu_baseline(args, cores)
            except Exception as e:   # the bench line must still be printed
                out["cpu_baseline"] = {"value": None, "unit": "leapfrog-steps*dims/s", "cores": cores, "kind": "port",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out))
    batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
