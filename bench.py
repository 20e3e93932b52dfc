#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on BASELINE.json's headline configuration (K2), one JSON line on rank 0.

  metric  : leapfrog-steps x dims / second (M1 = sum_chains sum_draws n_steps * D / wall seconds), post-warm-up
            steady state; draws/sec/chain (M2) is reported beside it.
  workload: K2 — iid N(3,1), dim 1024, 4096 chains PER GPU, maxdepth 10, per-chain diagonal mass matrix,
            DiagNutsSettings defaults with num_tune 400; x0 ~ U(-1,1) from each chain's generator, seed 20260928.
  step    : one NUTS draw of every chain (one pass of the hot path over the batch of 4096 chains), WITH the draw and
            its statistics recorded: every timed launch writes `Chain::draw`'s position ([steps][chains][dim] f64) and
            the per-draw statistics ([steps][chains] nm_draw_stats) into device buffers (reference src/chain.rs:151-188
            returns the position of every draw).
            Setup (untimed, reported as `adaptation`): engine creation, set_position, the 400 tuning draws.
            Then W untimed post-warm-up steps, then R repeats of EXACTLY K timed steps, each between
            barrier + device sync; `value` is the MEDIAN repeat (all repeats are listed in `repeats`).
  N GPUs  : one process per GPU (torch.distributed, RCCL backend); `python bench.py --gpus N` starts the N ranks itself
            (torch.distributed.run) when it is not already running under a launcher.  Chains shard with no data-path
            collective (global chain id = rank * 4096 + local id; results are invariant to the partition) => "weak"
            scaling; value = all ranks' steps*dims / max-over-ranks time.
  roofline: (schema_version 6) `bound`, `achieved`, `peak`, `unit`, `frac` describe ONE bound — the binding one of bound_model: for K2 the issue
            rate of necessary 64-lane f64 instructions (`valu_issue`), frac = achieved / peak = t_min / t_kernel.  What the kernel moved through
            HBM is `hbm_moved` (below: how it is measured).  Rounds 4 - 5 printed HBM GB/s in achieved / peak beside the issue bound's frac.
            `calibration` (beside every config): this box's ns per dependent f64 fma of a lone wavefront and its copy rate, and the ratio of the
            config's value to the driver's record of the previous round.
            HBM.  `hbm_moved.achieved_GBps` = bytes the dominant kernel (nuts_draw_kernel) actually moved through the memory-side
            counters (rocprofv3 FETCH_SIZE / WRITE_SIZE in separate --pmc passes of this same workload, taken live by
            this script on rank 0 at N = 1, corrected with the factors calibrated on known-size streams,
            profiles/*hbm_calibration.json) / its launch time (HIP events on the engine's stream, un-profiled run);
            `frac` = `frac_moved` = achieved / 8 TB/s <= 1.  The 64 B/(step x dim) algorithmic model of SURVEY §8(d) is
            reported beside it (`frac_sec8d_model`, `algorithmic`): it is NOT a bound for this design, whose live state is
            register / LDS resident (the model's rate exceeds the HBM peak).
  low_rank_adaptation: a whole `LowRankNutsSettings` warm-up (dim 128 x 1024 chains) with its estimator rounds on the device: wall time,
      the kernels' shares, the host's share.
  other_configs: BASELINE.json's other GPU configurations — K3 (funnel dim 101 x 8192 chains), K4 (8 schools dim 10: the whole
            65536-chain job on one GPU, and one GPU's shard of 8192 chains) and K5 (full-Sigma normal dim 256 x 4096 chains through
            the shared rank-256 transformation, the matrix-core kernel) — each measured by this script in the same run (one untimed
            warm-up, one timed launch of --other-steps recorded draws, HIP-event kernel time) with its own roofline from live
            rocprofv3 --pmc passes: HBM (FETCH_SIZE / WRITE_SIZE) for K3 / K4, f64 MFMA (SQ_VALU_MFMA_BUSY_CYCLES) for K5.
            The headline `value` stays K2.  --other-configs none skips them.
The state is resident in HBM before the timed region (positions are uploaded in set_position; draws stay on the
device), so `value` contains no PCIe traffic.
"""
import argparse
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E datasheet peak (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0          # measured float4 copy on MI355X per the same guide (this box's own probe is reported too)
ALGO_BYTES_PER_STEP_DIM = 64   # read z,v,g_z,sigma,mu + write z',v',g_z' in f64, per (step x dim)
KERNEL = "nuts_draw_kernel"
SQ_COUNTERS = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
               "SQ_INSTS_VALU", "SQ_INSTS_SALU"]


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--repeats", type=int, default=5, help="timed repeats of K steps each; value = the median")
    p.add_argument("--chains", type=int, default=4096, help="chains per GPU")
    p.add_argument("--dim", type=int, default=1024)
    p.add_argument("--num-tune", type=int, default=400)
    p.add_argument("--seed", type=int, default=20260928)
    p.add_argument("--dims-per-lane", type=int, default=0)
    p.add_argument("--no-record", action="store_true", help="do not record draws / statistics in the timed launches (comparison only)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--dist-always", action="store_true",
                   help="initialise torch.distributed even for one rank (the barrier and the timing reductions then run through the backend's "
                        "single-rank paths: lets a one-GPU box execute the RCCL code an 8-GPU node will)")
    p.add_argument("--cpu-chains", type=int, default=0, help="chains of the bounded CPU sample (0 = 8 per usable host core)")
    p.add_argument("--pmc", default="live", choices=["live", "profile", "off"],
                   help="HBM traffic of the timed launch: live rocprofv3 --pmc passes | latest committed profile, scaled | none")
    p.add_argument("--pmc-timeout", type=float, default=150.0)
    p.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # this process runs under rocprofv3 --pmc
    p.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one rank per GPU) | gloo (test rigs: ranks may share a GPU)")
    p.add_argument("--master-port", type=int, default=29511)
    p.add_argument("--other-configs", default="k3,k4_65536,k4_8192,k5", help="comma list of OTHER_CONFIGS keys reported in `other_configs` (rank 0, N = 1); none = skip")
    p.add_argument("--other-steps", type=int, default=100, help="timed draws of each other config")
    p.add_argument("--other-budget", type=float, default=150.0, help="seconds the other configs (runs + counter passes) may take in all")
    p.add_argument("--config", default="k2", help="k2 (the headline line; default) | k4: BASELINE's 8-schools configuration, 65536 chains sharded over the "
                                                  "ranks (strong scaling), with --pooled the opt-in pooled adaptation: ONE RCCL all_gather per window")
    p.add_argument("--pooled", action="store_true", help="--config k4: warm up with the cross-chain pooled Welford reduction (nuts_rs_amd/pooled.py)")
    return p.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# HBM traffic from the memory-side counters
# ------------------------------------------------------------------------------------------------------------------
def calibration():
    """FETCH_SIZE / WRITE_SIZE -> bytes factors measured on known-size 16 B/lane streams (tools/hbm_probe.py under
    rocprofv3 --pmc; MI355X_MICROARCH.md §HBM: FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950).  Falls
    back to the guide's documented x2 / x1 when no calibration file is committed."""
    best = {"fetch_factor": 2.0, "write_factor": 1.0, "source": "MI355X_MICROARCH.md §HBM (uncalibrated default)"}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_calibration.json"))):
        try:
            d = json.load(open(f))
            best = {"fetch_factor": float(d["fetch_factor"]), "write_factor": float(d["write_factor"]),
                    "source": os.path.relpath(f, ROOT), "copy_GBps": d.get("copy_GBps"), "triad_GBps": d.get("triad_GBps")}
        except Exception:
            continue
    return best


def read_counters(db, kernel=KERNEL):
    """{counter: value of the LAST dispatch of `kernel`} from a rocprofv3 rocpd database."""
    c = sqlite3.connect(db)
    res = {}
    for name, cname, val in c.execute("select kernel_name, counter_name, value from counters_collection order by start"):
        if kernel in name:
            res[cname] = val      # later dispatches overwrite: the last one is the timed launch
    return res


def pmc_live(args, config="k2", passes=None, kernel=KERNEL, steps=None, deadline=None):
    """Re-run a workload (tune, warm-up, ONE recorded K-step launch) under rocprofv3 --pmc, one pass per counter set
    (FETCH_SIZE and WRITE_SIZE do not fit one pass; tracing options are never combined with --pmc).  `config`: "k2" (the
    headline workload, this script's own arguments) or a key of OTHER_CONFIGS; counters are those of the LAST dispatch whose
    name contains `kernel` (the timed launch)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out = {}
    deadline = deadline or (time.time() + args.pmc_timeout)
    tmp = tempfile.mkdtemp(prefix="nm_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", NUTS_AMD_SELFTEST="0")     # (the parent process has checked the instantiations it runs: no self-test kernels under the counters)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--config", config, "--gpus", "1", "--steps", str(steps or args.steps),
             "--warmup", str(args.warmup), "--chains", str(args.chains), "--dim", str(args.dim),
             "--num-tune", str(args.num_tune), "--seed", str(args.seed), "--dims-per-lane", str(args.dims_per_lane)]
    if args.no_record:
        child.append("--no-record")
    if passes is None:
        passes = (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("sq", SQ_COUNTERS))
    for tag, counters in passes:
        left = deadline - time.time()
        if left < 10:
            out.setdefault("skipped", []).append(tag)
            continue
        cmd = [exe, "--pmc"] + counters + ["-d", tmp, "-o", tag, "--"] + child
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=left)
        except subprocess.TimeoutExpired:
            out.setdefault("skipped", []).append(tag + " (timeout)")
            continue
        dbs = glob.glob(os.path.join(tmp, "**", tag + "_results.db"), recursive=True)
        if r.returncode != 0 or not dbs:
            out.setdefault("skipped", []).append(f"{tag} (rc {r.returncode})")
            continue
        try:
            out.update(read_counters(dbs[0], kernel))
            for line in r.stdout.decode(errors="replace").splitlines():
                if line.startswith("{") and '"pmc_child"' in line:
                    out.setdefault("child", {})[tag] = json.loads(line)
        except Exception as e:  # noqa: BLE001
            out.setdefault("skipped", []).append(f"{tag} ({e})")
        for f in dbs:
            os.remove(f)
    shutil.rmtree(tmp, ignore_errors=True)
    need = [c for _, cs in passes for c in cs if c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES")]
    if any(c not in out for c in need):
        return None, f"counter passes incomplete: {out.get('skipped')}"
    return out, None


# ------------------------------------------------------------------------------------------------------------------
# A roofline that is a BOUND: t_min = max(necessary HBM bytes / 8 TB/s, necessary VALU cycles / (SIMDs x clock))
# ------------------------------------------------------------------------------------------------------------------
# f64 vector operations per (leapfrog step x element) that ANY evaluation of this tree needs (iid normal, diagonal mass matrix):
#   leapfrog 11 (v half step fma, z' fma, z' sigma mul, + mu fma, x - mu0, two muls of -diff^2/2, the logp sum's add, g = gx sigma mul,
#                v second half fma, kinetic fma);
#   U-turn tests 12 per closed sub-tree level (3 pairs x (difference: add + sub, two fmas)) x (1 level per leaf on average over a
#                balanced doubling + the top-level test of each of the ~ depth doublings: 12 x (1 + depth / leaves));
#   candidate bookkeeping / accounting: none per element.
NEC_VALU_LEAPFROG = 11.0
NEC_VALU_PER_TEST_LEVEL = 12.0
SHADER_HZ, N_SIMD, F64_VALU_CYCLES_PER_WAVE_INSTR = 2.4e9, 1024, 4.0


def bound_model(st, D, kern_s, traffic):
    """st: the statistics rows of a timed launch ([steps][chains]).  Necessary bytes from the per-depth table of
    tools/necessary_traffic.py (Belady MIN over the draw's vector access trace at the kernel's on-chip capacity per chain)."""
    tbl = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_necessary_traffic_k2.json"))):
        try:
            tbl = json.load(open(f)); src = os.path.relpath(f, ROOT)
        except Exception:
            continue
    if tbl is None or tbl.get("dim") != D:
        return {"note": "profiles/r*_necessary_traffic_k2.json missing or for another dim: python tools/necessary_traffic.py --dim D --out ..."}
    depth = st["depth"].ravel().astype(np.int64)
    n_steps = st["n_steps"].ravel().astype(np.float64)
    io = float(tbl["io_bytes_per_draw"])
    per = {int(k): float(v["necessary_bytes_per_draw"]) for k, v in tbl["per_depth"].items()}
    nec_bytes = float(sum(per.get(int(d), io) * cnt for d, cnt in zip(*np.unique(depth, return_counts=True))))
    levels = n_steps.sum() * 1.0 + depth.sum()            # sub-tree levels closed (one per leaf on average) + one top-level test per doubling
    nec_ops_elem = NEC_VALU_LEAPFROG * n_steps.sum() + NEC_VALU_PER_TEST_LEVEL * levels
    nec_cycles = nec_ops_elem * (D / 64.0) * F64_VALU_CYCLES_PER_WAVE_INSTR
    t_hbm = nec_bytes / (HBM_PEAK_GBS * 1e9)
    t_valu = nec_cycles / (N_SIMD * SHADER_HZ)
    t_min = max(t_hbm, t_valu)
    return {"frac": t_min / kern_s, "t_min_ms": t_min * 1e3, "t_kernel_ms": kern_s * 1e3, "binding": "valu" if t_valu >= t_hbm else "hbm",
            "terms_ms": {"hbm": t_hbm * 1e3, "valu": t_valu * 1e3}, "necessary_valu_instr": nec_ops_elem * (D / 64.0),
            "hbm": {"necessary_bytes": nec_bytes, "t_ms": t_hbm * 1e3, "frac": t_hbm / kern_s,
                    "necessary_bytes_per_chain_draw": nec_bytes / max(1, depth.size),
                    "moved_over_necessary": (traffic / nec_bytes) if traffic else None,
                    "enumeration": "per draw: read z, g_z, sigma, mu + write z, g_z + position row (7 vectors) + 192 B statistics; tree end points only "
                                   "where Belady's MIN over the draw's vector trace cannot keep them in the chain's on-chip share "
                                   f"({tbl['capacity_vectors']['for_tree_end_points']} vectors beside the leapfrog's working set: {src})"},
            "valu": {"necessary_wave_cycles": nec_cycles, "t_ms": t_valu * 1e3, "frac": t_valu / kern_s,
                     "f64_ops_per_element_step": nec_ops_elem / max(1.0, n_steps.sum()),
                     "enumeration": "11 f64 operations per (leapfrog x element) + 12 per (closed sub-tree level x element), 4 cycles per 64-lane f64 instruction, "
                                    "1024 SIMDs at 2.4 GHz"},
            "definition": "t_min = max(necessary HBM bytes / 8 TB/s, necessary VALU cycles / (1024 SIMDs x 2.4 GHz)); frac = t_min / t_kernel <= 1 by "
                          "construction; moving more bytes cannot raise it"}


def generic_bound(n_steps_cd, depth_cd, D, kern_s, traffic, elems_per_lane=None, latency=False):
    """A lower bound on the launch time of ANY engine that integrates these trees: the largest of
      t_hbm     necessary HBM bytes / 8 TB/s — per draw: read z, g_z, sigma, mu, write z, g_z and the position row (7 vectors) + the
                192 B statistics row (tree end points are not counted here: a LOWER bound stays one when terms are left out);
      t_valu    necessary f64 vector instructions / (1024 SIMDs x one 64-lane f64 instruction per 4 cycles x 2.4 GHz): 11 per
                (leapfrog x element) + 12 per (closed sub-tree level x element), elements packed 64 per instruction;
      t_chain   (latency=True: launches that last as long as their slowest chain) the leapfrogs of the deepest chain x the cycles one
                wavefront needs to ISSUE the necessary instructions of one leapfrog of one chain (a lone wavefront issues one instruction
                per 4 cycles, dependent or not: tools/probes/ubench_issue.hip).
    n_steps_cd / depth_cd: [draws][chains] of the timed launch."""
    n_steps = n_steps_cd.astype(np.float64)
    depth = depth_cd.astype(np.float64)
    draws = float(n_steps.size)
    nec_bytes = draws * (7.0 * D * 8.0 + 192.0)
    levels = n_steps.sum() + depth.sum()
    ops_elem = NEC_VALU_LEAPFROG * n_steps.sum() + NEC_VALU_PER_TEST_LEVEL * levels
    t_hbm = nec_bytes / (HBM_PEAK_GBS * 1e9)
    t_valu = ops_elem * (D / 64.0) * F64_VALU_CYCLES_PER_WAVE_INSTR / (N_SIMD * SHADER_HZ)
    terms = {"hbm": t_hbm, "valu": t_valu}
    if latency:
        per_chain = n_steps.sum(axis=0)
        epl = elems_per_lane or max(1.0, float(-(-D // 64)))
        ops_leaf = (NEC_VALU_LEAPFROG + NEC_VALU_PER_TEST_LEVEL * (levels / max(1.0, n_steps.sum()))) * epl
        terms["slowest_chain"] = float(per_chain.max()) * ops_leaf * F64_VALU_CYCLES_PER_WAVE_INSTR / SHADER_HZ
    binding = max(terms, key=terms.get)
    t_min = terms[binding]
    return {"frac": t_min / kern_s, "binding": binding, "t_min_ms": t_min * 1e3, "t_kernel_ms": kern_s * 1e3,
            "terms_ms": {k: v * 1e3 for k, v in terms.items()}, "necessary_bytes": nec_bytes, "necessary_valu_instr": ops_elem * (D / 64.0),
            "waste_ratio": (traffic / nec_bytes) if traffic else None}


ISSUE_PEAK_INSTR_S = N_SIMD * SHADER_HZ / F64_VALU_CYCLES_PER_WAVE_INSTR      # 64-lane f64 instructions per second the chip can issue
SCHEMA_VERSION = 6
# the driver's records of the previous round (BENCH_r05.json), for `calibration.ratio_to_previous_round`
PREVIOUS_ROUND = {"round": 5, "k2": 2.047e11, "k3": 1.81e8, "k4_65536": 4.79e9, "k4_8192": 2.08e9, "k5": 4.30e7,
                  "unit": {"k2": "leapfrog-steps*dims/s", "k3": "leapfrogs/s", "k4_65536": "leapfrogs/s", "k4_8192": "leapfrogs/s", "k5": "leapfrogs/s"}}


def one_bound(binding, terms_ms, kern_s, nec_valu_instr=None, nec_bytes=None):
    """The roofline object's four scalars for ONE bound: `bound`, `achieved`, `peak`, `unit` describe the SAME thing `frac` does
    (frac = achieved / peak = t_min / t_kernel <= 1).  valu_issue: necessary 64-lane f64 instructions per second against the chip's issue rate;
    hbm: necessary bytes per second against 8 TB/s; slowest_chain: the necessary issue time of the deepest chain against the launch time."""
    t_min = terms_ms[binding] * 1e-3
    if binding == "valu" and nec_valu_instr:
        return {"bound": "valu_issue", "achieved": nec_valu_instr / kern_s, "peak": ISSUE_PEAK_INSTR_S, "unit": "necessary 64-lane f64 instructions/s",
                "frac": (nec_valu_instr / kern_s) / ISSUE_PEAK_INSTR_S}
    if binding == "hbm" and nec_bytes:
        return {"bound": "hbm", "achieved": nec_bytes / kern_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s (necessary bytes)", "frac": nec_bytes / kern_s / 1e9 / HBM_PEAK_GBS}
    return {"bound": binding, "achieved": t_min * 1e3, "peak": kern_s * 1e3, "unit": "ms (necessary / measured)", "frac": t_min / kern_s}


_calibration_cache = {}


def box_calibration(key=None, value=None):
    """Fixed-work figures of THIS box beside every config: a lone wavefront's nanoseconds per dependent v_fma_f64 with one
    wavefront per SIMD (nm_probe_issue: what every BASELINE kernel here is bound by) and the copy rate of the engine's access shape
    (nm_probe_bandwidth) — so that a slower box is not read as a regression — and the ratio of this config's value to the driver's record of the
    previous round."""
    import ctypes as C
    from nuts_rs_amd import _lib
    if "box" not in _calibration_cache:
        L = _lib.load()
        out = {}
        try:
            ns = C.c_double()
            _lib.check(L.nm_probe_issue(0, 1 << 20, C.byref(ns)))
            out["issue_ns_per_dependent_f64_fma"] = ns.value
            out["issue_cycles_at_2p4GHz"] = ns.value * 2.4
            ms, rd, wr = C.c_double(), C.c_uint64(), C.c_uint64()
            _lib.check(L.nm_probe_bandwidth(0, 1 << 30, 5, C.byref(ms), C.byref(rd), C.byref(wr)))
            out["copy_GBps"] = (rd.value + wr.value) / (ms.value * 1e-3) / 1e9
        except Exception as e:  # noqa: BLE001
            out["error"] = f"{type(e).__name__}: {e}"
        _calibration_cache["box"] = out
    cal = dict(_calibration_cache["box"])
    if key in PREVIOUS_ROUND and value:
        cal["previous_round"] = {"round": PREVIOUS_ROUND["round"], "value": PREVIOUS_ROUND[key], "unit": PREVIOUS_ROUND["unit"][key]}
        cal["ratio_to_previous_round"] = value / PREVIOUS_ROUND[key]
    return cal


def pmc_profile(steps_dims):
    """Fallback: the latest committed profile's bytes per (leapfrog-step x dim), scaled to this run's steps x dims."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_k2.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("hbm_bytes_per_step_dim"):
            best = (f, d)
    if not best:
        return None, None
    f, d = best
    return d["hbm_bytes_per_step_dim"] * steps_dims, os.path.relpath(f, ROOT)


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline
# ------------------------------------------------------------------------------------------------------------------
def usable_cores():
    """Threads worth of CPU this process may use: the affinity mask, capped by the cgroup CPU quota (a container that shows 256
    logical CPUs may be allowed 16 CPUs' worth of time; more threads than that only get throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = f"{n} logical CPUs in the affinity mask"
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]           # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())    # cgroup v1
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None and quota < n:
        note += f", cgroup CPU quota {quota:g} CPUs: {max(1, int(round(quota)))} threads used"
        n = max(1, int(round(quota)))
    return n, note


def cpu_baseline(args, cores):
    """The CPU restatement of nuts-rs (oracle/, reference arithmetic: libm + SIMD-order sums) on this box's host
    cores, WALL CLOCK: one chain per task over `cores` threads (the reference's Rayon structure, src/sampler.rs:1116),
    same density / settings / seeds, a bounded sample of the same workload.  Labelled "port": it is NOT nuts-rs itself
    (no Rust toolchain in this image; `cargo` is probed and reported)."""
    from oracle import oracle as O
    n = args.cpu_chains or 8 * cores        # ~20 CPU-seconds at 16 threads: 400 warm-up + 400 timed draws per chain
    draws = 400
    s = O.default_settings(seed=args.seed, num_tune=args.num_tune, num_chains=n)
    x0 = O.init_positions_uniform(args.seed, 0, n, args.dim)
    r = O.run_wall(s, O.LOGP_IID_NORMAL, args.dim, [3.0], O.ref_cfg(), n, x0, args.num_tune, draws, n_threads=cores)
    rate = r["steps"] * args.dim / r["wall_seconds"] if r["wall_seconds"] > 0 else 0.0
    out = {
        "value": rate, "unit": "leapfrog-steps*dims/s", "cores": cores, "kind": "port", "timing": "wall clock",
        "sample": f"CPU restatement of nuts-rs (oracle/, not nuts-rs): {n} chains x dim {args.dim}, same settings/seed, "
                  f"{args.num_tune} warm-up draws per chain (phase 1, {r['warm_wall_seconds']:.2f} s wall) then {draws} "
                  f"timed draws per chain, one chain per task on {cores} threads; {r['steps']} steps in "
                  f"{r['wall_seconds']:.3f} s wall",
        "warmup_value": r["warm_steps"] * args.dim / r["warm_wall_seconds"] if r["warm_wall_seconds"] > 0 else 0.0,
        "reference_toolchain": {"cargo": shutil.which("cargo"), "rustc": shutil.which("rustc"),
                                "note": "BASELINE.md §4.1: the reference itself is timed only where cargo exists"},
    }
    # criterion replica of the reference's own bench (benches/sample.rs:76-98, :190-198): 1 chain, 1000 draws all inside
    # the warm-up (num_tune 1000), maxdepth 3, N(3,1), init 3.5 — wall ms on one core, dims 10 and 1000
    try:
        rep = {}
        for d in (10, 1000):
            s1 = O.default_settings(seed=42, num_tune=1000, num_draws=1000, maxdepth=3, num_chains=1)
            best = None
            for _ in range(3):
                rr = O.run_wall(s1, O.LOGP_IID_NORMAL, d, [3.0], O.ref_cfg(), 1, np.full((1, d), 3.5), 1000, 0, n_threads=1)
                best = rr["warm_wall_seconds"] if best is None else min(best, rr["warm_wall_seconds"])
            rep[f"sample_1000_{d}"] = {"ms": best * 1e3}
        out["criterion_replica"] = dict(rep, note="benches/sample.rs:190-198 on the CPU restatement, 1 thread, best of 3; "
                                                  "the chain key is this repo's Sampler seeding of seed 42, not StdRng's")
    except Exception as e:  # noqa: BLE001
        out["criterion_replica"] = {"failed": str(e)}
    return out


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json's other GPU configurations: measured by this script, reported inside the one JSON line
# ------------------------------------------------------------------------------------------------------------------
F64_MFMA_PEAK_TFLOPS = 78.6    # MI355X dense f64 matrix peak (MI355X_MICROARCH.md): 16 FMA / cycle / SIMD x 1024 SIMDs x 2.4 GHz
SHADER_CLOCK_HZ = 2.4e9
N_SIMDS = 1024


def _k5_target(dim, seed=55, rank=8, scale=100.0):
    rng = np.random.default_rng(seed)
    u = np.linalg.qr(rng.normal(size=(dim, rank)))[0]
    sigma = np.eye(dim) + u @ np.diag(rng.uniform(5.0, scale, rank)) @ u.T
    sc = np.exp(rng.normal(0, 0.5, dim))
    sigma = np.diag(sc) @ sigma @ np.diag(sc)
    prec = np.linalg.inv(sigma)
    return (prec + prec.T) / 2, sigma


OTHER_CONFIGS = {
    "k3": dict(name="K3: Neal's funnel dim 101 x 8192 chains, DiagNutsSettings defaults, num_tune 400", chains=8192, tune=400, bound="hbm",
               kernel="nuts_draw_kernel"),
    "k4_65536": dict(name="K4: 8 schools non-centered dim 10, all 65536 chains of the job on ONE GPU, DiagNutsSettings defaults, num_tune 400",
                     chains=65536, tune=400, bound="hbm", kernel="nuts_lane"),
    "k4_8192": dict(name="K4: 8 schools non-centered dim 10, one GPU's shard (8192 chains) of the 65536-chain job, num_tune 400",
                    chains=8192, tune=400, bound="hbm", kernel="nuts_"),
    "k5": dict(name="K5: N(0, Sigma) full Sigma dim 256 x 4096 chains through the exact dense preconditioner = LowRankMassMatrix of rank 256 shared "
                    "by all chains (frozen), step size adapted over 100 draws", chains=4096, tune=100, bound="mfma", kernel="nuts_lockstep"),
}


def run_other_config(key, seed, steps, warmup, record=True):
    """One of BASELINE's other configurations on the current GPU: untimed set-up + warm-up, then ONE timed launch of `steps` recorded draws."""
    import torch
    import nuts_rs_amd as N
    cfg = OTHER_CONFIGS[key]
    C_ = cfg["chains"]
    extra = {}
    if key == "k3":
        logp = N.LogpSpec.funnel(101)
        s = N.DiagNutsSettings(num_chains=C_, seed=seed, num_tune=cfg["tune"], num_draws=steps + warmup)
        b = N.ChainBatch(s, logp, C_)
    elif key.startswith("k4"):
        logp = N.LogpSpec.eight_schools()
        s = N.DiagNutsSettings(num_chains=C_, seed=seed, num_tune=cfg["tune"], num_draws=steps + warmup)
        b = N.ChainBatch(s, logp, C_)
    else:
        D = 256
        prec, sigma = _k5_target(D)
        w, u = np.linalg.eigh(sigma)
        logp = N.LogpSpec.mvn_precision(prec)
        s = N.LowRankNutsSettings(num_chains=C_, seed=seed, num_tune=cfg["tune"], num_draws=steps + warmup, freeze_transform=True)
        b = N.ChainBatch(s, logp, C_, lowrank_max_rank=D)
        extra["transform"] = (np.ones(D), np.zeros(D), w, np.ascontiguousarray(u.T), np.zeros(D))
    D = logp.dim
    assert (b.set_position(b.init_positions_uniform()) == 0).all()
    if "transform" in extra:
        b.set_transform(*extra["transform"])
    t0 = time.perf_counter()
    b.draw_device(cfg["tune"])
    t_tune = time.perf_counter() - t0
    c_tune = b.counters()
    if warmup:
        b.draw_device(warmup)
    d_pos = torch.empty((steps, C_, D), dtype=torch.float64, device="cuda") if record else None
    d_st = torch.zeros((steps, C_, N.STATS_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    b.reset_counters()
    t0 = time.perf_counter()
    b.draw_device(steps, d_pos.data_ptr() if record else 0, d_st.data_ptr(), sync=False)
    b.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = b.counters()
    st = np.frombuffer(d_st.cpu().numpy().tobytes(), dtype=N.STATS_DTYPE).reshape(steps, C_)
    n_steps = int(st["n_steps"].sum())
    ok = n_steps == c["total_leapfrogs"] and bool((st["draw"][-1] == cfg["tune"] + warmup + steps - 1).all()) and \
        (not record or bool(torch.isfinite(d_pos[-1]).all().item()))
    res = {"_n_steps": st["n_steps"].astype(np.int64), "_depth": st["depth"].astype(np.int64),
           "workload": cfg["name"], "chains": C_, "dim": D, "steps": steps, "value": n_steps * D / dt, "unit": "leapfrog-steps*dims/s",
           "leapfrogs_per_s": n_steps / dt, "ms_per_step": dt / steps * 1e3, "draws_per_sec_per_chain": steps / dt,
           "leapfrogs_per_draw": n_steps / (steps * C_), "kernel_ms_per_launch": c["kernel_ms"], "total_leapfrogs": n_steps,
           "divergence_rate": float(st["diverging"].mean()), "mean_depth": float(st["depth"].mean()),
           "warmup": {"draws": cfg["tune"], "seconds": t_tune, "kernel_ms": c_tune["kernel_ms"],
                      "leapfrogs_per_s": c_tune["total_leapfrogs"] / t_tune if t_tune > 0 else None},
           "draws_recorded": record, "recorded_buffers_verified": ok,
           "kernels": {"lane_launches": b.lane_launches(), "group_launches": b.group_launches(), "matrix_core_launches": b.tile_launches()}}
    b.close()
    return res


def other_config_roofline(args, key, res, deadline):
    """The roofline object of one other config, from live counter passes of the same workload (separate --pmc runs)."""
    cfg = OTHER_CONFIGS[key]
    kern_s = res["kernel_ms_per_launch"] * 1e-3
    n_steps, D = res["total_leapfrogs"], res["dim"]
    cal = calibration()
    if cfg["bound"] == "hbm":
        algo = n_steps * D * ALGO_BYTES_PER_STEP_DIM
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "kernel": cfg["kernel"],
                "kernel_ms_per_launch": res["kernel_ms_per_launch"],
                "algorithmic": {"bytes_per_step_dim": ALGO_BYTES_PER_STEP_DIM, "bytes_per_launch": algo, "rate_GBps": algo / kern_s / 1e9,
                                "frac_sec8d_model": algo / kern_s / 1e9 / HBM_PEAK_GBS}}
        if args.pmc == "live":
            got, why = pmc_live(args, key, (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"])), cfg["kernel"], res["steps"], deadline)
            if got:
                child_steps = (got.get("child", {}).get("fetch") or {}).get("steps") or n_steps
                fetch_b = got["FETCH_SIZE"] * 1024.0 * cal["fetch_factor"]
                write_b = got["WRITE_SIZE"] * 1024.0 * cal["write_factor"]
                traffic = (fetch_b + write_b) / child_steps * n_steps
                roof.update(traffic=traffic, achieved=traffic / kern_s / 1e9, frac=traffic / kern_s / 1e9 / HBM_PEAK_GBS,
                            traffic_source="live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload",
                            hbm_bytes_per_leapfrog=(fetch_b + write_b) / child_steps, moved_over_algorithmic=traffic / algo)
            else:
                roof["pmc_failed"] = why
        # `frac` is a BOUND (t_min / t_kernel <= 1, cannot be raised by moving more bytes); the utilisation of moved bytes is `frac_moved`
        bm = generic_bound(res["_n_steps"], res["_depth"], D, kern_s, roof["traffic"], latency=(key == "k3"))
        roof["hbm_moved"] = {"achieved_GBps": roof["achieved"], "peak_GBps": HBM_PEAK_GBS, "frac_moved": roof["frac"]}
        roof["frac_moved"] = roof["frac"]
        roof.update(one_bound(bm["binding"], bm["terms_ms"], kern_s, bm.get("necessary_valu_instr"), bm.get("necessary_bytes")))
        roof.update(binding=bm["binding"], waste_ratio=bm["waste_ratio"], t_min_ms=bm["t_min_ms"], bound_terms_ms=bm["terms_ms"],
                    frac_sec8d_model=roof["algorithmic"]["frac_sec8d_model"],
                    frac_definition="frac = t_min / t_kernel, t_min = max(necessary HBM bytes / 8 TB/s, necessary f64 VALU instructions / chip issue rate"
                                    + (", leapfrogs of the deepest chain x issue cycles of one leapfrog" if key == "k3" else "") + "); waste_ratio = HBM bytes moved / necessary")
        return roof
    flop = 2.0 * (4 * D * D + D * D)        # per chain-leapfrog: U'v and U s for x and for g_z (rank = dim), P x for the density
    useful = n_steps * flop / kern_s / 1e12
    roof = {"bound": "mfma", "achieved": useful, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": useful / F64_MFMA_PEAK_TFLOPS, "traffic": None,
            "kernel": cfg["kernel"], "kernel_ms_per_launch": res["kernel_ms_per_launch"],
            "flop_per_chain_leapfrog": flop, "note": "achieved = USEFUL f64 flop of the five dense products per chain-leapfrog / launch time; "
            "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch time x 2.4 GHz) also counts the products of idle columns"}
    if args.pmc == "live":
        got, why = pmc_live(args, key, (("mfma", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64", "GRBM_GUI_ACTIVE"]),), cfg["kernel"], res["steps"], deadline)
        if got:
            child = (got.get("child", {}).get("mfma") or {})
            ck = (child.get("kernel_ms") or res["kernel_ms_per_launch"]) * 1e-3
            roof["mfma_busy"] = got["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMDS * ck * SHADER_CLOCK_HZ)
            if got.get("GRBM_GUI_ACTIVE"):
                roof["mfma_busy_of_gui_active"] = got["SQ_VALU_MFMA_BUSY_CYCLES"] / (got["GRBM_GUI_ACTIVE"] / 8.0 * N_SIMDS)
            if got.get("SQ_INSTS_VALU_MFMA_MOPS_F64"):
                roof["mfma_issued_TFLOPs"] = got["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0 / ck / 1e12
        else:
            roof["pmc_failed"] = why
    roof["binding"] = "mfma"
    roof["waste_ratio"] = (roof["mfma_issued_TFLOPs"] / useful) if roof.get("mfma_issued_TFLOPs") else None      # matrix-core flops issued / useful
    roof["frac_definition"] = "frac = useful f64 flop of the dense products / launch time / 78.6 TFLOP/s (<= 1); waste_ratio = issued / useful matrix-core flops"
    return roof


def other_configs(args):
    keys = [k for k in args.other_configs.split(",") if k and k != "none"]
    out, deadline = [], time.time() + args.other_budget
    for k in keys:
        if k not in OTHER_CONFIGS:
            out.append({"workload": k, "error": "unknown config"})
            continue
        if time.time() > deadline - 15:
            out.append({"workload": OTHER_CONFIGS[k]["name"], "skipped": "time budget of --other-budget spent"})
            continue
        try:
            res = run_other_config(k, args.seed, args.other_steps, args.warmup, record=not args.no_record)
            res["roofline"] = other_config_roofline(args, k, res, deadline)
            res.pop("_n_steps", None); res.pop("_depth", None)
            res["key"] = k
            res["calibration"] = box_calibration(k, res["value"] if k == "k2" else res["leapfrogs_per_s"])
            out.append(res)
        except Exception as e:  # noqa: BLE001  (the bench line must still be printed)
            out.append({"workload": OTHER_CONFIGS[k]["name"], "key": k, "error": f"{type(e).__name__}: {e}"})
    return out


def low_rank_adaptation(seed):
    """`LowRankNutsSettings` adapting per chain: the whole warm-up of 1024 chains at dim 128 — draw
    kernels + the estimator rounds (`compute_update`, src/transform/adapt/low_rank.rs:73-142, on the device: csrc/lowrank_device.hip) — timed
    as a caller sees it, with the share of it that is host time.  tools/bench_lowrank_adapt.py is the stand-alone form (its profile:
    profiles/r04r_*)."""
    import ctypes as C
    import nuts_rs_amd as N
    dim, chains, tune = 128, 1024, 300
    rng = np.random.default_rng(3)
    u = np.linalg.qr(rng.normal(size=(dim, 4)))[0]
    sigma = np.eye(dim) + u @ np.diag([100.0, 50.0, 20.0, 10.0]) @ u.T
    sc = np.exp(rng.normal(0, 0.5, dim))
    sigma = np.diag(sc) @ sigma @ np.diag(sc)
    prec = np.linalg.inv(sigma)
    s = N.LowRankNutsSettings(num_chains=chains, seed=seed, num_tune=tune, num_draws=100)
    b = N.ChainBatch(s, N.LogpSpec.mvn_precision((prec + prec.T) / 2), chains)
    try:
        b.init_with_retries()
        t = time.time()
        _, st = b.draw_many(tune, positions=False)
        wall = time.time() - t
        c = b.counters()
        tm = (C.c_double * 6)()
        N.load_library().nm_debug_lowrank_timing(b._h, tm)
        on_dev = b.lowrank_device_updates()
        est_s = tm[2] if on_dev else 0.0
        return {"workload": f"LowRankNutsSettings adapting, full-precision normal dim {dim} x {chains} chains, num_tune {tune}",
                "warmup_wall_s": wall, "draw_kernels_s": c["kernel_ms"] * 1e-3, "estimator_rounds": int(tm[4]), "estimator_calls": int(tm[5]),
                "estimator_calls_on_device": on_dev, "estimator_kernel_s": est_s,
                "host_share": (wall - c["kernel_ms"] * 1e-3 - est_s) / wall,
                "updates_per_chain": float((st["transformation_update_id"] >= 0).sum() / chains),
                "n_eig_median": float(np.median(b.lowrank()[0])), "divergences_in_warmup": int(st["diverging"].sum())}
    finally:
        b.close()


# ------------------------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks (one per GPU) and relay rank 0's line."""
    import torch
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if args.dist_backend == "nccl" and ndev < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible (one rank per GPU over RCCL); "
                         f"use --dist-backend gloo to let ranks share a GPU on a test rig")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main_k4(args, N, torch, dist, rank, world, device_index, red_dev, barrier):
    """BASELINE configs[3]: hierarchical 8 schools (non-centered, dim 10), 65536 chains sharded over the ranks, RCCL only for the
    cross-chain adaptation reduction.  Strong scaling: the job's 65536 chains are split, `value` = all ranks' steps x dims / max time.
    --pooled: the warm-up pools the chains' Welford statistics over ALL ranks, one all_gather per window (not reference behaviour:
    every reference chain adapts alone; opt-in)."""
    total = 65536
    C_ = total // world
    logp = N.LogpSpec.eight_schools()
    D = logp.dim
    n_draws = args.steps * max(1, args.repeats) + args.warmup
    if args.pooled:
        from nuts_rs_amd import pooled
        settings = N.LowRankNutsSettings(num_chains=total, seed=args.seed, num_tune=args.num_tune, num_draws=n_draws, freeze_transform=True)
    else:
        settings = N.DiagNutsSettings(num_chains=total, seed=args.seed, num_tune=args.num_tune, num_draws=n_draws)
    batch = N.ChainBatch(settings, logp, C_, chain_id_offset=rank * C_, device=device_index)
    assert (batch.set_position(batch.init_positions_uniform()) == 0).all()
    barrier()
    t0 = time.perf_counter()
    n_updates = 0
    if args.pooled:
        n_updates = len(pooled.pooled_warmup(batch, args.num_tune, dist=dist, collective_device=None if args.dist_backend == "nccl" else "cpu"))
    else:
        batch.draw_device(args.num_tune)
    t_tune_local = time.perf_counter() - t0              # this rank's own warm-up (before the barrier: the first 8-GPU run is diagnosable per rank)
    tune_kernel_ms_local = float(batch.counters()["kernel_ms"])
    barrier()
    t_tune = time.perf_counter() - t0
    if args.warmup:
        batch.draw_device(args.warmup)
    d_pos = torch.empty((args.steps, C_, D), dtype=torch.float64, device=f"cuda:{device_index}")
    d_st = torch.zeros((args.steps, C_, N.STATS_DTYPE.itemsize), dtype=torch.uint8, device=f"cuda:{device_index}")
    torch.cuda.synchronize()
    reps = []
    for _ in range(max(1, args.repeats)):
        batch.reset_counters()
        barrier()
        t0 = time.perf_counter()
        batch.draw_device(args.steps, d_pos.data_ptr(), d_st.data_ptr(), sync=False)
        batch.synchronize()
        el_local = time.perf_counter() - t0
        barrier()
        el = time.perf_counter() - t0
        steps_local = float(batch.counters()["total_leapfrogs"])
        last_local = (el_local, float(batch.counters()["kernel_ms"]), steps_local)
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=red_dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sv = torch.tensor([steps_local], dtype=torch.float64, device=red_dev); dist.all_reduce(sv, op=dist.ReduceOp.SUM)
            reps.append((float(t.item()), float(sv.item())))
        else:
            reps.append((el, steps_local))
    # per-rank seconds: every rank's own warm-up and sampling times, gathered on rank 0
    mine = [float(rank), t_tune_local, tune_kernel_ms_local * 1e-3, last_local[0], last_local[1] * 1e-3, last_local[2]]
    per_rank = [mine]
    if dist is not None:
        g = [torch.zeros(len(mine), dtype=torch.float64, device=red_dev) for _ in range(world)]
        dist.all_gather(g, torch.tensor(mine, dtype=torch.float64, device=red_dev))
        per_rank = [[float(v) for v in t.tolist()] for t in g]
    if rank == 0:
        rates = [sv * D / el for el, sv in reps]
        mid = int(np.argsort(rates)[len(rates) // 2])
        el, sv = reps[mid]
        print(json.dumps({
            "metric": "leapfrog-steps*dims/sec at 65536 chains x dim 10 (8 schools, post-warm-up NUTS draws)", "value": rates[mid],
            "unit": "leapfrog-steps*dims/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"K4: hierarchical 8 schools non-centered dim 10, 65536 chains sharded x{world} ({C_} per GPU), "
                                   f"{'pooled cross-chain adaptation: one RCCL all_gather per window' if args.pooled else 'per-chain DiagNutsSettings adaptation, no collective'}, "
                                   f"num_tune {args.num_tune}, seed {args.seed}",
                       "chains_per_gpu": C_, "dim": D, "pooled": bool(args.pooled), "pooled_updates": n_updates, "backend": args.dist_backend,
                       "kernels": {"lane_launches": batch.lane_launches(), "group_launches": batch.group_launches()}},
            "leapfrogs_per_s": sv / el, "leapfrogs_per_draw": sv / (args.steps * total),
            "repeats": {"n": len(reps), "reported": "median", "values": rates},
            "adaptation": {"draws": args.num_tune, "seconds": t_tune},
            "per_rank": [{"rank": int(r[0]), "warmup_s": r[1], "warmup_kernel_s": r[2], "sampling_s_last_repeat": r[3], "sampling_kernel_s_last_repeat": r[4],
                          "leapfrogs_last_repeat": r[5]} for r in per_rank],
            "expected": {"note": "one MI355X measured alone (profiles/r05*): the whole 65536-chain job on ONE GPU ~5e9 leapfrogs/s (one chain per lane); "
                                 "a shard of 8192 chains ~2e9 leapfrogs/s (1024 wavefronts of 8 chains: one per SIMD, latency-bound) => 8 GPUs ~1.6e10 = ~3x one GPU "
                                 "on this STRONG-scaling config; K2 (weak scaling, no collective) scales with the GPU count"},
            "roofline": None, "cpu_baseline": None}))
    batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    import torch
    import nuts_rs_amd as N

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE is {world}")
    ndev = torch.cuda.device_count()
    if args.dist_backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
    device_index = local_rank % ndev
    torch.cuda.set_device(device_index)
    red_dev = "cuda" if args.dist_backend == "nccl" else "cpu"
    dist = None
    if world > 1 or args.dist_always:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(args.master_port))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist_mod.init_process_group(args.dist_backend)
        dist = dist_mod

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.pmc_child and args.config != "k2":       # a counter pass over one of the other configs
        r = run_other_config(args.config, args.seed, args.steps, args.warmup, record=not args.no_record)
        print(json.dumps({"pmc_child": True, "config": args.config, "steps": r["total_leapfrogs"], "kernel_ms": r["kernel_ms_per_launch"]}))
        return

    if args.config == "k4":
        return main_k4(args, N, torch, dist, rank, world, device_index, red_dev, barrier)

    C_, D = args.chains, args.dim
    settings = N.DiagNutsSettings(num_chains=C_ * world, seed=args.seed, num_tune=args.num_tune,
                                  num_draws=args.steps * max(1, args.repeats) + args.warmup)
    batch = N.ChainBatch(settings, N.LogpSpec.iid_normal(D, 3.0), C_, chain_id_offset=rank * C_,
                         device=device_index, dims_per_lane=args.dims_per_lane)
    x0 = batch.init_positions_uniform()
    status = batch.set_position(x0)
    assert (status == 0).all()
    # the draws and their statistics are recorded in these device buffers by every timed launch
    record = not args.no_record
    d_pos = d_st = None
    if record:
        d_pos = torch.empty((args.steps, C_, D), dtype=torch.float64, device=f"cuda:{device_index}")
        d_st = torch.zeros((args.steps, C_, N.STATS_DTYPE.itemsize), dtype=torch.uint8, device=f"cuda:{device_index}")
    p_pos, p_st = (d_pos.data_ptr(), d_st.data_ptr()) if record else (0, 0)
    torch.cuda.synchronize()       # torch's fill kernels run on torch's stream, the engine on its own
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
    # ---- R repeats of K timed steps
    R = 1 if args.pmc_child else max(1, args.repeats)
    reps = []
    for _ in range(R):
        batch.reset_counters()
        barrier()
        t0 = time.perf_counter()
        batch.draw_device(args.steps, p_pos, p_st, sync=False)
        batch.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        c = batch.counters()
        steps_local = c["total_leapfrogs"]
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sv = torch.tensor([float(steps_local)], dtype=torch.float64, device=red_dev)
            dist.all_reduce(sv, op=dist.ReduceOp.SUM)
            reps.append((float(t.item()), float(sv.item()), steps_local, c["kernel_ms"]))
        else:
            reps.append((elapsed, float(steps_local), steps_local, c["kernel_ms"]))
    if dist is not None:
        sv = torch.tensor([float(c_tune["total_leapfrogs"]), t_tune], dtype=torch.float64, device=red_dev)
        sv_max = sv.clone()
        dist.all_reduce(sv, op=dist.ReduceOp.SUM)
        dist.all_reduce(sv_max, op=dist.ReduceOp.MAX)
        tune_steps_total, t_tune_max = float(sv[0]), float(sv_max[1])
    else:
        tune_steps_total, t_tune_max = float(c_tune["total_leapfrogs"]), t_tune

    if args.pmc_child:
        el, st_tot, st_loc, kms = reps[0]
        print(json.dumps({"pmc_child": True, "steps": st_loc, "kernel_ms": kms, "recorded": record}))
        batch.close()
        return

    recorded_ok = None
    st_all = None
    if record and rank == 0:     # the last repeat's buffers hold real draws: finite positions, the right draw indices
        st_all = np.frombuffer(d_st.cpu().numpy().tobytes(), dtype=N.STATS_DTYPE).reshape(args.steps, C_)
        st_host = st_all[-1]
        first_draw = args.num_tune + args.warmup + (R - 1) * args.steps
        recorded_ok = bool(torch.isfinite(d_pos[-1]).all().item()) and bool((st_host["draw"] == first_draw + args.steps - 1).all()) \
            and bool((st_host["n_steps"] > 0).all())
    if rank == 0:
        rates = [s_tot * D / el for el, s_tot, _, _ in reps]
        order = np.argsort(rates)
        mid = int(order[len(order) // 2])                       # the median repeat IS the reported measurement
        elapsed_max, steps_total, steps_local, kern_ms = reps[mid]
        value = rates[mid]
        kern_s = kern_ms * 1e-3                                 # HIP events on the engine's own stream (nm_engine_get_counters)
        algo_bytes = steps_local * D * ALGO_BYTES_PER_STEP_DIM
        cal = calibration()
        traffic = traffic_src = None
        pmc_detail = {}
        if world == 1 and args.pmc == "live":
            got, why = pmc_live(args)
            if got:
                child_steps = (got.get("child", {}).get("fetch") or {}).get("steps") or steps_local
                fetch_b = got["FETCH_SIZE"] * 1024.0 * cal["fetch_factor"]
                write_b = got["WRITE_SIZE"] * 1024.0 * cal["write_factor"]
                per_step_dim = (fetch_b + write_b) / (child_steps * D)
                traffic = per_step_dim * steps_local * D     # the profiled launch's bytes per step x dim, at this launch's steps
                traffic_src = "live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload on this GPU"
                pmc_detail = {"fetch_bytes": fetch_b, "write_bytes": write_b, "hbm_bytes_per_step_dim": per_step_dim,
                              "profiled_launch_steps": child_steps, "calibration": cal}
                if "SQ_WAVE_CYCLES" in got and got["SQ_WAVE_CYCLES"]:
                    wc = got["SQ_WAVE_CYCLES"]
                    pmc_detail["issue"] = {
                        "note": "second bound: a wavefront's dependent chain (one wave per SIMD at 512 registers)",
                        "wait_any_frac": got.get("SQ_WAIT_ANY", 0.0) / wc,
                        "wait_inst_frac": got.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                        "active_inst_frac": got.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
                        "valu_active_frac": got.get("SQ_ACTIVE_INST_VALU", 0.0) / wc,
                        "valu_insts_per_leapfrog_wave": got.get("SQ_INSTS_VALU", 0.0) / max(1, child_steps),
                        "salu_insts_per_leapfrog_wave": got.get("SQ_INSTS_SALU", 0.0) / max(1, child_steps)}
                if got.get("skipped"):
                    pmc_detail["skipped_passes"] = got["skipped"]
            else:
                pmc_detail = {"live_failed": why}
        if traffic is None and args.pmc != "off":
            traffic, traffic_src = pmc_profile(steps_local * D)
        achieved = (traffic / kern_s / 1e9) if traffic else None
        bm = (bound_model(st_all, D, kern_s, traffic) if st_all is not None else None) or {}
        out = {
            "metric": "leapfrog-steps*dims/sec at 4096 chains x dim 1024 (post-warm-up NUTS draws)",
            "value": value, "unit": "leapfrog-steps*dims/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"K2: iid N(3,1) dim {D} x {C_} chains per GPU, maxdepth 10, per-chain diag mass matrix, "
                                   f"DiagNutsSettings defaults, num_tune {args.num_tune} (untimed), seed {args.seed}",
                       "chains_per_gpu": C_, "dim": D, "num_tune": args.num_tune, "parallelism": f"chains sharded x{world}, no collective",
                       "rng": "ChaCha8 stream + ziggurat normals (reference semantics)",
                       "draws_recorded": record, "recorded_buffers_verified": recorded_ok},
            "repeats": {"n": R, "reported": "median", "values": rates, "min": min(rates), "max": max(rates),
                        "ms_per_step": [el / args.steps * 1e3 for el, _, _, _ in reps]},
            "draws_per_sec_per_chain": args.steps / elapsed_max,
            "leapfrogs_per_draw": steps_total / (args.steps * C_ * world),
            "adaptation": {"value": tune_steps_total * D / t_tune_max, "unit": "leapfrog-steps*dims/s",
                           "draws": args.num_tune, "seconds": t_tune_max},
            "schema_version": SCHEMA_VERSION,
            "roofline": {**(one_bound(bm["binding"], bm["terms_ms"], kern_s, bm.get("necessary_valu_instr"), (bm.get("hbm") or {}).get("necessary_bytes"))
                            if bm.get("binding") else {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None}),
                         # bound / achieved / peak / unit / frac describe ONE bound (schema 6; rounds 4 - 5 printed HBM GB/s beside the issue bound's frac);
                         # what the kernel MOVED through HBM is `hbm_moved`
                         "hbm_moved": {"achieved_GBps": achieved, "peak_GBps": HBM_PEAK_GBS, "frac_moved": (achieved / HBM_PEAK_GBS) if achieved else None},
                         "binding": bm.get("binding"), "waste_ratio": (bm.get("hbm") or {}).get("moved_over_necessary"),
                         "t_min_ms": bm.get("t_min_ms"), "t_kernel_ms": kern_s * 1e3,
                         "frac_moved": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "frac_sec8d_model": algo_bytes / kern_s / 1e9 / HBM_PEAK_GBS,
                         "frac_definition": "frac = achieved / peak of the BINDING bound = t_min / t_kernel with t_min = max(necessary HBM bytes / 8 TB/s, necessary f64 VALU instructions / "
                                            "(1024 SIMDs x 2.4 GHz / 4)) — a bound: <= 1 and not raised by moving more bytes (bound_model has the "
                                            "enumeration); waste_ratio = HBM bytes moved / necessary; frac_moved = moved bytes (PMC) / launch time / "
                                            "8 TB/s = achieved / peak (a utilisation); frac_sec8d_model = 64 B x steps x dims / launch time / 8 TB/s "
                                            "(SURVEY 8(d)'s streaming model; exceeds 1 because live points, sigma, mu never leave the CU)",
                         "bound_model": bm,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": KERNEL, "kernel_ms_per_launch": kern_ms, "launches": 1,
                         "frac_of_measured_copy": (achieved / (cal.get("copy_GBps") or HBM_COPY_GBS)) if achieved else None,
                         "measured_copy_GBps": cal.get("copy_GBps"), "measured_triad_GBps": cal.get("triad_GBps"),
                         "algorithmic": {"bytes_per_step_dim": ALGO_BYTES_PER_STEP_DIM, "bytes_per_launch": algo_bytes,
                                         "rate_GBps": algo_bytes / kern_s / 1e9,
                                         "moved_over_algorithmic": (traffic / algo_bytes) if traffic else None,
                                         "note": "SURVEY §8(d)'s streaming model; not a bound here: live points, sigma, mu "
                                                 "stay in registers / LDS, HBM carries the tree's end points instead"},
                         "pmc": pmc_detail},
        }
        if world == 1:
            out["calibration"] = box_calibration("k2", value)
        if world == 1 and args.other_configs != "none":
            batch.close()                                    # (K2's 1.5 GB of chain state and the record buffers go first)
            batch = None
            del d_pos, d_st
            torch.cuda.empty_cache()
            out["other_configs"] = other_configs(args)
            try:
                out["low_rank_adaptation"] = low_rank_adaptation(args.seed)
            except Exception as e:  # noqa: BLE001  (the bench line must still be printed)
                out["low_rank_adaptation"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            cores, cores_note = usable_cores()
            try:
                out["cpu_baseline"] = cpu_baseline(args, cores)
                out["cpu_baseline"]["cores_note"] = cores_note
            except Exception as e:   # the bench line must still be printed
                out["cpu_baseline"] = {"value": None, "unit": "leapfrog-steps*dims/s", "cores": cores, "kind": "port",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out))
    if batch is not None:
        batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
