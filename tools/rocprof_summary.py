"""Summarise rocprofv3 rocpd databases into small text/JSON files for profiles/.

  python tools/rocprof_summary.py trace <trace_results.db> <out.txt>
  python tools/rocprof_summary.py pmc <dir with fetch/write/tcc/trace _results.db> <out.json> [kernel substring]
"""
import json
import os
import sqlite3
import sys


def trace(db, out):
    c = sqlite3.connect(db)
    lines = ["# rocprofv3 --kernel-trace --stats : per-kernel totals (top_kernels view; durations in microseconds)",
             "# (under bench.py the draw kernel is dispatched three times: the adaptation phase, the untimed warm-up steps, the timed steps —",
             "#  the LAST dispatch in the per-dispatch list is the launch `roofline.kernel_ms_per_launch` refers to; avg_us averages all of them)"]
    for r in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        lines.append("%s | calls=%d | total_us=%.1f | avg_us=%.1f | pct=%.3f" % r)
    lines += ["", "# per dispatch (durations in ns): kernel | grid | wg | dur_ns | vgpr | agpr | sgpr | lds | scratch"]
    for r in c.execute("select name, grid_x, workgroup_x, duration, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, "
                       "scratch_size from kernels order by start"):
        lines.append("%s | grid=%d | wg=%d | dur_ns=%d | vgpr=%d | agpr=%d | sgpr=%d | lds=%d | scratch=%d" % r)
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


def counters(db, sub):
    c = sqlite3.connect(db)
    res = {}
    for name, cname, val in c.execute("select kernel_name, counter_name, value from counters_collection order by start"):
        if sub in name:
            res.setdefault(cname, []).append(val)
    return res


def bench_line(logdir, tag):
    """the JSON line bench.py printed in the pass `tag` (its stdout log), or None"""
    try:
        for line in open(os.path.join(logdir, tag + "_stdout.log")):
            if line.startswith("{") and '"metric"' in line:
                return json.loads(line)
    except OSError:
        pass
    return None


def pmc(d, out, sub="nuts_draw_kernel", logdir=None):
    """The LAST dispatch of the kernel is bench.py's timed launch (tune, warm-up, timed)."""
    c = sqlite3.connect(os.path.join(d, "trace_results.db"))
    durs = [r[0] for r in c.execute("select duration from kernels where name like ? order by start", ("%" + sub + "%",))]
    res = {"kernel": sub, "dispatches": len(durs), "timed_launch_duration_ns": durs[-1] if durs else None}
    for f in ("fetch", "write", "tcc", "sq"):
        p = os.path.join(d, f + "_results.db")
        if os.path.exists(p):
            for k, v in counters(p, sub).items():
                res[k] = v[-1]
    if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
        # MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide
        # (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE has no documented correction.
        res["fetch_bytes_corrected"] = res["FETCH_SIZE"] * 1024 * 2
        res["write_bytes"] = res["WRITE_SIZE"] * 1024
        res["hbm_bytes_per_launch"] = res["fetch_bytes_corrected"] + res["write_bytes"]
        if res["timed_launch_duration_ns"]:
            res["hbm_GBps_over_kernel"] = res["hbm_bytes_per_launch"] / res["timed_launch_duration_ns"]
    # per (leapfrog-step x dim), so that a run with another step count can scale it (bench.py --pmc profile)
    line = bench_line(logdir, "fetch") if logdir else None
    if line and "hbm_bytes_per_launch" in res:
        steps_dims = line["leapfrogs_per_draw"] * line["steps"] * line["config"]["chains_per_gpu"] * line["config"]["dim"]
        res["workload"] = {"chains": line["config"]["chains_per_gpu"], "dim": line["config"]["dim"], "steps": line["steps"],
                           "draws_recorded": line["config"].get("draws_recorded")}
        res["leapfrog_steps_x_dims_of_launch"] = steps_dims
        res["hbm_bytes_per_step_dim"] = res["hbm_bytes_per_launch"] / steps_dims
        res["hbm_bytes_per_chain_draw"] = res["hbm_bytes_per_launch"] / (line["steps"] * line["config"]["chains_per_gpu"])
    if res.get("SQ_WAVE_CYCLES"):
        wc = res["SQ_WAVE_CYCLES"]
        res["issue"] = {k: res.get(c, 0.0) / wc for k, c in (("wait_any_frac", "SQ_WAIT_ANY"), ("wait_inst_frac", "SQ_WAIT_INST_ANY"),
                                                          ("active_inst_frac", "SQ_ACTIVE_INST_ANY"), ("valu_active_frac", "SQ_ACTIVE_INST_VALU"))}
    if "TCC_HIT_sum" in res and "TCC_MISS_sum" in res:
        res["l2_hit_rate"] = res["TCC_HIT_sum"] / (res["TCC_HIT_sum"] + res["TCC_MISS_sum"])
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


def k5(d, out, sub, logdir):
    """counters of the LAST dispatch of the kernel `sub` (bench_k5.py's timed launch) + the bench line of the un-profiled run"""
    c = sqlite3.connect(os.path.join(d, "trace_results.db"))
    durs = [r[0] for r in c.execute("select duration from kernels where name like ? order by start", ("%" + sub + "%",))]
    res = {"kernel": sub, "dispatches": len(durs), "timed_launch_duration_ns": durs[-1] if durs else None}
    for f in ("mfma", "sq", "lds", "fetch", "write", "tcc"):
        p = os.path.join(d, f + "_results.db")
        if os.path.exists(p):
            for k, v in counters(p, sub).items():
                res[k] = v[-1]
    try:
        res["bench"] = json.loads(open(os.path.join(logdir, "bench.json")).read().strip().splitlines()[-1])
    except Exception:
        pass
    g = res.get
    if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("GRBM_GUI_ACTIVE"):
        # MfmaUtil of rocprofv3's derived metrics: busy cycles summed over SIMDs / (kernel cycles x SIMDs); 256 CUs x 4 SIMDs.
        # GRBM_GUI_ACTIVE comes back SUMMED over the 8 XCDs, so kernel cycles = GRBM_GUI_ACTIVE / 8.
        res["mfma_util"] = res["SQ_VALU_MFMA_BUSY_CYCLES"] / (res["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        res["mfma_util_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles); GRBM_GUI_ACTIVE is summed over the 8 XCDs"
    if g("SQ_INSTS_VALU_MFMA_MOPS_F64") and res["timed_launch_duration_ns"]:
        res["mfma_f64_TFLOPs"] = res["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0 / res["timed_launch_duration_ns"] / 1e3
        res["mfma_f64_frac_of_78.6TF"] = res["mfma_f64_TFLOPs"] / 78.6
    if g("SQ_WAVE_CYCLES"):
        wc = res["SQ_WAVE_CYCLES"]
        res["issue"] = {k: res.get(cn, 0.0) / wc for k, cn in (("wait_any_frac", "SQ_WAIT_ANY"), ("wait_inst_frac", "SQ_WAIT_INST_ANY"),
                        ("active_inst_frac", "SQ_ACTIVE_INST_ANY"), ("valu_active_frac", "SQ_ACTIVE_INST_VALU"), ("lds_active_frac", "SQ_ACTIVE_INST_LDS"))}
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
        res["lds_bank_conflict_frac"] = res["SQ_LDS_BANK_CONFLICT"] / res["SQ_LDS_IDX_ACTIVE"]
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        res["hbm_bytes"] = res["FETCH_SIZE"] * 2048.0 + res["WRITE_SIZE"] * 1024.0
        if res["timed_launch_duration_ns"]:
            res["hbm_GBps"] = res["hbm_bytes"] / res["timed_launch_duration_ns"]
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum"):
        res["l2_hit_rate"] = res["TCC_HIT_sum"] / (res["TCC_HIT_sum"] + res["TCC_MISS_sum"])
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        trace(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "k5":
        k5(*sys.argv[2:6])
    else:
        pmc(sys.argv[2], sys.argv[3], *(sys.argv[4:6]))
