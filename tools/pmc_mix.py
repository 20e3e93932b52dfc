"""Dynamic instruction mix of the LAST dispatch of a kernel (tools/mix_driver.py's timed launch): several rocprofv3 --pmc passes
(counters only, no tracing option combined), normalised per leapfrog of one chain's wavefront.
  python tools/pmc_mix.py <out.json> <kernel substring> <waves_per_chain_denominator> -- python tools/mix_driver.py k2one
`denominator`: chains per wavefront (1 for the wave kernel, 8 for the 8-lane kernels) -> instructions per leapfrog-wave = count / (leapfrogs / denominator)."""
import json, os, sqlite3, subprocess, sys, glob, shutil
out, sub, denom = sys.argv[1], sys.argv[2], float(sys.argv[3])
cmd = sys.argv[sys.argv.index("--") + 1:]
GROUPS = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA"],
    ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH"],
    ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT"],
    ["SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_SALU", "SQ_INST_CYCLES_SMEM", "SQ_WAIT_INST_LDS"],
    ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_SENDMSG", "SQ_INSTS_VSKIPPED", "SQ_INSTS_GDS", "SQ_INSTS_FLAT"],
    ["SQ_IFETCH", "SQ_INSTS_WAVE32", "SQ_THREAD_CYCLES_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_EXP_GDS", "SQ_WAVES", "SQ_INSTS_VALU_MFMA_MOPS_F64"],
]
res = {"kernel": sub, "cmd": cmd, "counters": {}, "errors": []}
line = None
for gi, g in enumerate(GROUPS):
    d = "/tmp/pmc_mix_%d" % gi
    shutil.rmtree(d, ignore_errors=True)
    p = subprocess.run(["rocprofv3", "--pmc"] + g + ["-d", d, "-o", "m", "--"] + cmd, capture_output=True, text=True, env=dict(os.environ, TMPDIR="/tmp"))
    for l in p.stdout.splitlines():
        if l.startswith("{") and "mix_driver" in l:
            line = json.loads(l)
    dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    if not dbs:
        res["errors"].append({"group": g, "stderr": p.stderr[-600:]})
        # a group with an unknown counter fails as a whole: retry one by one
        for cn in g:
            shutil.rmtree(d, ignore_errors=True)
            p = subprocess.run(["rocprofv3", "--pmc", cn, "-d", d, "-o", "m", "--"] + cmd, capture_output=True, text=True, env=dict(os.environ, TMPDIR="/tmp"))
            for db in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
                c = sqlite3.connect(db)
                for name, cname, val in c.execute("select kernel_name, counter_name, value from counters_collection order by start"):
                    if sub in name:
                        res["counters"][cname] = val
        continue
    c = sqlite3.connect(dbs[0])
    for name, cname, val in c.execute("select kernel_name, counter_name, value from counters_collection order by start"):
        if sub in name:
            res["counters"][cname] = val          # the last dispatch wins
res["workload"] = line
if line:
    lw = line["leapfrogs"] / denom
    res["per_leapfrog_wave"] = {k: v / lw for k, v in res["counters"].items()}
    wc = res["counters"].get("SQ_WAVE_CYCLES")
    if wc:
        res["frac_of_wave_cycles"] = {k: v / wc for k, v in res["counters"].items() if k.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_INST_CYCLES"))}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res.get("per_leapfrog_wave"), indent=1))
print(json.dumps(res.get("frac_of_wave_cycles"), indent=1))
print("errors:", res["errors"])
