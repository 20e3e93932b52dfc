"""HBM streaming probes on this GPU (nm_probe_bandwidth: 16 B/lane copy / triad / read / write) and, from rocprofv3
--pmc passes over the same probes, the FETCH_SIZE / WRITE_SIZE -> bytes calibration bench.py uses.

  python tools/hbm_probe.py run [--mib 1024] [--iters 20]          -> one JSON line: GB/s per probe
  python tools/hbm_probe.py calibrate <dir with fetch_results.db, write_results.db> <out.json>
     (the two databases come from `rocprofv3 --pmc FETCH_SIZE -d <dir> -o fetch -- python tools/hbm_probe.py run --iters 3`
      and the same with WRITE_SIZE / -o write; tools/hbm_calibrate.sh does all of it)
Arrays are 1 GiB each by default: 4x the 256 MiB Infinity Cache, so the counters see HBM, not the cache.
"""
import ctypes as C
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KINDS = {"copy": 0, "triad": 1, "read": 2, "write": 3, "copy_nt": 4}


def run(mib=1024, iters=20):
    from nuts_rs_amd import _lib
    L = _lib.load()
    out = {"bytes_per_array": mib << 20, "iters": iters}
    for name, kind in KINDS.items():
        ms, rd, wr = C.c_double(), C.c_uint64(), C.c_uint64()
        _lib.check(L.nm_probe_bandwidth(kind, mib << 20, iters, C.byref(ms), C.byref(rd), C.byref(wr)))
        out[name] = {"ms": ms.value, "bytes_read": rd.value, "bytes_written": wr.value,
                     "GBps": (rd.value + wr.value) / (ms.value * 1e-3) / 1e9}
    print(json.dumps(out))
    return out


def last_per_kind(db, counter):
    """{probe kind index: counter value of its last dispatch}"""
    c = sqlite3.connect(db)
    res = {}
    for name, cname, val in c.execute("select kernel_name, counter_name, value from counters_collection order by start"):
        m = re.search(r"probe_flat_kernel<\(?(?:int\))?(\d), *\(?(?:int\))?8>", name)     # the widest flat variant
        if m and cname == counter:
            res[int(m.group(1))] = val
    return res


def calibrate(d, out, mib=1024, rates=None):
    n = float(mib << 20)
    fetch = last_per_kind(os.path.join(d, "fetch_results.db"), "FETCH_SIZE")
    write = last_per_kind(os.path.join(d, "write_results.db"), "WRITE_SIZE")
    res = {"bytes_per_array": n, "unit_note": "FETCH_SIZE / WRITE_SIZE are reported in KiB",
           "raw_FETCH_SIZE_KiB": fetch, "raw_WRITE_SIZE_KiB": write}
    # bytes factor = known bytes / (counter * 1024), per probe with that side non-empty
    ff = {k: (2 * n if k == 1 else n) / (fetch[k] * 1024.0) for k in (0, 1, 2, 4) if fetch.get(k)}
    wf = {k: n / (write[k] * 1024.0) for k in (0, 1, 3, 4) if write.get(k)}
    res["fetch_factor_by_probe"], res["write_factor_by_probe"] = ff, wf
    # the engine's traffic is 16 B/lane loads, plain and nt stores: calibrate on copy (+ nt copy for the stores)
    res["fetch_factor"] = ff.get(0) or 2.0
    res["write_factor"] = (wf.get(0, 1.0) + wf.get(4, wf.get(0, 1.0))) / 2.0
    if rates and os.path.exists(rates):          # the un-profiled probe rates of the same box
        r = json.loads(open(rates).read().strip().splitlines()[-1])
        for k in KINDS:
            res[k + "_GBps"] = r[k]["GBps"]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    a = sys.argv[1:]
    opt = lambda k, d: int(a[a.index(k) + 1]) if k in a else d
    if a and a[0] == "calibrate":
        calibrate(a[1], a[2], opt("--mib", 1024), a[a.index("--rates") + 1] if "--rates" in a else None)
    else:
        run(opt("--mib", 1024), opt("--iters", 20))
