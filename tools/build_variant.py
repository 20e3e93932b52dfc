#!/usr/bin/env python3
"""A tuning build of some translation units with extra flags, linked with the regular objects of the others:
    python tools/build_variant.py <tag> "<extra flags>" unit[@variant] [unit[@variant] ...]
-> nuts_rs_amd/libnuts_amd_<tag>.so (run a tool with NUTS_AMD_LIB=<that file>).  The units go through the same compile path as the regular
build (assembly scans, exec-spill repair: nuts_rs_amd.build._compile_scanned) with their variant's flags (@small / @large / @inl)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nuts_rs_amd import build as B  # noqa: E402


def main():
    tag, extra, units = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
    out_dir = os.path.join(B.OBJ, tag)
    os.makedirs(out_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    for u in units:
        assert u in B.UNITS, f"{u} is not a unit of the build ({[x for x in B.UNITS if x.startswith(u.split('@')[0])]})"

    def one(u):
        variant = u.partition("@")[2]
        obj = os.path.join(out_dir, os.path.basename(B._obj(u)))
        flags = B.FLAGS + B.VARIANT_FLAGS[variant] + extra
        B._compile_scanned(hipcc, flags, B._src(u), obj, u + " [" + tag + "]", allow_calls=variant not in B.NO_CALL_VARIANTS)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 1)) as ex:
        new = list(ex.map(one, units))
    objs = [B._obj(u) for u in B.UNITS if u not in units] + new
    missing = [o for o in objs if not os.path.exists(o)]
    assert not missing, f"regular objects missing (run python -m nuts_rs_amd.build first): {missing[:3]}"
    lib = os.path.join(B.HERE, f"libnuts_amd_{tag}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
    print("built", lib)


if __name__ == "__main__":
    main()
