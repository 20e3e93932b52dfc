"""Print registers / scratch / LDS of every kernel of one translation unit (from the code-object metadata notes).

  python tools/kernel_resources.py [nuts_rs_amd/csrc/build/kern_iid_normal.o] [name substring]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else "nuts_rs_amd/csrc/build/kern_iid_normal.o"
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as d:
        out, fat = os.path.join(d, "co"), os.path.join(d, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out}"],
                              stderr=subprocess.DEVNULL)
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", out], text=True)
    for blk in notes.split("  - .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        dem = subprocess.check_output(["c++filt", name], text=True).strip()
        if sub not in dem:
            continue
        g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
        print(f"{dem[:90]:90s} vgpr={g('vgpr_count'):>4s} sgpr={g('sgpr_count'):>4s} scratch={g('private_segment_fixed_size'):>5s} "
              f"lds={g('group_segment_fixed_size'):>6s} vspill={g('vgpr_spill_count')}")


if __name__ == "__main__":
    main()
