"""Scan device assembly for out-of-line calls inside the kernels that carry SEVERAL chains per wavefront.

In those kernels (grpN::nuts_group_draw_kernel, lane::*) every branch of the tree logic is non-uniform over the wavefront: a callee entered there runs with a partial exec mask, and round 4 saw twice that lanes of the
WAITING chains came back from such a call with live registers changed (a callee that parks scalar registers in vector lanes
— v_writelane ignores exec — or saves "callee-saved" VGPRs under the caller's partial mask).  The engine's rule since then: nothing out of
line in these kernels.  This tool makes the rule a build step: an `s_swappc_b64` inside a function whose name matches --kernels fails.

(The lockstep matrix-core kernel calls its state machines and refresh units out of line ON PURPOSE, under branches on the wavefront's
index — uniform over the wavefront, the whole exec mask enters the callee — so it is not in the default pattern.)

  python tools/check_divergent_calls.py file.s [--kernels REGEX]      (default: nuts_group_draw_kernel|nuts_lane)"""
import re
import sys

path = sys.argv[1]
pat = re.compile(sys.argv[sys.argv.index("--kernels") + 1] if "--kernels" in sys.argv else r"nuts_group_draw_kernel|nuts_lane")
name, bad = None, 0
for i, l in enumerate(open(path)):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        name = m.group(1)
        continue
    if name and pat.search(name) and re.match(r"^\s+s_swappc_b64\b", l):
        bad += 1
        if bad <= 20:
            print(f"{path}:{i + 1}: {l.strip()}   (kernel {name})")
print(f"{path}: {bad} out-of-line calls inside several-chains-per-wavefront kernels")
sys.exit(1 if bad else 0)
