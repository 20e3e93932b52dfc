"""The branch-free special functions and the one-routine merge arithmetic of csrc/dev_math.hpp (round 5: exp_sl, log_sl, log1p_sl,
log1p_unit, merge_math_impl) against the general-purpose routines of rounds 1-4 they replace (dexp_branchy / dlog_branchy /
dlog1p_branchy in merge_weights' original order: reference src/nuts.rs:172-207, src/math/util.rs:6-19) — bit for bit on the host,
over special values, sub-normals, table boundaries and 13 million random operands, Bernoulli words AT the threshold included.
The oracle keeps the rounds-1-4 sequences (oracle/nmo_math.hpp), so the GPU parity suite checks the device side of the same claim."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "merge_math_check.hip")
EXE = os.path.join(ROOT, "tests", "cpp", "merge_math_check")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc compiles the host harness (it includes dev_math.hpp)")
def test_branch_free_merge_arithmetic_equals_the_general_routines_bit_for_bit():
    deps = [SRC, os.path.join(ROOT, "nuts_rs_amd", "csrc", "dev_math.hpp"), os.path.join(ROOT, "nuts_rs_amd", "csrc", "detmath_tables.hpp")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call([HIPCC, "-O2", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", SRC, "-o", EXE])
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout
