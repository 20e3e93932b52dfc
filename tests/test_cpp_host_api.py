"""The C++ host side (include/nuts_amd.hpp: DiagNutsSettings / ChainBatch / Progress / Sampler mirroring the reference's
interface over the C ABI).  tests/cpp/host_api_demo.cpp is the reference's README example written against it.
CPU: it compiles with g++, links against libnuts_amd.so and fails loudly without a device.  GPU: its draws equal the
oracle's for BASELINE config K1, through the per-draw `draw()` loop and through the Sampler control plane."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "host_api_demo")


def build_demo():
    src = os.path.join(ROOT, "tests", "cpp", "host_api_demo.cpp")
    deps = [src, os.path.join(ROOT, "include", "nuts_amd.hpp"), os.path.join(ROOT, "include", "nuts_amd.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        libdir = os.path.join(ROOT, "nuts_rs_amd")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), src,
                               "-o", EXE, "-L", libdir, "-lnuts_amd", f"-Wl,-rpath,{libdir}", "-pthread"])
    return EXE


def test_cpp_host_api_builds_and_refuses_to_run_without_a_gpu():
    import torch
    exe = build_demo()
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 3 and "NM_ERR_NO_DEVICE" in r.stdout and "no CPU fallback" in r.stdout


@pytest.mark.gpu
def test_cpp_host_api_k1_matches_oracle(oracle, tmp_path):
    exe = build_demo()
    out = tmp_path / "k1.bin"
    r = subprocess.run([exe, str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    pos = np.fromfile(out).reshape(1400, 4, 10)
    s = oracle.default_settings(seed=0, num_chains=4)
    pos_o, st_o, steps, failed = oracle.run(s, oracle.LOGP_IID_NORMAL, 10, [3.0], oracle.gpu_cfg(64), 4, np.zeros((4, 10)), 1400)
    assert failed == 0 and (pos.view(np.uint64) == pos_o.view(np.uint64)).all()
    assert f"{steps} leapfrogs" in r.stdout


def test_cpp_settings_types_defaults():
    src = os.path.join(ROOT, "tests", "cpp", "settings_check.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "settings_check")
    libdir = os.path.join(ROOT, "nuts_rs_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), src,
                           "-o", exe, "-L", libdir, "-lnuts_amd", f"-Wl,-rpath,{libdir}", "-pthread"])
    assert subprocess.run([exe]).returncode == 0
