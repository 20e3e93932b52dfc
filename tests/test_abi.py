"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/nuts_amd.h declares,
refuses to compute without a GPU (no CPU fallback), and the host-side mirror carries the reference's defaults."""
import ctypes as C
import json
import os
import re

import pytest

import nuts_rs_amd as N
from nuts_rs_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "nuts_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = N.load_library()
    names = declared_symbols()
    assert len(names) >= 27
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/nuts_amd.h but not exported"
    assert set(names) == set(_lib.ABI_SYMBOLS)
    header = open(os.path.join(ROOT, "include", "nuts_amd.h")).read()
    assert L.nm_abi_version() == int(header.split("#define NM_ABI_VERSION")[1].split()[0])


def test_struct_layouts_match_header():
    # every field is 8 bytes wide (header contract)
    assert C.sizeof(_lib.NmSettings) == 8 * len(_lib.NmSettings._fields_) == 8 * 49
    assert _lib.STATS_DTYPE.itemsize == 8 * 24
    assert C.sizeof(_lib.NmDrawOutputs) == 8 * 16
    assert C.sizeof(_lib.NmEngineConfig) == 72 and C.sizeof(_lib.NmLogpSpec) == 64


def test_defaults_are_the_references():
    kats = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")))["default_settings"]
    s = _lib.NmSettings()
    N.load_library().nm_settings_default(C.byref(s))
    c = N.DiagNutsSettings().to_c()
    for k, v in kats.items():
        if k != "source":
            assert getattr(s, k) == v and getattr(c, k) == v, k


def test_host_helpers_need_no_gpu(oracle):
    L = N.load_library()
    key = (C.c_uint8 * 32)()
    assert L.nm_chain_rng_key(5, 7, key) == 0
    assert bytes(key) == oracle.chain_key(5, 7)                      # same published algorithms, independent code
    import numpy as np
    x = np.empty((3, 6))
    assert L.nm_init_positions_uniform(9, 2, 3, 6, x.ctypes.data) == 0
    assert (x == oracle.init_positions_uniform(9, 2, 3, 6)).all()


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(N.NutsAmdError) as e:
        N.ChainBatch(N.DiagNutsSettings(), N.LogpSpec.iid_normal(10), 4)
    assert e.value.status == 2 and "no CPU fallback" in str(e.value)


def test_argument_validation():
    L = N.load_library()
    h = C.c_void_p()
    s = N.DiagNutsSettings().to_c()
    spec = N.LogpSpec.funnel(0).to_c()
    assert L.nm_engine_create(C.byref(s), C.byref(spec), 4, None, C.byref(h)) == 1      # dim 0 is served for the iid normal only
    spec = N.LogpSpec.iid_normal(10).to_c()
    assert L.nm_engine_create(C.byref(s), C.byref(spec), 0, None, C.byref(h)) == 1      # no chains
    s2 = N.DiagNutsSettings(maxdepth=40).to_c()
    assert L.nm_engine_create(C.byref(s2), C.byref(spec), 4, None, C.byref(h)) == 4     # unsupported
    s3 = N.DiagNutsSettings().to_c()
    s3.step_size_method = 7                                                             # not a StepSizeAdaptMethod
    assert L.nm_engine_create(C.byref(s3), C.byref(spec), 4, None, C.byref(h)) == 1
    s4 = N.DiagNutsSettings(num_tune=0).to_c()                                          # reference asserts early_end < num_tune
    assert L.nm_engine_create(C.byref(s4), C.byref(spec), 4, None, C.byref(h)) == 1
    assert b"early_end" in L.nm_last_error()
