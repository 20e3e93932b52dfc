"""The build defends itself: nuts_rs_amd.build compiles every HIP unit — and build_density_module every USER density —
through one hipcc run that keeps the device assembly (-save-temps), scans it with tools/check_store_hazard.py in both modes
and fails on an unguarded instance.  Here: the unit / module compiled with the store guard taken out (-DNM_X_NO_STORE_GUARD) is REJECTED,
the regular ones pass."""
import os

import pytest

from nuts_rs_amd import build as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")


def test_density_module_build_rejects_an_unguarded_store(tmp_path):
    hdr = os.path.join(ROOT, "tests", "user_density", "my_diag_normal.hpp")
    assert os.path.exists(hdr)
    out = str(tmp_path / "ok.so")
    B.build_density_module(hdr, "MyDiagNormal", 1000, out)          # the regular build passes the scan
    assert os.path.exists(out)
    with pytest.raises(B.StoreHazardError) as e:
        B.build_density_module(hdr, "MyDiagNormal", 1000, str(tmp_path / "bad.so"), extra_flags=["-DNM_X_NO_STORE_GUARD"])
    assert "unguarded store-data hazard" in str(e.value) and "buffer_store_dwordx4" in str(e.value)
    assert not os.path.exists(tmp_path / "bad.so")


def test_density_module_of_a_small_tiling_rejects_out_of_line_device_calls(tmp_path, monkeypatch):
    """DESIGN §22, fourth incident: the units of the small tilings (<= 4 doubles per lane) and user modules of those tilings are built with every
    special function inlined, and the scan rejects such a unit if it still contains an s_swappc_b64 (here: the inlining switched off again behind
    the build's back)."""
    hdr = os.path.join(ROOT, "tests", "user_density", "my_diag_normal.hpp")
    monkeypatch.setenv("NM_MODULE_EXTRA_FLAGS", "-DNM_DETMATH_INLINE=0 -Wno-macro-redefined")
    with pytest.raises(B.StoreHazardError) as e:
        B.build_density_module(hdr, "MyDiagNormal", 100, str(tmp_path / "calls.so"))
    assert "out-of-line device call" in str(e.value)
    assert not os.path.exists(tmp_path / "calls.so")
    monkeypatch.delenv("NM_MODULE_EXTRA_FLAGS")
    B.build_density_module(hdr, "MyDiagNormal", 100, str(tmp_path / "ok.so"))          # the regular build of the same module passes
    assert os.path.exists(tmp_path / "ok.so")
