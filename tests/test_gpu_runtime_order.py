"""One HIP runtime per process whatever the import order (round 4: `build()` followed by `smoke()` in one process loaded the
engine's library before torch, two copies of libamdhip64 were then alive and the engine's saw no device)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ENGINE_FIRST = """
import nuts_rs_amd as N
N.load_library().nm_abi_version()          # the engine's library first ...
import torch                                # ... then torch
assert torch.cuda.is_available()
s = N.DiagNutsSettings(num_chains=4, seed=1, num_tune=5, num_draws=5)
b = N.ChainBatch(s, N.LogpSpec.iid_normal(10, 3.0), 4, device=0)
b.set_position(b.init_positions_uniform())
pos, st = b.draw_many(10)
b.close()
rt = sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l))
assert len(rt) == 1, rt
print('ok', int(st['n_steps'].sum()))
"""


def _hip_runtimes_loaded(code):
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def test_engine_library_shares_torchs_hip_runtime_on_cpu():
    """Loading the library alone already maps torch's copy of the runtime (no torch import), and importing torch afterwards adds none."""
    code = ("import nuts_rs_amd as N, sys\nN.load_library()\nassert 'torch' not in sys.modules\nimport torch\n"
            "print(len(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l)))")
    assert _hip_runtimes_loaded(code).strip() == "1"


@pytest.mark.gpu
def test_engine_before_torch_in_one_process():
    assert _hip_runtimes_loaded(ENGINE_FIRST).startswith("ok")


@pytest.mark.gpu
def test_library_check_then_smoke_in_one_process():
    """What `python __graft_entry__.py smoke` does after compiling: the ABI check of build() (loads the library, no torch), then
    smoke() (imports torch) in the same process."""
    code = ("import __graft_entry__ as g\nimport nuts_rs_amd\nfrom nuts_rs_amd import _lib\nL = nuts_rs_amd.load_library()\n"
            "[getattr(L, n) for n in _lib.ABI_SYMBOLS]\ng.smoke()\n")
    assert "smoke ok" in _hip_runtimes_loaded(code)
