import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch bundles its own HIP runtime; when libnuts_amd (system ROCm) touches the device first, torch's later
    # initialisation finds no GPU.  Tests that use torch tensors next to the engine need torch to initialise first.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:      # noqa: BLE001 — CPU box, or no torch: nothing to order
        pass


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; built on demand with g++)."""
    from oracle import oracle as O
    O.lib()
    return O


# Collection order: under `-x` a peripheral seam must not be able to hide the hot path.  The core parity
# suites run first — units of the fused kernels' building blocks, then the whole-chain parity sweeps on every BASELINE config —
# then the rows of SURVEY §8 in the order the scope table lists them; files not named keep their alphabetical order at the end.
_ORDER = [
    "test_entry", "test_abi", "test_stream_discipline", "test_integration_binding", "test_oracle_golden",   # CPU: boundary + oracle pins
    "test_gpu_units", "test_gpu_parity", "test_gpu_statistics",            # §8(a) T / D / A rows, (d) configs
    "test_gpu_lane_chains",                                                 # §8(a) one chain per lane (dim <= 16)
    "test_cpp_host_api", "test_density_module", "test_gpu_host_callback",  # §8(b) driver seam, user densities
    "test_gpu_lowrank", "test_gpu_tile_diag",                               # §8(f)2
    "test_gpu_trajectory_kinds", "test_gpu_mclmc",                          # §8(f)4
    "test_gpu_wide_chains", "test_gpu_distributed", "test_distributed_cpu", "test_controller",
    "test_gpu_math_seam",                                                    # §8(b) per-vector seam: last of the named ones
]


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(_ORDER)}

    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return rank.get(mod, len(_ORDER))

    items.sort(key=key)       # stable: the order inside a file is kept
