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
