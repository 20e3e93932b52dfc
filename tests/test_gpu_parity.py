"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle on the same seeds — draw for draw,
bit for bit (positions and every sampler statistic).  north_star's tolerance is 1e-9 relative; the engine is
built to do better: identical arithmetic contract => identical bits => identical discrete decisions."""
import numpy as np
import pytest

import nuts_rs_amd as N
from helpers import assert_bit_exact, run_engine, run_oracle

pytestmark = pytest.mark.gpu


def test_k1_readme_config_bit_exact(oracle):
    """BASELINE configs[0]: 10-dim iid N(3,1), 4 chains, DiagNutsSettings::default(), x0 = zeros (README.md:44-79)."""
    s = N.DiagNutsSettings(num_chains=4, seed=0)
    logp = N.LogpSpec.iid_normal(10, 3.0)
    x0 = np.zeros((4, 10))
    n = s.num_tune + s.num_draws
    pos_g, st_g, ex = run_engine(s, logp, 4, x0, n)
    pos_o, st_o, steps, failed = run_oracle(oracle, s, logp, 4, x0, n)
    assert failed == 0 and (ex["status"] == 0).all()
    assert_bit_exact(pos_g, st_g, pos_o, st_o)
    assert ex["counters"]["total_leapfrogs"] == steps
    # behavioural envelope of the reference's own tests (src/adapt_strategy.rs:367-435): converged, no divergences
    assert st_g["diverging"].sum() == 0
    assert abs(pos_g[s.num_tune:].mean() - 3.0) < 0.1
