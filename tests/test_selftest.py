"""nuts_rs_amd.selftest: the known answers are data computed by the oracle (CPU test: regenerating them gives the
committed file), the built library reproduces them on every kernel family (GPU test), and a library that does not is reported."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_golden_file_is_what_the_oracle_computes(oracle, tmp_path):
    import nuts_rs_amd as N
    from nuts_rs_amd import selftest
    from helpers import oracle_settings
    gold = json.load(open(selftest.GOLDEN))["cases"]
    assert set(gold) == set(selftest.CASES)
    for name, (mk, chains, seed, draws, threads) in selftest.CASES.items():
        logp = mk()
        s = N.DiagNutsSettings(num_chains=chains, seed=seed, num_tune=20, num_draws=draws)
        x0 = oracle.init_positions_uniform(s.seed, 0, chains, logp.dim)
        pos, st, steps, failed = oracle.run(oracle_settings(oracle, s), logp.kind, logp.dim, logp.params, oracle.gpu_cfg(threads), chains, x0, draws, n_threads=2)
        g = gold[name]
        assert failed == 0 and steps == g["total_leapfrogs"] and st["n_steps"].astype(int).tolist() == g["n_steps"]
        assert [format(int(v), "016x") for v in pos[-1].reshape(-1).view(np.uint64)] == g["last_position_bits"]


@pytest.mark.gpu
def test_selftest_passes_on_the_built_library_and_reports_a_wrong_answer(tmp_path, monkeypatch):
    from nuts_rs_amd import selftest
    assert selftest.run() == len(selftest.CASES) * len(selftest.FAMILIES)
    # a corrupted known answer stands in for a library that computes something else
    bad = json.load(open(selftest.GOLDEN))
    bad["cases"]["eight_schools"]["n_steps"][7][3] += 1
    f = tmp_path / "bad.json"
    f.write_text(json.dumps(bad))
    monkeypatch.setattr(selftest, "GOLDEN", str(f))
    with pytest.raises(selftest.SelfTestError) as e:
        selftest.run()
    assert "eight_schools" in str(e.value) and "[7, 3]" in str(e.value)
