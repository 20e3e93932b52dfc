"""The known answers of every kernel instantiation (nuts_rs_amd/selftest_instantiations.json): the data is complete and
machine-independent (CPU), and the check rejects a library that does not reproduce an answer — at first use of the instantiation (GPU)."""
import copy
import json
import os

import numpy as np
import pytest

import nuts_rs_amd as N
from nuts_rs_amd import selftest, selftest_cases as SC


def test_every_case_has_an_answer_and_every_answer_a_case():
    gold = json.load(open(selftest.GOLDEN_INST))["cases"]
    ids = [SC.case_id(c) for c in SC.cases()]
    assert len(ids) == len(set(ids)) == 349
    assert set(ids) == set(gold)
    assert all(g["failed"] == 0 and len(g["sha256"]) == 64 and g["leapfrogs"] > 0 for g in gold.values())
    # every (density, family, tiling) the one-chain-per-block kernels are instantiated for has both ends of its range of dims (8 schools: dim 10 only)
    keys = {}
    for c in SC.cases():
        keys.setdefault((c["dens"], c["fam"], c["dpl"], c["w"]), []).append(c["dim"])
    assert all(len(v) == 2 or k[0] == "schools" for k, v in keys.items()), [k for k, v in keys.items() if len(v) != 2 and k[0] != "schools"]


def test_the_inputs_of_a_case_are_exactly_representable():
    """No exp / log / BLAS / LAPACK in the construction of a case's inputs: every number is a small dyadic rational, so the bits do not depend on
    the machine that builds them (round 6: numpy's matmul gave the GPU box other precision matrices than the box that generated the answers)."""
    for c in [dict(dens="mvn", fam="lr_frozen", dpl=8, w=1, dim=300, end="top"), dict(dens="diag", fam="nuts", dpl=4, w=1, dim=129, end="bottom")]:
        r = SC.make_run(N, c)
        arrays = [r["logp"].params] + ([np.concatenate([np.ravel(a) for a in r["transform"]])] if isinstance(r["transform"], tuple) else [])
        for a in arrays:
            m, e = np.frexp(a)
            assert np.all(np.abs(m * 2.0 ** 10 - np.round(m * 2.0 ** 10)) == 0), "an input with more than 10 mantissa bits"
    p = SC.make_run(N, dict(dens="mvn", fam="nuts", dpl=8, w=1, dim=300, end="top"))["logp"].params.reshape(300, 300)
    assert (p == p.T).all() and np.linalg.eigvalsh(p).min() > 0.5


@pytest.mark.gpu
def test_first_use_checks_the_instantiation_and_rejects_a_wrong_answer(monkeypatch):
    monkeypatch.delenv("NUTS_AMD_SELFTEST", raising=False)
    key = ("iid", "nuts", 2, 1)
    ids = [SC.case_id(c) for c in SC.cases() if (c["dens"], c["fam"], c["dpl"], c["w"]) == key]
    assert len(ids) == 2

    def engine():
        b = N.ChainBatch(N.DiagNutsSettings(num_chains=2, seed=5), N.LogpSpec.iid_normal(100, 0.5), 2)
        b.close()

    # the library in the tree reproduces its answers: the first engine of the instantiation runs the check, the second does not
    monkeypatch.setattr(selftest, "_checked", set())
    engine()
    assert key in selftest._checked
    # a library that answers differently (here: the answer on file is changed instead of the library) is rejected before anything is drawn
    bad = copy.deepcopy(selftest._gold_inst())
    bad[ids[1]]["sha256"] = "0" * 64
    monkeypatch.setattr(selftest, "_inst_gold", bad)
    monkeypatch.setattr(selftest, "_checked", set())
    with pytest.raises(selftest.SelfTestError) as e:
        engine()
    assert ids[1] in str(e.value) and "known answer" in str(e.value)
    # NUTS_AMD_SELFTEST=0 turns the check off
    monkeypatch.setenv("NUTS_AMD_SELFTEST", "0")
    monkeypatch.setattr(selftest, "_checked", set())
    engine()
    assert key not in selftest._checked


@pytest.mark.gpu
def test_run_all_names_every_failing_instantiation(monkeypatch):
    bad = copy.deepcopy(selftest._gold_inst())
    victims = [SC.case_id(c) for c in SC.cases() if (c["dens"], c["fam"], c["dpl"], c["w"]) == ("diag", "micro", 4, 1)]
    assert len(victims) == 2
    for v in victims:
        bad[v]["sha256"] = "f" * 64
    monkeypatch.setattr(selftest, "_inst_gold", bad)
    with pytest.raises(selftest.SelfTestError) as e:
        selftest.run_all(only="diag-micro-")
    assert all(v in str(e.value) for v in victims) and "2 of" in str(e.value)
    monkeypatch.setattr(selftest, "_inst_gold", None)
    assert selftest.run_all(only="diag-micro-4x1") == 2


def test_family_of_maps_every_settings_constructor_to_its_kernel_family():
    """`LowRankNutsSettings` / `LowRankMclmcSettings` are factories that return the Diag classes with LowRankSettings inside: the first-use check
    must still find the LrWrap instantiations (round 6: the first version keyed on the class name and let a wrong low-rank kernel through)."""
    assert SC.family_of(N.DiagNutsSettings()) == ["nuts"]
    assert SC.family_of(N.DiagNutsSettings(trajectory_kind=1)) == ["exact"]
    assert SC.family_of(N.DiagNutsSettings(trajectory_kind=2)) == ["micro"]
    assert SC.family_of(N.LowRankNutsSettings()) == ["lr_frozen", "lr_adapt"]
    assert SC.family_of(N.DiagMclmcSettings()) == ["mclmc"]
    assert SC.family_of(N.LowRankMclmcSettings()) == ["lr_mclmc"]


def test_answers_on_file_are_what_the_oracle_computes_today():
    """A sample of the 349 answers recomputed by the CPU oracle (no GPU): the data file and the oracle cannot drift apart unnoticed."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_selftest_golden", os.path.join(root, "tools", "gen_selftest_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    sys.modules["gen_selftest_golden"] = gen
    spec.loader.exec_module(gen)
    gold = json.load(open(selftest.GOLDEN_INST))["cases"]
    want = {"iid-nuts-2x1-dim2", "schools-nuts-2x1-dim10", "diag-micro-4x1-dim129", "funnel-lr_frozen-2x1-dim125", "iid-lr_adapt-4x1-dim129", "mvn-mclmc-2x1-dim2",
            "iid-nuts-16x1-dim513"}
    seen = set()
    for c in SC.cases():
        cid = SC.case_id(c)
        if cid not in want:
            continue
        pos, st, steps, failed = gen.oracle_answer(c)
        assert failed == gold[cid]["failed"] == 0
        assert steps == gold[cid]["leapfrogs"] and SC.digest(pos, st) == gold[cid]["sha256"], cid
        seen.add(cid)
    assert seen == want, want - seen
