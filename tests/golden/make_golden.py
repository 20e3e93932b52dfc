"""Generates tests/golden/*.json.  Run from the repo root: `python tests/golden/make_golden.py`.

Two kinds of fixtures:
  reference_kats.json  — inputs and expected outputs TRANSCRIBED FROM THE REFERENCE'S OWN TESTS (data only, with
                         file:line of each test) plus published cipher test vectors.  These pin the oracle.
  oracle_k1_pins.json  — a few seeded draws of BASELINE config K1 produced by the CPU oracle (gpu arithmetic
                         contract).  The reference holds no fixed-seed golden draws and cannot be built here
                         (Rust), so these pin the oracle/engine against regressions, not the crate.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_kats():
    k = {}
    # ChaCha keystream, all-zero key / nonce / counter: ChaCha20 from RFC 7539 §2.3.2-style zero-key block
    # (also draft-strombergson-chacha-test-vectors-01 TC1), ChaCha8 from the same draft (TC1, 8 rounds).
    k["chacha_zero_key_block0"] = {
        "source": "draft-strombergson-chacha-test-vectors-01, TC1 (256-bit all-zero key, zero IV), block 0",
        "rounds20": "76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                    "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586",
        "rounds8": "3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e"
                   "984ce172b9216f419f445367456d5619314a42a3da86b001387bfdb80e0cfe42",
    }
    # reference src/transform/mod.rs:175-250  test_diag_transform_position_and_gradient
    k["diag_transform_position_and_gradient"] = {
        "source": "/root/reference/src/transform/mod.rs:175-250",
        "sigma2": [1.0, 4.0, 9.0], "draw_mean": [0.0, 0.0, 0.0], "grad_mean": [0.0, 0.0, 0.0],
        "x": [1.0, 2.0, 3.0], "expect_z": [1.0, 1.0, 1.0], "expect_gz": [-1.0, -1.0, -1.0], "tol": 1e-12,
    }
    # reference src/transform/mod.rs:254-318  test_diag_round_trip
    k["diag_round_trip"] = {
        "source": "/root/reference/src/transform/mod.rs:254-318",
        "sigma2": [2.0, 0.5, 3.0], "draw_mean": [0.0, 0.0, 0.0], "grad_mean": [0.0, 0.0, 0.0],
        "x": [1.5, -0.3, 2.1], "tol": 1e-12,
    }
    # reference src/transform/mod.rs:322-377  test_diag_nonzero_mean
    k["diag_nonzero_mean"] = {
        "source": "/root/reference/src/transform/mod.rs:322-377",
        "sigma2": [4.0, 1.0, 9.0], "draw_mean": [3.0, -1.0, 2.0], "grad_mean": [0.0, 0.0, 0.0],
        "x": [5.0, 0.0, 5.0], "expect_z": [1.0, 1.0, 1.0], "tol": 1e-12,
    }
    # reference src/math/util.rs:880-890 (check_logaddexp), :964-968 (check_neginf), regression seed
    # proptest-regressions/math.txt:7
    k["logaddexp"] = {
        "source": "/root/reference/src/math/util.rs:880-890,:964-968; proptest-regressions/math.txt:7",
        "neginf_cases": [[float("-inf"), 2.0, 2.0], [2.0, float("-inf"), 2.0]],
        "regression_xy": [[4.8329699435311735, 9.38911339170414]],
        "range": [-10.0, 10.0], "abs_tol_vs_naive": 1e-10,
    }
    # shrunk failing inputs the reference keeps for its SIMD primitives (proptest-regressions/math.txt:8-12);
    # expected values are the scalar formulas of src/math/util.rs:893-961 within max_ulps = 32
    k["primitive_regressions"] = {
        "source": "/root/reference/proptest-regressions/math.txt:8-12; formulas src/math/util.rs:893-961",
        "axpy": [{"x": [2.9394791070664547e110, 0.0], "y": [float("inf"), 0.0], "a": -2.4153502104628106e222},
                 {"x": [0.0, 0.0, 0.0, 1.2271235629394547e205, 0.0, 0.0, -0.0, 0.0],
                  "y": [0.0, 0.0, 0.0, 7.121658452243713e81, 0.0, 0.0, 0.0, 0.0], "a": -6.261465657118442e-124}],
        "scalar_prods3": [{"x1": [0.0], "x2": [0.0], "x3": [-4.0946726283401733e139], "y1": [0.0],
                           "y2": [1.3157422010991668e73]}],
        "axpy_out": [{"a": 1.033664102276113e155, "x": [-1.847508293460042e-54, 0.0, 0.0],
                      "y": [1.8293708670672727e101, 0.0, 0.0]}],
        "vector_dot": [{"x": [0.0, 0.0, 0.0, -0.0], "y": [-0.0, 0.0, 0.0, float("inf")]}],
        "max_ulps": 32,
    }
    # reference defaults (src/sampler.rs:507-531,:630-634; src/adapt_strategy.rs:56-69; src/stepsize/adapt.rs:320-329;
    # src/stepsize/dual_avg.rs:22-31; src/stepsize/adam.rs:25-33; src/transform/adapt/diagonal.rs:99-106)
    k["default_settings"] = {
        "source": "/root/reference/src/sampler.rs:507-531,:630-634 and nested Default impls",
        "num_tune": 400, "num_draws": 1000, "maxdepth": 10, "mindepth": 0, "max_energy_error": 1000.0,
        "check_turning": 1, "extra_doublings": 0, "seed": 0, "num_chains": 6, "early_window": 0.3,
        "step_size_window": 0.15, "mass_matrix_switch_freq": 80, "early_mass_matrix_switch_freq": 10,
        "mass_matrix_update_freq": 1, "mass_matrix_window_growth": 1.5, "store_mass_matrix": 0,
        "use_grad_based_estimate": 1, "target_accept": 0.8, "initial_step": 0.1, "has_jitter": 1, "jitter": 0.1,
        "step_size_method": 0, "da_k": 0.75, "da_t0": 10.0, "da_gamma": 0.05, "da_max_step_size": 3.141592653589793,
        "adam_beta1": 0.9, "adam_beta2": 0.999, "adam_epsilon": 1e-8, "adam_learning_rate": 0.05,   # src/stepsize/adam.rs:25-33
    }
    return k


def oracle_k1_pins():
    from oracle import oracle as O
    s = O.default_settings(seed=0, num_chains=4)
    x0 = np.zeros((4, 10))
    pos, st, steps, failed = O.run(s, O.LOGP_IID_NORMAL, 10, [3.0], O.gpu_cfg(64), 4, x0, 1400)
    assert failed == 0
    picks = [0, 1, 2, 10, 119, 120, 339, 340, 399, 400, 401, 1000, 1399]
    return {
        "config": "K1: 10-dim iid N(3,1), 4 chains, DiagNutsSettings::default(), seed 0, x0 = zeros; oracle gpu_cfg(64)",
        "draws": picks,
        "positions_hex": [[[float(v).hex() for v in pos[t, c]] for c in range(4)] for t in picks],
        "step_size_hex": [[float(st["step_size"][t, c]).hex() for c in range(4)] for t in picks],
        "depth": [[int(st["depth"][t, c]) for c in range(4)] for t in picks],
        "n_steps": [[int(st["n_steps"][t, c]) for c in range(4)] for t in picks],
        "total_steps": int(steps),
    }


if __name__ == "__main__":
    with open(os.path.join(HERE, "reference_kats.json"), "w") as f:
        json.dump(reference_kats(), f, indent=1)
    with open(os.path.join(HERE, "oracle_k1_pins.json"), "w") as f:
        json.dump(oracle_k1_pins(), f, indent=1)
    print("wrote fixtures")
