"""Lane-group kernel (8 chains per wavefront) against the wave-per-chain kernel: same draws, bit for bit."""
import sys, time
import numpy as np
import torch  # noqa: F401  (before the engine library)
sys.path.insert(0, ".")
import nuts_rs_amd as N

sys.path.insert(0, "tests")
from helpers import assert_bit_exact  # noqa: E402


def run(settings, logp, n_chains, x0, n_tune, n_draws, lane_groups, split=(1,)):
    b = N.ChainBatch(settings, logp, n_chains, lane_groups=lane_groups)
    b.set_position(x0)
    b.draw_many(n_tune)
    out_p, out_s = [], []
    t0 = time.time()
    for part in split:
        p, s = b.draw_many(n_draws * part // sum(split))
        out_p.append(p); out_s.append(s)
    dt = time.time() - t0
    b.close()
    return np.concatenate(out_p), np.concatenate(out_s), dt


def main():
    rng = np.random.default_rng(3)
    cases = []
    for dim in (1, 2, 3, 7, 10, 15, 16, 17, 24, 31, 32, 33, 47, 63, 64):
        cases.append(("iid", N.LogpSpec.iid_normal(dim, 0.5), dim, {}))
    cases.append(("diag", N.LogpSpec.diag_normal(np.exp(rng.normal(size=10))), 10, {}))
    cases.append(("diag", N.LogpSpec.diag_normal(np.exp(rng.normal(size=29))), 29, {}))
    cases.append(("diag", N.LogpSpec.diag_normal(np.exp(rng.normal(size=50))), 50, {}))
    cases.append(("schools", N.LogpSpec.eight_schools(), 10, {}))
    cases.append(("iid-extra", N.LogpSpec.iid_normal(10, 0.0), 10, dict(extra_doublings=2, mindepth=2)))
    cases.append(("iid-nojitter", N.LogpSpec.iid_normal(10, 0.0), 10, dict(_jitter=None)))
    cases.append(("iid-adam", N.LogpSpec.iid_normal(10, 0.0), 10, dict(_method=N.STEP_ADAM)))
    cases.append(("iid-deep", N.LogpSpec.iid_normal(12, 0.0), 12, dict(maxdepth=4)))
    cases.append(("schools-div", N.LogpSpec.eight_schools(), 10, dict(max_energy_error=5.0)))
    for name, logp, dim, kw in cases:
        n_chains = 70
        kw = dict(kw)
        jit = kw.pop("_jitter", 0.1); method = kw.pop("_method", N.STEP_DUAL_AVERAGE)
        st = N.DiagNutsSettings(num_tune=120, seed=11, **kw)
        st.adapt_options.step_size_settings.jitter = jit
        st.adapt_options.step_size_settings.method = method
        x0 = rng.uniform(-1, 1, size=(n_chains, dim))
        pa, sa, _ = run(st, logp, n_chains, x0, 120, 60, 1)
        pb, sb, _ = run(st, logp, n_chains, x0, 120, 60, 2, split=(1, 2))
        try:
            assert_bit_exact(pb, sb, pa, sa)
            print(name, dim, "OK  depth mean", sa["depth"].mean(), "div", int(sa["diverging"].sum()))
        except AssertionError as e:
            print(name, dim, "MISMATCH", str(e)[:300])


if __name__ == "__main__":
    main()
