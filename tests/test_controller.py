"""The `Sampler` control plane (reference src/sampler.rs:1229-1552) over the batched engine.

CPU part: the control logic against a stand-in engine with ChainBatch's interface (no compute).  GPU part: the same
commands over the HIP engine, and the trace against a plain `draw_many`."""
import time

import numpy as np
import pytest

import nuts_rs_amd as N
from nuts_rs_amd._lib import STATS_DTYPE


class FakeBatch:
    """ChainBatch's interface; every draw takes `delay` seconds; chain c diverges on draw 7 + c (post-warm-up)."""

    def __init__(self, settings, logp, n_chains, chain_id_offset, device, delay=0.002, fail_at=None):
        self.s, self.n, self.dim, self.t, self.delay, self.fail_at = settings, n_chains, logp.dim, 0, delay, fail_at
        self.closed = False

    def init_positions_uniform(self):
        return np.zeros((self.n, self.dim))

    def init_with_retries(self, x0=None, max_tries=500):
        self.set_position(self.init_positions_uniform() if x0 is None else x0)
        return None, None

    def set_position(self, x0):
        assert x0.shape == (self.n, self.dim)

    def draw_many(self, n, positions=True, stats=True):
        pos = np.zeros((n, self.n, self.dim)) if positions else None
        st = np.zeros((n, self.n), dtype=STATS_DTYPE)
        for i in range(n):
            time.sleep(self.delay)
            if self.fail_at is not None and self.t == self.fail_at:
                raise RuntimeError("logp failure in chain 1")
            st["draw"][i] = self.t
            st["chain"][i] = np.arange(self.n)
            st["tuning"][i] = self.t < self.s.num_tune
            st["n_steps"][i] = 3 + np.arange(self.n)
            st["step_size"][i] = 0.5
            st["diverging"][i] = (7 + np.arange(self.n)) == self.t
            if pos is not None:
                pos[i] = self.t
            self.t += 1
        return pos, st

    def close(self):
        self.closed = True


def make(settings, **kw):
    made = []

    def factory(s, logp, n, off, dev):
        made.append(FakeBatch(s, logp, n, off, dev, **kw))
        return made[-1]
    return factory, made


def test_runs_to_completion_and_reports_progress():
    s = N.DiagNutsSettings(num_tune=5, num_draws=20, num_chains=3)
    factory, made = make(s)
    calls = []
    smp = N.Sampler(s, N.LogpSpec.iid_normal(4), chunk_draws=4, engine_factory=factory,
                    callback=N.ProgressCallback(lambda el, pr: calls.append((el, pr)), rate=0.01))
    res = smp.wait_timeout(10.0)
    assert res.kind == "trace" and made[0].closed
    pos, st = res.trace["positions"], res.trace["stats"]
    assert pos.shape == (25, 3, 4) and (st["draw"][:, 0] == np.arange(25)).all()
    pr = smp.progress()
    assert [p.finished_draws for p in pr] == [25] * 3 and all(p.total_draws == 25 and p.started and not p.tuning for p in pr)
    # ChainProgress::update (src/sampler.rs:1038-1050): divergences counted only after warm-up, with their draw index
    assert [p.divergences for p in pr] == [1, 1, 1] and [p.divergent_draws for p in pr] == [[7], [8], [9]]
    assert [p.latest_num_steps for p in pr] == [3, 4, 5] and [p.total_num_steps for p in pr] == [75, 100, 125]
    assert len(calls) >= 3 and calls[-1][1][0].finished_draws == 25                # first, periodic and final callback
    assert all(b[0] >= a[0] for a, b in zip(calls, calls[1:]))                     # elapsed is monotone


def test_pause_resume_abort_and_timeout():
    s = N.DiagNutsSettings(num_tune=10, num_draws=2000, num_chains=2)
    factory, made = make(s, delay=0.001)
    smp = N.Sampler(s, N.LogpSpec.iid_normal(2), chunk_draws=5, engine_factory=factory)
    res = smp.wait_timeout(0.05)
    assert res.kind == "timeout" and res.sampler is smp                            # still running: the sampler comes back
    smp.pause()
    time.sleep(0.05)
    a = smp.progress()[0].finished_draws
    time.sleep(0.1)
    assert smp.progress()[0].finished_draws == a and 0 < a < 2010                   # paused between launches
    err, partial = smp.inspect()
    assert err is None and partial["stats"].shape[0] == a
    smp.resume()
    time.sleep(0.1)
    assert smp.progress()[0].finished_draws > a
    err, trace = smp.abort()
    assert err is None and made[0].closed
    n = trace["stats"].shape[0]
    assert a < n < 2010 and n % 5 == 0 and (trace["positions"][:, 0, 0] == np.arange(n)).all()
    assert smp.is_finished()


def test_engine_error_is_reported_with_the_partial_trace():
    s = N.DiagNutsSettings(num_tune=4, num_draws=50, num_chains=2)
    factory, made = make(s, fail_at=17)
    res = N.Sampler(s, N.LogpSpec.iid_normal(2), chunk_draws=4, engine_factory=factory).wait_timeout(10.0)
    assert res.kind == "err" and "logp failure" in str(res.error)
    assert res.trace["stats"].shape[0] == 16 and made[0].closed                    # the completed launches are kept


@pytest.mark.gpu
def test_sampler_over_the_hip_engine_matches_draw_many():
    s = N.DiagNutsSettings(num_tune=60, num_draws=40, num_chains=6, seed=5)
    logp = N.LogpSpec.iid_normal(33, 3.0)
    smp = N.Sampler(s, logp, chunk_draws=7)
    smp.pause()
    time.sleep(0.05)
    smp.resume()
    res = smp.wait_timeout(120.0)
    assert res.kind == "trace"
    b = N.ChainBatch(s, logp, 6)
    b.set_position(b.init_positions_uniform())
    pos, st = b.draw_many(100)
    b.close()
    assert (res.trace["positions"].view(np.uint64) == pos.view(np.uint64)).all()   # chunking / pausing is invisible
    assert (res.trace["stats"]["n_steps"] == st["n_steps"]).all()
    pr = smp.progress()
    assert all(p.finished_draws == 100 and not p.tuning for p in pr)
    assert [p.total_num_steps for p in pr] == st["n_steps"].sum(axis=0).tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("make", [lambda: N.DiagMclmcSettings(num_tune=50, num_draws=30, num_chains=5, seed=6),
                                  lambda: N.DiagNutsSettings(num_tune=50, num_draws=30, num_chains=5, seed=6,
                                                             trajectory_kind=N.KineticEnergyKind.MICROCANONICAL)],
                         ids=["mclmc", "microcanonical_nuts"])
def test_sampler_and_sample_helper_with_the_other_integrators(make):
    """`Sampler` and `sample()` drive the MCLMC sampler and the non-Euclidean trajectory kinds like any other settings."""
    s = make()
    logp = N.LogpSpec.iid_normal(21, 3.0)
    res = N.Sampler(s, logp, chunk_draws=9).wait_timeout(120.0)
    assert res.kind == "trace"
    b = N.ChainBatch(s, logp, 5)
    b.set_position(b.init_positions_uniform())
    pos, st = b.draw_many(80)
    b.close()
    assert (res.trace["positions"].view(np.uint64) == pos.view(np.uint64)).all()
    assert (res.trace["stats"]["n_steps"] == st["n_steps"]).all()
    pos_s, st_s = N.sample(s, logp)          # (init_with_retries: attempt 0 is the uniform init_position, like set_position above)
    assert (pos_s.view(np.uint64) == pos.view(np.uint64)).all() and (st_s["n_steps"] == st["n_steps"]).all()
