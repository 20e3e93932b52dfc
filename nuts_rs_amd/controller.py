"""The reference's `Sampler` control plane over the batched engine (reference src/sampler.rs:1229-1552).

The reference runs one `ChainProcess` per chain on a Rayon pool and steers them from a controller thread through
`pause / resume (Continue) / progress / inspect / flush / abort / wait_timeout`.  Here one controller thread owns one
`ChainBatch` (all chains of one GPU) and advances it `chunk_draws` draws per kernel launch; commands take effect
between launches.  Names, return shapes and the bookkeeping of `ChainProgress` (src/sampler.rs:1009-1051) are the
reference's; the trace is a dict of host arrays instead of a `TraceStorage`.
"""
import threading
import inspect
import time
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np

from .sampler import STATS_DTYPE, ChainBatch, DiagNutsSettings, LogpSpec


@dataclass
class ChainProgress:                      # src/sampler.rs:1009-1051
    finished_draws: int = 0
    total_draws: int = 0
    divergences: int = 0
    tuning: bool = True
    started: bool = False
    latest_num_steps: int = 0
    total_num_steps: int = 0
    step_size: float = 0.0
    runtime: float = 0.0                  # seconds (the reference keeps a Duration)
    divergent_draws: List[int] = field(default_factory=list)

    def update(self, diverging, tuning, num_steps, step_size, draw_duration):
        if diverging and not tuning:
            self.divergences += 1
            self.divergent_draws.append(self.finished_draws)
        self.finished_draws += 1
        self.tuning = bool(tuning)
        self.latest_num_steps = int(num_steps)
        self.total_num_steps += int(num_steps)
        self.step_size = float(step_size)
        self.runtime += draw_duration


@dataclass
class ProgressCallback:                   # src/sampler.rs:1262-1265
    callback: Callable[[float, List[ChainProgress]], None]
    rate: float                           # seconds between calls


class SamplerWaitResult:                  # src/sampler.rs:1247-1251
    """kind: 'trace' (finished: .trace), 'timeout' (.sampler is still running), 'err' (.error, .trace or None)"""

    def __init__(self, kind, trace=None, sampler=None, error=None):
        self.kind, self.trace, self.sampler, self.error = kind, trace, sampler, error

    def __repr__(self):
        return f"SamplerWaitResult({self.kind})"


class Sampler:
    """`Sampler::new(model, settings, trace_config, num_cores, callback)` for one GPU's batch of chains.

    engine_factory(settings, logp, n_chains, chain_id_offset, device) -> an object with ChainBatch's interface
    (the tests inject a stand-in to exercise the control logic without a GPU; the default is the HIP engine).
    """

    def __init__(self, settings: DiagNutsSettings, logp: LogpSpec, x0=None, callback: Optional[ProgressCallback] = None,
                 chunk_draws: int = 16, chain_id_offset: int = 0, device: int = -1, store_positions: bool = True,
                 engine_factory=None):
        self.settings = settings
        self._total = settings.num_tune + settings.num_draws
        self._n = settings.num_chains
        self._chunk = max(1, int(chunk_draws))
        self._store_positions = store_positions
        self._callback = callback
        self._lock = threading.Lock()              # guards _trace_pos/_trace_stats/_progress
        self._cmd = threading.Condition()
        self._paused = False
        self._abort = False
        self._done = threading.Event()
        self._error = None
        # ChainProgress of every chain, kept as arrays (thousands of chains per batch)
        self._finished, self._started, self._runtime = 0, False, 0.0
        self._divergences = np.zeros(self._n, dtype=np.int64)
        self._tuning = np.ones(self._n, dtype=bool)
        self._latest_steps = np.zeros(self._n, dtype=np.int64)
        self._total_steps = np.zeros(self._n, dtype=np.int64)
        self._step_size = np.zeros(self._n)
        self._divergent = []                        # (draw index, chain) of post-warm-up divergences
        self._pos = self._st = None                 # the trace, allocated once by the controller thread and filled in place
        factory = engine_factory or (lambda s, l, n, off, dev: ChainBatch(s, l, n, chain_id_offset=off, device=dev))
        self._thread = threading.Thread(target=self._main, name="nuts-amd-controller",
                                        args=(factory, logp, x0, chain_id_offset, device), daemon=True)
        self._thread.start()

    # ---- controller thread (the reference's main_loop + the ChainProcess bodies, src/sampler.rs:1120-1199, :1330-1460)
    def _main(self, factory, logp, x0, chain_id_offset, device):
        batch = None
        try:
            batch = factory(self.settings, logp, self._n, chain_id_offset, device)
            batch.init_with_retries(x0)              # ChainProcess init loop: up to 500 initial points per chain (sampler.rs:1133-1147)
            with self._lock:
                self._started = True
            start, pause_time, last_cb = time.monotonic(), 0.0, None
            self._fire_callback(0.0)
            last_cb = time.monotonic()
            finished = 0
            # decided ONCE from the signature: a TypeError raised inside a real draw_many (after the kernel has advanced the
            # chains) must surface, not trigger a second, trace-skipping call
            try:
                takes_out = "out" in inspect.signature(batch.draw_many).parameters
            except (TypeError, ValueError):
                takes_out = False
            while finished < self._total:
                with self._cmd:
                    if self._paused and not self._abort:
                        p0 = time.monotonic()
                        while self._paused and not self._abort:
                            self._cmd.wait(timeout=0.05)
                        pause_time += time.monotonic() - p0
                    if self._abort:
                        break
                n = min(self._chunk, self._total - finished)
                if self._st is None:                # one trace for the whole run (no per-chunk arrays to concatenate at the end)
                    self._st = np.zeros((self._total, self._n), dtype=STATS_DTYPE)
                    if self._store_positions:
                        self._pos = np.empty((self._total, self._n, logp.dim))
                dst = (self._pos[finished:finished + n] if self._store_positions else None, self._st[finished:finished + n])
                t0 = time.monotonic()
                if takes_out:
                    pos, st = batch.draw_many(n, positions=self._store_positions, out=dst)
                else:                               # an engine stand-in without `out=`
                    pos, st = batch.draw_many(n, positions=self._store_positions)
                    if pos is not None:
                        dst[0][...] = pos
                    dst[1][...] = st
                per_draw = (time.monotonic() - t0) / n
                with self._lock:
                    # ChainProgress::update for every (draw, chain) of the chunk (src/sampler.rs:1038-1050)
                    div = (st["diverging"] != 0) & (st["tuning"] == 0)
                    self._divergences += div.sum(axis=0)
                    self._divergent += [(finished + int(t), int(c)) for t, c in np.argwhere(div)]
                    self._tuning = st["tuning"][-1] != 0
                    self._latest_steps = st["n_steps"][-1].astype(np.int64)
                    self._total_steps += st["n_steps"].sum(axis=0).astype(np.int64)
                    self._step_size = st["step_size"][-1].copy()
                    self._runtime += per_draw * n
                    self._finished = finished + n
                finished += n
                if self._callback is not None and time.monotonic() - last_cb >= self._callback.rate:
                    self._fire_callback(time.monotonic() - start - pause_time)
                    last_cb = time.monotonic()
            self._fire_callback(time.monotonic() - start - pause_time)
        except BaseException as e:   # noqa: BLE001 — reported through wait_timeout / abort like the reference's anyhow::Error
            self._error = e
        finally:
            if batch is not None:
                try:
                    batch.close()
                except Exception:   # noqa: BLE001
                    pass
            self._done.set()

    def _fire_callback(self, elapsed):
        if self._callback is not None:
            self._callback.callback(elapsed, self.progress())

    def _snapshot(self, final=False):
        """the draws finished so far: a copy while the sampler runs, the trace itself (trimmed) once its thread has ended"""
        with self._lock:
            f = self._finished
            pos = None if self._pos is None else (self._pos[:f] if final else self._pos[:f].copy())
            st = None if self._st is None else (self._st[:f] if final else self._st[:f].copy())
        return {"positions": pos, "stats": st}

    # ---- commands (src/sampler.rs:1463-1551)
    def pause(self):
        with self._cmd:
            self._paused = True
            self._cmd.notify_all()

    def resume(self):
        with self._cmd:
            self._paused = False
            self._cmd.notify_all()

    def flush(self):
        """Nothing is buffered outside host memory; kept for interface parity (`Sampler::flush`)."""

    def progress(self) -> List[ChainProgress]:
        with self._lock:
            by_chain = {}
            for t, c in self._divergent:
                by_chain.setdefault(c, []).append(t)
            return [ChainProgress(self._finished, self._total, int(self._divergences[c]), bool(self._tuning[c]),
                                  self._started, int(self._latest_steps[c]), int(self._total_steps[c]),
                                  float(self._step_size[c]), self._runtime, by_chain.get(c, []))
                    for c in range(self._n)]

    def inspect(self):
        """(error or None, trace so far) without stopping the sampler (`Sampler::inspect`)."""
        return self._error, self._snapshot()

    def abort(self):
        """Stop after the launch in flight and return (error or None, trace so far) (`Sampler::abort`)."""
        with self._cmd:
            self._abort = True
            self._cmd.notify_all()
        self._thread.join()
        return self._error, self._snapshot(final=True)

    def is_finished(self):
        return self._done.is_set()

    def wait_timeout(self, timeout: float) -> SamplerWaitResult:
        """`Sampler::wait_timeout`: the trace when sampling finished within `timeout` seconds, else the sampler back."""
        if not self._done.wait(timeout):
            return SamplerWaitResult("timeout", sampler=self)
        self._thread.join()
        if self._error is not None:
            return SamplerWaitResult("err", error=self._error, trace=self._snapshot(final=True))
        return SamplerWaitResult("trace", trace=self._snapshot(final=True))
