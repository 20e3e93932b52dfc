"""Sharding helpers for one-process-per-GPU runs (torch.distributed; RCCL on GPUs, gloo in the CPU tests).

Parity mode needs no collective: chains are independent (reference src/sampler.rs:1105-1126) and the chain RNG
depends only on (seed, global chain id), so a rank just owns a contiguous block of chain ids.

OPT-IN pooled adaptation (north_star's "RCCL cross-chain Welford reduction").  It is NOT reference
behaviour — every reference chain adapts alone (src/adapt_strategy.rs:24-39) — and it changes results, so it is a
separate driver, `pooled_warmup`, never the default.  All chains of all ranks share ONE diagonal transformation:
the engine runs with the transformation frozen (its own mass-matrix adaptation off, step size adapting per chain as
usual), records draws and gradients of a window in device buffers, each rank reduces its chains' window to
(count, mean[D], M2[D]) for draws and for gradients with the engine's reduction kernel (nm_pooled_partials: only healthy
chains' `is_good` draws count, like the reference's collector), the ranks exchange those 2 (2 D + 1) doubles with
ONE all_gather per window (torch.distributed: RCCL over xGMI with the nccl backend, issued on a side stream so the next
window's kernel is already running; gloo in the tests), merge them in rank order with Chan's formula, and every rank
sets sigma = (var_x / var_g)^(1/4), mean = x_bar + sigma^2 g_bar (the reference's diagonal estimate,
src/transform/diagonal.rs:107-131) for all its chains with one broadcast upload (nm_engine_set_transform).
`pooled_welford` is the merge with numpy arrays (pinned by tests/test_distributed_cpu.py).
"""
import numpy as np


def shard_chains(n_chains_total, world_size, rank):
    """Contiguous block partition: returns (chain_id_offset, n_local).  Remainder chains go to the low ranks."""
    base, rem = divmod(n_chains_total, world_size)
    n_local = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, n_local


def welford_partial(x):
    """(count, mean, M2) of the rows of x ([n, D])."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    mean = x.mean(axis=0) if n else np.zeros(x.shape[1])
    m2 = ((x - mean) ** 2).sum(axis=0) if n else np.zeros(x.shape[1])
    return float(n), mean, m2


def chan_merge(a, b):
    """Chan et al. pairwise merge of two (count, mean, M2) triples."""
    na, ma, sa = a
    nb, mb, sb = b
    n = na + nb
    if n == 0:
        return 0.0, ma, sa
    d = mb - ma
    return n, ma + d * (nb / n), sa + sb + d * d * (na * nb / n)


def pooled_welford(local, dist=None):
    """all_gather the per-rank triples (payload 2D+1 doubles per rank) and merge them in rank order."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    import torch
    n, mean, m2 = local
    payload = torch.from_numpy(np.concatenate([[n], mean, m2]))
    gathered = [torch.empty_like(payload) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, payload)
    d = len(mean)
    out = None
    for g in gathered:
        g = g.numpy()
        t = (float(g[0]), g[1:1 + d].copy(), g[1 + d:].copy())
        out = t if out is None else chan_merge(out, t)
    return out


def _partials_device(pos, grad, stats, stream):
    """[2][1 + 2 D] = (count, mean[D], M2[D]) of the window's draws and gradients, by the engine's own reduction kernel
    (nm_pooled_partials, csrc/pooled_reduce.hip) on `stream`: rows of stopped chains and draws the reference's collector
    would not keep (not `is_good`) are left out."""
    import torch
    from . import _lib
    w, nc, dim = pos.shape
    out = torch.empty((2, 1 + 2 * dim), dtype=torch.float64, device=pos.device)
    _lib.check_status(_lib.load().nm_pooled_partials(w * nc, dim, pos.data_ptr(), grad.data_ptr(), stats.data_ptr(), out.data_ptr(),
                                                     stream.cuda_stream), _lib.load().nm_pooled_last_error)
    return out


def _merge_payloads(payloads, d):
    """Chan merge of gathered [count, mean[D], M2[D]] payloads in rank order (torch tensors on one device)."""
    n, mean, m2 = payloads[0][0], payloads[0][1:1 + d], payloads[0][1 + d:]
    for p in payloads[1:]:
        nb, mb, sb = p[0], p[1:1 + d], p[1 + d:]
        if float(nb) == 0.0:            # a rank whose chains kept no draw in this window (all stopped / diverging): nothing to merge,
            continue                    # and 0 / 0 below would poison the mean (the device merge skips empty partials too)
        if float(n) == 0.0:
            n, mean, m2 = nb, mb, sb
            continue
        tot = n + nb
        delta = mb - mean
        mean = mean + delta * (nb / tot)
        m2 = m2 + sb + delta * delta * (n * nb / tot)
        n = tot
    return n, mean, m2


def pooled_warmup(batch, num_tune, dist=None, windows=None, collective_device=None, on_window=None):
    """Warm up `batch` (a ChainBatch created from LowRankNutsSettings(freeze_transform=True, num_tune=num_tune)) with ONE
    diagonal transformation pooled over all chains of all ranks.  Returns the list of (draw index, sigma, mean) updates.

    windows: draw counts of the successive windows (default: the reference's schedule shape — 10-draw windows in the
    first 30 %, then 80 growing by 1.5x, none in the last 15 % where only the step size adapts).
    dist: an initialised torch.distributed module or None (single process).  With the nccl backend the payload stays on
    the GPU (RCCL); with gloo (tests) it goes through the host (`collective_device="cpu"`)."""
    import torch
    dim, nc = batch.logp.dim, batch.n_chains
    if windows is None:
        early_end, final = int(0.3 * num_tune), num_tune - int(0.15 * num_tune)
        windows, t, w = [], 0, 80
        while t + 10 <= early_end:
            windows.append(10); t += 10
        while t + w <= final:
            windows.append(w); t += w; w = max(w + 1, int(round(w * 1.5)))
    dev = torch.device("cuda", torch.cuda.current_device())
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    cdev = torch.device(collective_device) if collective_device else dev
    side = torch.cuda.Stream(device=dev)
    updates, done, pending = [], 0, None

    def finish(p):
        work, gathered, at = p
        if work is not None:
            work.wait()
        with torch.cuda.stream(side):
            # the merge of the gathered partials in rank order + the pooled transformation: the engine's own kernel (nm_pooled_finish,
            # csrc/pooled_reduce.hip — what a host without Python calls after nm_pooled_exchange)
            allp = torch.stack([g.to(dev) for g in gathered]).contiguous()
            sigma = torch.empty(dim, dtype=torch.float64, device=dev)
            mean = torch.empty(dim, dtype=torch.float64, device=dev)
            cnt = torch.zeros(1, dtype=torch.float64, device=dev)
            from . import _lib
            _lib.check_status(_lib.load().nm_pooled_finish(len(gathered), dim, allp.data_ptr(), sigma.data_ptr(), mean.data_ptr(), cnt.data_ptr(),
                                                           side.cuda_stream), _lib.load().nm_pooled_last_error)
        side.synchronize()
        if float(cnt.item()) < 3.0:     # fewer than three pooled draws: no estimate (a RunningVariance asserts count > 1,
            return                      # adapt/diagonal.rs:48; with 2 the ratio of variances is noise) — keep the current transformation
        s_h, m_h = sigma.cpu().numpy(), mean.cpu().numpy()
        batch.set_transform(s_h, m_h, np.zeros(0), np.zeros((0, dim)), np.zeros(dim))
        updates.append((at, s_h, m_h))
        if on_window:
            on_window(at, s_h, m_h)

    from .sampler import STATS_DTYPE
    for w in windows:
        pos = torch.empty((w, nc, dim), dtype=torch.float64, device=dev)
        grad = torch.empty((w, nc, dim), dtype=torch.float64, device=dev)
        # zero-filled statistics: a chain that has stopped writes no rows, and an all-zero row is not a draw the collector keeps
        stats = torch.zeros((w, nc, STATS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        torch.cuda.current_stream(dev).synchronize()       # (the fill runs on torch's stream, the engine on its own)
        batch.draw_device_ex(w, positions=pos.data_ptr(), stats=stats.data_ptr(), gradient=grad.data_ptr())     # the window's kernel
        if pending is not None:                # the previous window's exchange ran beside this kernel; apply it now:
            finish(pending)                    # the pooled transformation lags the draws by ONE window (documented, deliberate)
        done += w
        with torch.cuda.stream(side):          # this window's partials + the collective, off the engine's stream
            for t in (pos, grad, stats):
                t.record_stream(side)
            payload = _partials_device(pos, grad, stats, side).reshape(-1).to(cdev)
            gathered = [torch.empty_like(payload) for _ in range(world)]
            if world > 1:
                work = dist.all_gather(gathered, payload, async_op=True)
            else:
                gathered, work = [payload], None
        pending = (work, gathered, done)
    if pending is not None:
        finish(pending)
    if done < num_tune:
        batch.draw_device(num_tune - done)     # the final window: step size only (reference adapt_strategy.rs:215-221)
    return updates
