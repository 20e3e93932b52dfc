"""Sharding helpers for one-process-per-GPU runs (torch.distributed; RCCL on GPUs, gloo in the CPU tests).

Parity mode needs no collective: chains are independent (reference src/sampler.rs:1105-1126) and the chain RNG
depends only on (seed, global chain id), so a rank just owns a contiguous block of chain ids.

`pooled_welford` is the merge an OPT-IN pooled-adaptation mode would use (north_star's "cross-chain Welford
reduction").  It is NOT reference behaviour (every reference chain adapts alone, src/adapt_strategy.rs:24-39) and
the engine does not use it; it is here with its test so the payload and the math are pinned: per rank
(count, mean[D], M2[D]) gathered with all_gather and merged with Chan's parallel formula.
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
