"""N > 1 path on CPU: world_size 2, gloo backend.  Covers what a multi-GPU run adds on top of the single-GPU engine:
the chain partition (ids, init positions, RNG keys invariant to the partition), the max/sum reductions bench.py does
on its timing scalars, and the opt-in pooled-Welford merge (not reference behaviour; see nuts_rs_amd/pooled.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import ctypes as C
    import torch
    import torch.distributed as dist
    import nuts_rs_amd as N
    from nuts_rs_amd import pooled
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total, dim, seed = 11, 6, 99
    off, n_local = pooled.shard_chains(total, world, rank)
    L = N.load_library()
    x0 = np.empty((n_local, dim))
    assert L.nm_init_positions_uniform(seed, off, n_local, dim, x0.ctypes.data) == 0
    keys = []
    for c in range(n_local):
        k = (C.c_uint8 * 32)()
        L.nm_chain_rng_key(seed, off + c, k)
        keys.append(bytes(k))
    # bench.py's reductions: MAX over ranks of the elapsed time, SUM of the step counts
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    steps = torch.tensor([float(1000 * (rank + 1))], dtype=torch.float64)
    dist.all_reduce(steps, op=dist.ReduceOp.SUM)
    # pooled Welford over the ranks' draws
    rng = np.random.default_rng(5)
    draws = rng.normal(size=(40, dim)) * np.arange(1, dim + 1)
    lo, hi = (0, 17) if rank == 0 else (17, 40)
    merged = pooled.pooled_welford(pooled.welford_partial(draws[lo:hi]), dist)
    q.put((rank, off, n_local, x0, keys, float(t.item()), float(steps.item()), merged))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_partition_and_reductions(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, off0, n0, x0a, k0, t0, s0, m0), (r1, off1, n1, x0b, k1, t1, s1, m1) = res
    assert (off0, n0, off1, n1) == (0, 6, 6, 5)                       # contiguous blocks, remainder to rank 0
    # the partition is invisible: positions / keys equal those of a single process that owns all 11 chains
    full = oracle.init_positions_uniform(99, 0, 11, 6)
    assert (np.concatenate([x0a, x0b]) == full).all()
    assert k0 + k1 == [oracle.chain_key(99, c) for c in range(11)]
    assert t0 == t1 == pytest.approx(0.2) and s0 == s1 == 3000.0
    rng = np.random.default_rng(5)
    draws = rng.normal(size=(40, 6)) * np.arange(1, 7)
    for m in (m0, m1):
        assert m[0] == 40
        assert np.allclose(m[1], draws.mean(axis=0), rtol=1e-13, atol=1e-13)
        assert np.allclose(m[2], ((draws - draws.mean(axis=0)) ** 2).sum(axis=0), rtol=1e-12)


def test_shard_chains_covers_everything():
    from nuts_rs_amd import pooled
    for total in (1, 7, 8, 4096, 65536, 65537):
        for world in (1, 2, 3, 8):
            blocks = [pooled.shard_chains(total, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and sum(n for _, n in blocks) == total
            for (o, n), (o2, _) in zip(blocks, blocks[1:]):
                assert o + n == o2
