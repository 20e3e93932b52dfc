"""N > 1 on the GPU: two ranks sharing GPU 0 (gloo), launched like the driver launches bench.py.
  * parity mode: the concatenated draws of the two ranks ARE the draws of one engine that owns all the chains, bit for bit
    (chains are independent units; the RNG depends only on (seed, global chain id); no data-path collective)
  * the opt-in pooled adaptation: both ranks end up with the same pooled transformation, equal to the one a single
    process computes from the same two partial reductions
  * bench.py --gpus 2 starts its own ranks and reports the whole job"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import nuts_rs_amd as N

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(mode, outdir, nproc=2, backend="gloo"):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), mode, str(outdir)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NM_TEST_BACKEND=backend)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]


def test_two_ranks_equal_one_engine(tmp_path):
    launch("shard", tmp_path)
    a, b = (np.load(tmp_path / f"shard_{r}.npz") for r in (0, 1))
    dim, C_, tune, draws, seed = 24, 20, 60, 30, 77
    s = N.DiagNutsSettings(num_chains=2 * C_, seed=seed, num_tune=tune)
    e = N.ChainBatch(s, N.LogpSpec.diag_normal(np.exp(np.linspace(-2, 2, dim))), 2 * C_)
    e.set_position(e.init_positions_uniform())
    pos, st = e.draw_many(tune + draws)
    e.close()
    both = np.concatenate([a["pos"], b["pos"]], axis=1)
    assert (both.view(np.uint64) == pos.view(np.uint64)).all()
    assert (np.concatenate([a["n_steps"], b["n_steps"]], axis=1) == st["n_steps"]).all()
    assert (np.concatenate([a["chain"], b["chain"]], axis=1) == st["chain"]).all()
    assert (np.concatenate([a["step"], b["step"]], axis=1) == st["step_size"]).all()


def test_pooled_adaptation_two_ranks(tmp_path):
    launch("pooled", tmp_path)
    a, b = (np.load(tmp_path / f"pooled_{r}.npz") for r in (0, 1))
    assert a["sigma"].shape == (3, 24)
    assert (a["sigma"] == b["sigma"]).all() and (a["mean"] == b["mean"]).all()       # one transformation for the whole job
    # the target is N(0, diag(1/p)): the pooled sigma approaches its standard deviations, the pooled mean its centre
    true_sd = np.exp(np.linspace(-2, 2, 24)) ** -0.5
    assert np.abs(np.log(a["sigma"][-1] / true_sd)).max() < 0.25
    assert np.abs(a["mean"][-1] / true_sd).max() < 0.5
    sample = np.concatenate([a["pos"], b["pos"]], axis=1).reshape(-1, 24)
    assert np.abs(sample.std(axis=0) / true_sd - 1).max() < 0.15


def test_bench_starts_its_own_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "6", "--warmup", "2",
                        "--repeats", "2", "--chains", "64", "--dim", "128", "--num-tune", "30", "--pmc", "off", "--master-port", str(free_port())],
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["chains_per_gpu"] == 64 and d["value"] > 0
    assert abs(d["leapfrogs_per_draw"] * 6 * 128 * 128 / (d["ms_per_step"] * 6e-3) / d["value"] - 1) < 1e-6   # whole-job aggregate
    # asking for more ranks than GPUs over RCCL fails loudly instead of running fewer
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "2", "--pmc", "off"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and b"GPU(s) visible" in r.stderr


def _visible_gpus():
    import torch
    return torch.cuda.device_count()


def test_rccl_pooled_adaptation_and_bench():
    """The N > 1 path over RCCL itself (torch.distributed backend "nccl"), not gloo: the pooled adaptation's all_gather with its
    payload on the device, and bench.py's barrier / reductions.  With two or more GPUs visible: two ranks, one per GPU; on a
    one-GPU box: the same code with world size 1 (RCCL initialises, the collectives are its single-rank paths) — so that an
    8-GPU node needs no code the suite has not executed."""
    import tempfile
    n = 2 if _visible_gpus() >= 2 else 1
    with tempfile.TemporaryDirectory() as d:
        launch("pooled", d, nproc=n, backend="nccl")
        runs = [np.load(os.path.join(d, f"pooled_{r}.npz")) for r in range(n)]
    true_sd = np.exp(np.linspace(-2, 2, 24)) ** -0.5
    for r in runs[1:]:
        assert (r["sigma"] == runs[0]["sigma"]).all() and (r["mean"] == runs[0]["mean"]).all()
    assert np.abs(np.log(runs[0]["sigma"][-1] / true_sd)).max() < (0.25 if n == 2 else 0.4)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dist-backend", "nccl", "--steps", "6", "--warmup", "2",
                        "--repeats", "2", "--chains", "64", "--dim", "128", "--num-tune", "30", "--pmc", "off", "--no-cpu-baseline",
                        "--dist-always", "--master-port", str(free_port())],
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    d = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == n and d["value"] > 0


def test_pooled_partials_kernel_matches_numpy():
    """nm_pooled_partials (csrc/pooled_reduce.hip): count / mean / M2 of the rows the reference's collector keeps."""
    import ctypes as C
    import torch
    from nuts_rs_amd import _lib, pooled
    rng = np.random.default_rng(5)
    w, nc, dim = 7, 300, 37
    x, g = rng.normal(size=(w, nc, dim)), rng.normal(size=(w, nc, dim)) * 3 + 1
    st = np.zeros((w, nc), dtype=N.STATS_DTYPE)
    st["index_in_trajectory"] = rng.integers(-6, 7, size=(w, nc))
    st["diverging"] = rng.random((w, nc)) < 0.2
    st["chain_status"][:, ::17] = 6                                   # stopped chains
    good = (st["chain_status"] == 0) & np.where(st["diverging"] != 0, np.abs(st["index_in_trajectory"]) > 4, st["index_in_trajectory"] != 0)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8)).cuda()
    tx, tg, ts = dev(x), dev(g), dev(st)
    torch.cuda.synchronize()
    out = pooled._partials_device(tx.view(torch.float64).reshape(w, nc, dim), tg.view(torch.float64).reshape(w, nc, dim), ts,
                                  torch.cuda.current_stream()).cpu().numpy()
    for k, a in enumerate((x, g)):
        rows = a[good]
        n_, mean_, m2_ = pooled.welford_partial(rows)
        assert out[k, 0] == n_ == good.sum()
        assert np.allclose(out[k, 1:1 + dim], mean_, rtol=1e-13, atol=1e-13) and np.allclose(out[k, 1 + dim:], m2_, rtol=1e-12)


def test_pooled_exchange_and_finish_through_the_c_abi():
    """nm_pooled_exchange (world 1: the copy path; with a communicator: ncclAllGather, librccl resolved at run time) + nm_pooled_finish:
    the rank-order Chan merge of the gathered partials and the pooled diagonal transformation, against pooled.py's own merge — what
    a host without Python (the Rust side behind the C ABI) calls between nm_pooled_partials and nm_engine_set_transform."""
    import torch
    from nuts_rs_amd import _lib, pooled
    L = _lib.load()
    rng = np.random.default_rng(9)
    dim, world = 23, 5
    payloads = []
    for r in range(world):
        n = 0 if r == 2 else int(rng.integers(3, 40))                 # one rank kept no draw in this window
        rows_x, rows_g = rng.normal(size=(n, dim)) * 2 + 1, rng.normal(size=(n, dim)) * 0.5 - 1
        px, pg = (pooled.welford_partial(a) if n else (0.0, np.zeros(dim), np.zeros(dim)) for a in (rows_x, rows_g))
        payloads.append(np.concatenate([[px[0]], px[1], px[2], [pg[0]], pg[1], pg[2]]))
    gathered = torch.tensor(np.stack(payloads), dtype=torch.float64, device="cuda")
    sigma, mean, cnt = (torch.empty(k, dtype=torch.float64, device="cuda") for k in (dim, dim, 1))
    stream = torch.cuda.current_stream().cuda_stream
    # exchange with one rank = copy of the payload
    one = torch.empty(2 * (1 + 2 * dim), dtype=torch.float64, device="cuda")
    _lib.check_status(L.nm_pooled_exchange(None, 1, dim, gathered[0].data_ptr(), one.data_ptr(), stream), L.nm_pooled_last_error)
    _lib.check_status(L.nm_pooled_finish(world, dim, gathered.data_ptr(), sigma.data_ptr(), mean.data_ptr(), cnt.data_ptr(), stream), L.nm_pooled_last_error)
    torch.cuda.synchronize()
    assert (one.cpu().numpy() == payloads[0]).all()
    d2 = 2 * dim + 1
    n_, mx, vx = pooled._merge_payloads([torch.tensor(p[:d2]) for p in payloads], dim)
    _, mg, vg = pooled._merge_payloads([torch.tensor(p[d2:]) for p in payloads], dim)
    want_sigma = np.clip(np.sqrt(np.sqrt(vx.numpy() / vg.numpy())), 1e-10, 1e10)
    assert float(cnt.item()) == float(n_)
    assert np.allclose(sigma.cpu().numpy(), want_sigma, rtol=1e-14)
    assert np.allclose(mean.cpu().numpy(), mx.numpy() + want_sigma ** 2 * mg.numpy(), rtol=1e-13, atol=1e-13)


def test_pooled_exchange_over_a_real_rccl_communicator():
    """nm_pooled_exchange (csrc/pooled_reduce.hip: dlopen librccl, ncclAllGather on the CALLER's communicator) with a
    real two-rank ncclComm_t made by ncclCommInitRank — two processes on GPU 0.  RCCL may refuse two ranks on one device
    ("duplicate GPU"): then the test is skipped with RCCL's own reason on record, and the first multi-GPU box runs it for real."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
        ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_two_rank_worker.py"), str(r), d], env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
        outs = []
        for p in ps:
            try:
                outs.append(p.communicate(timeout=240))
            except subprocess.TimeoutExpired:
                for q in ps:
                    q.kill()
                pytest.skip("RCCL with two ranks on one GPU did not return within 240 s (its rendezvous needs one device per rank)")
        res = []
        for r in range(2):
            f = os.path.join(d, f"result_{r}.json")
            assert os.path.exists(f), (outs[r][0].decode(errors="replace")[-1500:] + outs[r][1].decode(errors="replace")[-1500:])
            res.append(json.load(open(f)))
    skips = [x["skip"] for x in res if "skip" in x]
    if skips:
        pytest.skip("two-rank RCCL communicator on one GPU: " + skips[0])
    assert all(x.get("ok") for x in res), res
