"""Worker of tests/test_gpu_distributed.py: one rank of a torch.distributed job whose ranks share GPU 0 (gloo for the
collectives — a test rig; on a real node the backend is nccl = RCCL, one rank per GPU).  Modes:
  shard  : rank r draws chains [r C, (r+1) C) of a 2C-chain job (parity mode, no data-path collective)
  pooled : the opt-in pooled adaptation (nuts_rs_amd/pooled.py) over both ranks' chains
Results go to an .npz per rank in the directory given as argv[2]."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode, outdir = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    import nuts_rs_amd as N
    from nuts_rs_amd import pooled
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("NM_TEST_BACKEND", "gloo")
    if backend == "nccl":            # RCCL: one rank per GPU (LOCAL_RANK), the collective's payload stays on the device
        dev_index = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(dev_index)
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    else:
        dev_index = 0
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
    dim, C_, tune, draws, seed = 24, 20, 60, 30, 77
    off, n_local = pooled.shard_chains(world * C_, world, rank)
    prec = np.exp(np.linspace(-2, 2, dim))
    if mode == "shard":
        s = N.DiagNutsSettings(num_chains=world * C_, seed=seed, num_tune=tune)
        b = N.ChainBatch(s, N.LogpSpec.diag_normal(prec), n_local, chain_id_offset=off, device=dev_index)
        b.set_position(b.init_positions_uniform())
        pos, st = b.draw_many(tune + draws)
        np.savez(os.path.join(outdir, f"shard_{rank}.npz"), pos=pos, n_steps=st["n_steps"], chain=st["chain"], step=st["step_size"])
    else:
        s = N.LowRankNutsSettings(num_chains=world * C_, seed=seed, num_tune=tune, freeze_transform=True)
        b = N.ChainBatch(s, N.LogpSpec.diag_normal(prec), n_local, chain_id_offset=off, device=dev_index)
        b.set_position(b.init_positions_uniform())
        ups = pooled.pooled_warmup(b, tune, dist, windows=[10, 10, 20], collective_device=None if backend == "nccl" else "cpu")
        pos, st = b.draw_many(draws)
        np.savez(os.path.join(outdir, f"pooled_{rank}.npz"), pos=pos, n_steps=st["n_steps"], sigma=np.stack([u[1] for u in ups]),
                 mean=np.stack([u[2] for u in ups]))
    b.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
