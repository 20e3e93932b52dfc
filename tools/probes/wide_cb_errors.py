import numpy as np, sys
sys.path.insert(0, ".")
import nuts_rs_amd as N
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
n = 3
s = N.DiagNutsSettings(num_chains=n, seed=8, num_tune=10, maxdepth=4)
state = {"calls": 0, "raised": 0}
def flaky(chain, x):
    state["calls"] += 1
    if chain == 1 and state["calls"] > 40:
        state["raised"] += 1
        raise ValueError("unrecoverable")
    if chain == 2 and abs(x[dim - 1]) > 2.5:
        raise N.RecoverableLogpError()
    return -0.5 * float(np.dot(x, x)), -x
b = N.ChainBatch(s, N.LogpSpec.host_callback(dim, flaky, threads=1), n)
print("set_position", b.set_position(b.init_positions_uniform(), raise_on_error=False))
pos, st = b.draw_many(16, raise_on_error=False)
print(state, b.host_logp_calls())
print(st["chain_status"].T)
print(st["diverging"].T)
print(st["n_steps"].T)
print(b.status() if hasattr(b, "status") else "")
b.close()
