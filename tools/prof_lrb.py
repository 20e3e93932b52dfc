"""Phase profile of the estimator kernel (a -DNM_LRB_PROF build of lowrank_device.hip: python tools/build_variant.py lrbprof "-DNM_LRB_PROF=1" lowrank_device.hip):
   NUTS_AMD_LIB=nuts_rs_amd/libnuts_amd_lrbprof.so python tools/prof_lrb.py [--dim 128] [--n 100] [--windows 1024]"""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from nuts_rs_amd import _lib

NAMES = {0: "rescale", 1: "qr draws", 2: "qr grads", 3: "qr subspace", 4: "project + cov", 6: "G^1/2, G^1/2 D G^1/2", 8: "sqrt, G^-1/2, products",
         10: "filter + output", 11: "eigh: tridiagonalise", 12: "eigh: accumulate", 13: "eigh: QL"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=128); ap.add_argument("--n", type=int, default=100); ap.add_argument("--windows", type=int, default=1024)
    a = ap.parse_args()
    import test_lowrank_estimator_builtin as T
    L = _lib.load()
    rng = np.random.default_rng(1)
    base = [T.correlated_window(rng, a.dim, a.n, 4) for _ in range(8)]
    D = np.stack([base[w % 8][0].T for w in range(a.windows)]); G = np.stack([base[w % 8][1].T for w in range(a.windows)])
    m = min(a.dim, 2 * a.n); nw = a.windows
    stds, mean, mu = np.zeros((nw, a.dim)), np.zeros((nw, a.dim)), np.zeros((nw, a.dim))
    vals, vecs = np.zeros((nw, m)), np.zeros((nw, m, a.dim))
    n_eig, status = np.zeros(nw, dtype=np.uint64), np.zeros(nw, dtype=np.uint64)
    prof = (C.c_ulonglong * 16)()
    has = hasattr(L, "nm_debug_lrb_prof")
    out = {}
    for rep in range(2):
        if has:
            L.nm_debug_lrb_prof(prof, 1)
        t = time.time()
        er = L.nm_lowrank_test_block_device(a.dim, a.n, nw, D.ctypes.data, G.ctypes.data, 1e-5, 2.0, stds.ctypes.data, mean.ctypes.data, n_eig.ctypes.data,
                                            vals.ctypes.data, vecs.ctypes.data, mu.ctypes.data, status.ctypes.data, None)
        out["wall_s_incl_copies"] = time.time() - t
        assert er == 0
    if has:
        L.nm_debug_lrb_prof(prof, 0)
        tot = sum(prof)
        out["ms_per_block"] = tot / nw / 1e5
        out["phases_ms_per_block"] = {NAMES.get(i, str(i)): round(prof[i] / nw / 1e5, 3) for i in range(16) if prof[i]}
    out.update(dim=a.dim, n=a.n, windows=nw, n_eig_median=float(np.median(n_eig)), failed=int(status.sum()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
