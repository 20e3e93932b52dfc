"""Reproduce / rule out the nm_vec_new zero-fill race.

nm_vec_new = hipMalloc + zero fill; the very next nm_vec_* call launches a kernel on the handle's hipStreamNonBlocking
stream that stores into the new vector.  If the fill is a null-stream operation it is not ordered against that stream and
may land after (part of) the kernel's store.  The probe repeats [new vector -> std_norm_flow into it -> read back] and
counts read-backs that differ from the expected values (element-wise formula, computed once by the device on a vector
that was synchronised first).  `NUTS_AMD_LIB=<path> python tools/probes/vec_new_race.py [iters] [dim]`."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nuts_rs_amd as N          # noqa: E402
from nuts_rs_amd import _lib     # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    L = _lib.load()
    spec = N.LogpSpec.iid_normal(dim, 0.0).to_c()
    h = C.c_void_p()
    assert L.nm_math_create(C.byref(spec), C.byref(h)) == 0
    rng = np.random.default_rng(1)
    p, v = rng.normal(size=dim), rng.normal(size=dim)

    def vec(data=None):
        x = C.c_void_p()
        assert L.nm_vec_new(h, C.byref(x)) == 0
        if data is not None:
            assert L.nm_vec_read_from_slice(h, x, np.ascontiguousarray(data).ctypes.data) == 0
        return x

    def get(x):
        out = np.empty(dim)
        assert L.nm_vec_write_to_slice(h, x, out.ctypes.data) == 0
        return out

    vp = vec(p)
    # reference result: the output vector is written by a blocking copy first, so nothing is pending on it
    vo, vv = vec(np.ones(dim)), vec(v)
    assert L.nm_vec_std_norm_flow(h, vp, vo, vv, C.c_double(0.37)) == 0
    want = get(vo)
    L.nm_vec_free(vo); L.nm_vec_free(vv)
    bad, first_bad, zeros = 0, None, 0
    keep = []
    for i in range(iters):
        vv = vec(v)
        vo = vec()                      # fresh: hipMalloc + zero fill, then straight into the kernel
        assert L.nm_vec_std_norm_flow(h, vp, vo, vv, C.c_double(0.37)) == 0
        got = get(vo)
        if not (got.view(np.uint64) == want.view(np.uint64)).all():
            bad += 1
            wrong = np.flatnonzero(got.view(np.uint64) != want.view(np.uint64))
            zeros += int((got[wrong] == 0.0).all())
            if first_bad is None:
                first_bad = (i, int(wrong[0]), int(wrong[-1]), len(wrong))
        keep.append((vo, vv))
        if len(keep) > 8:               # free with a delay so addresses are reused in varying patterns
            a, b = keep.pop(0)
            L.nm_vec_free(a); L.nm_vec_free(b)
    print({"lib": _lib.LIB_PATH, "iters": iters, "dim": dim, "mismatching_readbacks": bad,
           "of_which_all_wrong_elements_are_zero": zeros, "first_bad(iter,first_idx,last_idx,count)": first_bad})
    return 0


if __name__ == "__main__":
    sys.exit(main())
