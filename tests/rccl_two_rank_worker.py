"""Worker of tests/test_gpu_distributed.py::test_pooled_exchange_over_a_real_rccl_communicator: rank `argv[1]` of a TWO-rank RCCL
communicator created with ncclCommInitRank (ctypes on librccl — no torch.distributed), both ranks on GPU 0 where RCCL permits that.
The unique id travels through a file in argv[2].  Writes result_<rank>.json: {"ok": true, ...} or {"skip": "<RCCL's reason>"}."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, d = int(sys.argv[1]), sys.argv[2]
    out = os.path.join(d, f"result_{rank}.json")

    def done(obj):
        json.dump(obj, open(out, "w"))
        sys.exit(0)

    import torch                                       # the HIP runtime + device buffers
    from nuts_rs_amd import _lib
    L = _lib.load()
    torch.cuda.set_device(0)
    rccl = None
    for name in (os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so", "librccl.so.1"):
        try:
            rccl = C.CDLL(name, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            continue
    if rccl is None:
        done({"skip": "librccl.so not found"})

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    rccl.ncclGetErrorString.restype = C.c_char_p
    uid = UniqueId()
    idf = os.path.join(d, "uid.bin")
    if rank == 0:
        rc = rccl.ncclGetUniqueId(C.byref(uid))
        if rc != 0:
            done({"skip": f"ncclGetUniqueId: {rccl.ncclGetErrorString(rc).decode()}"})
        open(idf + ".tmp", "wb").write(bytes(uid.internal))
        os.replace(idf + ".tmp", idf)
    else:
        t0 = time.time()
        while not os.path.exists(idf):
            if time.time() - t0 > 60:
                done({"skip": "rank 0 never published the unique id"})
            time.sleep(0.05)
        C.memmove(C.byref(uid), open(idf, "rb").read(), 128)
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rc = rccl.ncclCommInitRank(C.byref(comm), 2, uid, rank)
    if rc != 0:
        done({"skip": f"ncclCommInitRank with two ranks on one GPU: {rccl.ncclGetErrorString(rc).decode()} (rc {rc})"})
    dim = 37
    w = 2 * (1 + 2 * dim)
    payload = torch.arange(w, dtype=torch.float64, device="cuda") + 1000.0 * (rank + 1)
    gathered = torch.zeros(2 * w, dtype=torch.float64, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    L.nm_pooled_exchange.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.nm_pooled_last_error.restype = C.c_char_p
    rc = L.nm_pooled_exchange(comm, 2, dim, payload.data_ptr(), gathered.data_ptr(), st.cuda_stream)
    st.synchronize()
    if rc != 0:
        done({"ok": False, "error": (L.nm_pooled_last_error() or b"").decode()})
    g = gathered.cpu().numpy().reshape(2, w)
    want = np.stack([np.arange(w) + 1000.0, np.arange(w) + 2000.0])
    ok = bool((g == want).all())
    rccl.ncclCommDestroy(comm)
    done({"ok": ok, "rank": rank, "first": g[:, 0].tolist()})


if __name__ == "__main__":
    main()
