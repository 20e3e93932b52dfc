import ctypes as C, sys
sys.path.insert(0, '/root/repo')
import torch
import nuts_rs_amd as N
L = N.load_library()
out = (C.c_uint * 64)()
print('rc', L.nm_debug_xcc_ids(out, 64))
print(list(out))
