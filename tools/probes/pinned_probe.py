import time, torch, numpy as np
n = 3_400_000_000 // 8
d = torch.empty(n, dtype=torch.float64, device="cuda"); d.fill_(1.0); torch.cuda.synchronize()
t = time.time(); h = torch.empty(n, dtype=torch.float64, pin_memory=True); t_alloc = time.time() - t
t = time.time(); h.copy_(d, non_blocking=True); torch.cuda.synchronize(); t_pin = time.time() - t
t = time.time(); h.copy_(d, non_blocking=True); torch.cuda.synchronize(); t_pin2 = time.time() - t
p = np.empty(n)
t = time.time(); pt = torch.from_numpy(p); pt.copy_(d); torch.cuda.synchronize(); t_page = time.time() - t
t = time.time(); pt.copy_(d); torch.cuda.synchronize(); t_page2 = time.time() - t
t = time.time(); torch.cuda.cudart().cudaHostRegister(p.ctypes.data, p.nbytes, 0); t_reg = time.time() - t
t = time.time(); pt.copy_(d, non_blocking=True); torch.cuda.synchronize(); t_regcopy = time.time() - t
t = time.time(); p2 = np.empty(n); p2[:] = 0; t_touch = time.time() - t
del h
t = time.time(); h = torch.empty(n, dtype=torch.float64, pin_memory=True); t_alloc2 = time.time() - t
print(dict(GB=n * 8 / 1e9, pinned_alloc_s=t_alloc, pinned_realloc_s=t_alloc2, d2h_pinned_GBps=n * 8 / t_pin / 1e9, d2h_pinned2_GBps=n * 8 / t_pin2 / 1e9,
           d2h_pageable_first_GBps=n * 8 / t_page / 1e9, d2h_pageable_GBps=n * 8 / t_page2 / 1e9, host_register_s=t_reg,
           d2h_registered_GBps=n * 8 / t_regcopy / 1e9, numpy_alloc_touch_s=t_touch))
