#!/usr/bin/env python
"""Host-link ceiling of the box: pinned host -> device copies of 256 MiB chunks (the size of the library's staging chunks),
and the rate at which host threads can fill such a chunk (memcpy / 3-way interleave), to read the C3 streaming numbers against."""
import json
import time

import numpy as np
import torch

n = 256 << 20
src = torch.empty(n, dtype=torch.uint8).pin_memory()
dst = torch.empty(n, dtype=torch.uint8, device="cuda")
for _ in range(3):
    dst.copy_(src, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 40
for _ in range(reps):
    dst.copy_(src, non_blocking=True)
torch.cuda.synchronize()
h2d = reps * n / (time.perf_counter() - t0) / 1e9
a = np.ones(n // 4, np.float32)
b = np.empty_like(a)
t0 = time.perf_counter()
for _ in range(5):
    np.copyto(b, a)
one_core = 5 * n / (time.perf_counter() - t0) / 1e9
print(json.dumps({"pinned_h2d_GBps_256MiB_chunks": h2d, "one_core_memcpy_GBps": one_core}))
