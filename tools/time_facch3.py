#!/usr/bin/env python3
"""Device-resident timing of gmr1_hip_facch3_decode_batch_dev (no copies in the timed region)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
api = pkg.api
api.load()
api.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000
rng = np.random.default_rng(0)
eb = torch.from_numpy(rng.integers(-127, 128, size=(n, 416), dtype=np.int8)).cuda()
l2 = torch.zeros((n, 10), dtype=torch.uint8, device="cuda")
crc = torch.zeros(n, dtype=torch.int32, device="cuda")
conv = torch.zeros(n, dtype=torch.int32, device="cuda")
f = api.load().gmr1_hip_facch3_decode_batch_dev
f.restype = C.c_int
st = torch.cuda.current_stream().cuda_stream


def step():
    assert f(C.c_void_p(st), C.c_int(n), C.c_void_p(eb.data_ptr()), None, C.c_void_p(l2.data_ptr()), None,
             C.c_void_p(crc.data_ptr()), C.c_void_p(conv.data_ptr())) == 0


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"facch3: {n} frames in {dt * 1e3:.3f} ms = {n / dt / 1e6:.1f} Mframes/s")
