#!/usr/bin/env python3
"""Device-resident timing of gmr1_hip_xch_dc12_decode_batch_dev and gmr1_hip_rach_decode_batch_dev
(no copies in the timed region)."""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
rng = np.random.default_rng(0)
st = torch.cuda.current_stream().cuda_stream


def timed(name, step, unit_bytes):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {n} bursts in {dt * 1e3:.3f} ms = {n / dt / 1e6:.2f} Mbursts/s, "
          f"{n * unit_bytes / dt / 1e9:.1f} GB/s algorithmic")


eb = torch.from_numpy(rng.integers(-127, 128, size=(n, 432), dtype=np.int8)).cuda()
l2 = torch.zeros((n, 24), dtype=torch.uint8, device="cuda")
crc = torch.zeros(n, dtype=torch.int32, device="cuda")
conv = torch.zeros(n, dtype=torch.int32, device="cuda")
f = api.load().gmr1_hip_xch_dc12_decode_batch_dev
f.restype = C.c_int
timed("xch_dc12", lambda: f(C.c_void_p(st), C.c_int(n), C.c_void_p(eb.data_ptr()), C.c_void_p(l2.data_ptr()),
                            C.c_void_p(crc.data_ptr()), C.c_void_p(conv.data_ptr())), 432 + 24 + 8)

eb2 = torch.from_numpy(rng.integers(-127, 128, size=(n, 494), dtype=np.int8)).cuda()
msk = torch.zeros(n, dtype=torch.uint8, device="cuda")
rach = torch.zeros((n, 18), dtype=torch.uint8, device="cuda")
rv = torch.zeros(n, dtype=torch.int32, device="cuda")
crc2 = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
g = api.load().gmr1_hip_rach_decode_batch_dev
g.restype = C.c_int
timed("rach", lambda: g(C.c_void_p(st), C.c_int(n), C.c_void_p(eb2.data_ptr()), C.c_void_p(msk.data_ptr()),
                        C.c_void_p(rach.data_ptr()), C.c_void_p(rv.data_ptr()), C.c_void_p(conv.data_ptr()),
                        C.c_void_p(crc2.data_ptr())), 494 + 1 + 18 + 16)
