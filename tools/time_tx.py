#!/usr/bin/env python3
"""Device-resident timing of the transmit direction: gmr1_hip_{bcch,tch3,tch9}_encode_batch_dev and
gmr1_hip_mod_batch_dev (no copies in the timed region).  Usage: time_tx.py [bursts]"""
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
torch.cuda.init()
api.load()
api.init(0)
L = api.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def timed(name, step, unit_bytes, units=n):
    for _ in range(3):
        rc = step()
        assert rc == 0, (name, rc, L.gmr1_hip_last_error())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {units} bursts in {dt * 1e3:.3f} ms = {units / dt / 1e6:.1f} Mbursts/s, "
          f"{units * unit_bytes / dt / 1e9:.1f} GB/s algorithmic ({unit_bytes} B per burst)", flush=True)


l2 = dev(rng.integers(0, 256, (n, 24), dtype=np.uint8))
eb = torch.zeros((n, 432), dtype=torch.uint8, device="cuda")
timed("bcch_encode", lambda: L.gmr1_hip_bcch_encode_batch_dev(st, C.c_int(n), ptr(l2), ptr(eb)), 24 + 424)
timed("xch_dc12_encode", lambda: L.gmr1_hip_xch_dc12_encode_batch_dev(st, C.c_int(n), ptr(l2), ptr(eb)), 24 + 432)

fr = dev(rng.integers(0, 256, (n, 20), dtype=np.uint8))
bs = dev(rng.integers(0, 2, (n, 4), dtype=np.uint8))
ci = dev(rng.integers(0, 2, (n, 208), dtype=np.uint8))
e3 = torch.zeros((n, 212), dtype=torch.uint8, device="cuda")
timed("tch3_encode (ciphered)", lambda: L.gmr1_hip_tch3_encode_batch_dev(st, C.c_int(n), C.c_int(0), ptr(fr), ptr(bs), ptr(ci), ptr(e3)),
      20 + 4 + 208 + 212)

n9 = n // 4
p9 = dev(rng.integers(0, 256, (n9, 60), dtype=np.uint8))
sa = dev(rng.integers(0, 2, (n9, 10), dtype=np.uint8))
stt = dev(rng.integers(0, 2, (n9, 4), dtype=np.uint8))
e9 = torch.zeros((n9, 662), dtype=torch.uint8, device="cuda")
timed("tch9_9k6_encode (runs of 1000)", lambda: L.gmr1_hip_tch9_encode_batch_dev(st, C.c_int(2), C.c_int(n9), C.c_int(1000), ptr(p9), ptr(sa),
                                                                               ptr(stt), None, ptr(e9)), 60 + 14 + 662, units=n9)

info = api.burst_info("bcch")
ebm = dev(rng.integers(0, 2, (n, info.ebits), dtype=np.uint8))
out = torch.zeros((n, info.len, 2), dtype=torch.float32, device="cuda")
timed("mod (BCCH burst format)", lambda: L.gmr1_hip_mod_batch_dev(st, C.c_int(0), C.c_int(0), C.c_int(n), ptr(ebm), ptr(out)),
      info.ebits + info.len * 8)
