#!/usr/bin/env python3
"""Device-resident timing of the NT9 decoders: gmr1_hip_facch9_decode_batch_dev and gmr1_hip_tch9_decode_batch_dev
(modes 0 / 1 / 2 = 2k4 / 4k8 / 9k6), random soft bits.  GPU box, repo root: python3 tools/time_nt9.py [bursts]"""
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
lib = api.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1)
eb = torch.randint(-127, 128, (n, 662), generator=g, device=dev, dtype=torch.int8)
l2 = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
sa = torch.zeros((n, 10), dtype=torch.int8, device=dev)
stt = torch.zeros((n, 4), dtype=torch.int8, device=dev)
crc = torch.zeros(n, dtype=torch.int32, device=dev)
conv = torch.zeros(n, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def timed(name, step):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%-12s %8.3f ms per %d bursts  %7.1f Mbursts/s" % (name, dt * 1e3, n, n / dt / 1e6))


def facch9():
    rc = lib.gmr1_hip_facch9_decode_batch_dev(C.c_void_p(st), C.c_int(n), C.c_void_p(eb.data_ptr()), None,
                                              C.c_void_p(l2.data_ptr()), C.c_void_p(sa.data_ptr()), C.c_void_p(stt.data_ptr()),
                                              C.c_void_p(crc.data_ptr()), C.c_void_p(conv.data_ptr()))
    assert rc == 0, rc


timed("facch9", facch9)
seq = 100
for mode, name in ((0, "tch9 2k4"), (1, "tch9 4k8"), (2, "tch9 9k6")):
    def tch9(mode=mode):
        rc = lib.gmr1_hip_tch9_decode_batch_dev(C.c_void_p(st), C.c_int(n // seq), C.c_int(seq), C.c_int(mode),
                                                C.c_void_p(eb.data_ptr()), None, C.c_void_p(l2.data_ptr()),
                                                C.c_void_p(sa.data_ptr()), C.c_void_p(stt.data_ptr()), C.c_void_p(conv.data_ptr()))
        assert rc == 0, rc
    timed(name, tch9)
