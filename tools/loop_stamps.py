#!/usr/bin/env python3
"""Cycle stamps of one BCCH burst inside the receive loop (chain 0, round 55): where a round's 19 microseconds go.
Needs the profiling build (python osmo-gmr_amd/build.py --profile).  Run on the GPU box from the repo root."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GMR1_HIP_LIBRARY", os.path.join(ROOT, "osmo-gmr_amd", "libgmr1_hip_prof.so"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from __graft_entry__ import load_package
import workloads

pkg = load_package()
api = pkg.api
torch.cuda.init()
L = api.load()
api.init(0)
A, sps, seconds = 64, 4, 20.0
ns = int(seconds * 23400 * sps)
host = [workloads.bcch_carrier(pkg, 700 + a, seconds=seconds, sps=sps, stn=(5 * a) % 24, delay=a % 8, cfo_hz=40.0 * (a - 3),
                               esn0_db=10.0 + a)[0] for a in range(8)]
base = torch.from_numpy(np.concatenate(host).view(np.float32)).cuda()
iq = torch.cat([base] * 8)[:A * ns * 2].contiguous()
offset = np.arange(A, dtype=np.uint64) * np.uint64(ns)
length = np.full(A, ns, np.uint64)
st = torch.cuda.current_stream().cuda_stream
if os.environ.get("LOOP_FLAG"):
    assert L.gmr1_hip_prof_flag(int(os.environ["LOOP_FLAG"])) == 0
for _ in range(3):
    api.rx_run_dev(st, iq.data_ptr(), offset, length, sps=sps)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    api.rx_run_dev(st, iq.data_ptr(), offset, length, sps=sps)
e1.record()
torch.cuda.synchronize()
print(f"{A} carriers x {seconds} s: {e0.elapsed_time(e1) / 5:.3f} ms per run (acquisition + loop)")
out = (C.c_ulonglong * 16)()
assert L.gmr1_hip_prof_stamps(out) == 0
t = np.array(list(out), np.int64)
names = ["start", "window loaded + statistics", "staging + correlation", "peak + timing bisection", "sync terms (frequency, phase)",
         "pass 2 (soft bits)", "branch metrics", "Viterbi + survivors + CRC"]
print("cycles since the burst started (shader clock), and per phase:")
for k in range(1, 8):
    print(f"  {names[k]:34s} {t[k] - t[0]:8d}  (+{t[k] - t[k - 1]})")
print(f"  inside the timing phase: coarse peak found at {t[8] - t[0]} (+{t[8] - t[2]}), nine halvings done at {t[9] - t[0]} (+{t[9] - t[8]}), "
      f"peak value + bookkeeping +{t[3] - t[9]}")
print(f"  finer: parameters looked up at +{t[10] - t[2]} of the timing phase, window argmax reduced at +{t[11] - t[2]}, coarse peak at +{t[8] - t[2]}; "
      f"decoder: forward pass {t[12] - t[6]}, survivor walk {t[13] - t[12]}, CRC + results {t[7] - t[13]}")
