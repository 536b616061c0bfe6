#!/usr/bin/env python3
"""Cycle stamps of one BCCH burst inside the pipelined receive loop (chain 0, round 55): what each of its three stages takes.
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
A, sps, seconds = 64, 4, 60.0      # (round 55 = the stamped one sits in the middle of the first time slice at this length)
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
out = (C.c_ulonglong * 32)()
assert L.gmr1_hip_prof_stamps(out) == 0
t = np.array(list(out), np.int64)
print("cycle stamps of chain 0's round 55 (shader clock).  The burst runs through four stages, a tick apart: F = front (S beside it), P = pass 2 + operand table, V = decoder forward pass, S = decoder tail + verdict")
f = [("window (prepared: statistics only)", 1, 0), ("staging + correlation", 2, 1), ("parameters looked up", 10, 2), ("window argmax reduced", 11, 10),
     ("coarse peak", 8, 11), ("nine halvings", 9, 8), ("peak value + bookkeeping", 3, 9), ("sync terms (frequency, phase)", 4, 3)]
print(f"F: front, {t[4] - t[0]} cycles")
for name, k, k0 in f:
    print(f"  {name:38s} +{t[k] - t[k0]:6d}")
print(f"P: pass 2 + operand table, {t[6] - t[14]} cycles (starts {t[14] - t[0]} after F's start)")
print(f"  {'pass 2 (soft bits)':38s} +{t[5] - t[14]:6d}")
print(f"  {'branch metrics -> operand table':38s} +{t[6] - t[5]:6d}")
print(f"V: decoder's forward pass, {t[12] - t[15]} cycles (starts {t[15] - t[0]} after F's start)")
print(f"S: the decoder's tail a tick later (beside its help to F): survivor walk + CRC {t[7] - t[24]} cycles (starts {t[24] - t[0]} after F's start)")
print(f"  {'survivor walk':38s} +{t[13] - t[24]:6d}")
print(f"  {'CRC':38s} +{t[7] - t[13]:6d}")
print(f"F's tick around that front (walk, hand-over, listing): {t[23] - t[16]} cycles")
for name, k, k0 in [("pred published, barrier passed", 17, 16), ("verdict checked", 18, 17), ("CCCH bursts listed (global)", 19, 18),
                    ("burst operands set, front started", 0, 19), ("front", 4, 0), ("results read, feedback assumed, hand-over written", 20, 4),
                    ("frame logged, chain advanced", 21, 20), ("round logged (global)", 22, 21), ("next round listed", 23, 22)]:
    print(f"  {name:50s} +{t[k] - t[k0]:6d}")
if t[25] > 0:
    print(f"chain 0's whole walk: {t[27]} rounds in {t[25]} ticks ({t[26]} squashes), {t[28]} shader cycles = {t[28] / t[25]:.0f} per tick; "
          f"{t[29] / 100.0:.1f} us on the 100 MHz clock = {t[29] / 100.0 / t[25]:.3f} us per tick, shader clock {t[28] / (t[29] / 100.0) / 1e3:.2f} GHz")
