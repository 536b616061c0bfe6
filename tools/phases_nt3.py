#!/usr/bin/env python3
"""Time gmr1_hip_demod_batch_dev on NT3 speech bursts per phase cut-off (GMR1_HIP_DBG_STOP = 1 load+normalise,
2 correlation, 3 peak / timing, 4 frequency, 5 phase, 0 everything): one process per cut-off (the variable is read once).
Usage: phases_nt3.py [stop] [bursts]"""
import ctypes as C
import os
import subprocess
import sys
import time

if len(sys.argv) < 2:
    for st in (1, 2, 3, 4, 5, 0):
        # the cut-offs exist only in the profiling build: python osmo-gmr_amd/build.py --profile
        prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "osmo-gmr_amd", "libgmr1_hip_prof.so")
        env = dict(os.environ, GMR1_HIP_DBG_STOP=str(st), GMR1_HIP_LIBRARY=os.environ.get("GMR1_HIP_LIBRARY", prof))
        subprocess.run([sys.executable, os.path.abspath(__file__), str(st)], env=env, check=False)
    sys.exit(0)

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402
import workloads  # noqa: E402

pkg = load_package()
api = pkg.api
torch.cuda.init()
api.load()
api.init(0)
L = api.load()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
base = 40_000
wl = workloads.nt3_mix(pkg, base, seed=5)
reps = n // base
iq = torch.from_numpy(wl["iq"].view(np.float32)).cuda().repeat(reps)
sp = wl["speech"]
off = torch.from_numpy(np.concatenate([(sp + r * base) for r in range(reps)]).astype(np.int64) * wl["stride"]).cuda()
fs = torch.from_numpy(np.tile(wl["freq_shift"][sp], reps)).cuda()
m = off.numel()
eb = torch.zeros((m, 212), dtype=torch.int8, device="cuda")
sid = torch.zeros(m, dtype=torch.int32, device="cuda")
toa = torch.zeros(m, dtype=torch.float32, device="cuda")
rv = torch.zeros(m, dtype=torch.int32, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def step():
    return L.gmr1_hip_demod_batch_dev(st, C.c_int(4), C.c_int(m), C.c_int(4), C.c_int(474), P(iq), P(off), P(fs), P(eb), C.c_int(212),
                                      P(sid), P(toa), None, None, P(rv))


for _ in range(3):
    assert step() == 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"stop={sys.argv[1]}: {m} NT3 speech bursts in {dt * 1e3:.3f} ms = {m / dt / 1e6:.1f} Mbursts/s", flush=True)
