#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline path through the host-pointer entry point
(gmr1_hip_rx_bcch_ccch_batch: hipMalloc + H2D of the IQ + kernel + D2H of the results)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  (brings the ROCm runtime up first)
from __graft_entry__ import load_package  # noqa: E402
import workloads  # noqa: E402

torch.cuda.init()
pkg = load_package()
api = pkg.api
api.load()
api.init(0)
n = 100_000
wl = workloads.bcch_ccch_mix(pkg, n=n, seed=3)
for _ in range(2):
    api.rx_bcch_ccch_batch(wl["iq"], wl["offset"], wl["kind"], sps=4, want_ebits=False, want_ssyms=False)
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    r = api.rx_bcch_ccch_batch(wl["iq"], wl["offset"], wl["kind"], sps=4, want_ebits=False, want_ssyms=False)
dt = (time.perf_counter() - t0) / reps
gb = wl["iq"].nbytes / 1e9
print(f"host boundary: {n} bursts ({gb:.2f} GB of IQ from pageable host memory) in {dt * 1e3:.1f} ms = "
      f"{n / dt / 1e6:.2f} Mbursts/s, {gb / dt:.1f} GB/s over PCIe incl. allocation")
