#!/usr/bin/env python3
"""Which soft bits of the fused path differ from the oracle's by more than 1 LSB, and why (GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import oracle_lib, workloads
pkg = load_package(); api = pkg.api; api.load(); api.init(0); oracle_lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2001
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
wl = workloads.bcch_ccch_mix(pkg, n=n, seed=seed)
got = api.rx_bcch_ccch_batch(wl["iq"], wl["offset"], wl["kind"], sps=4)
ref = oracle_lib.demod_decode_batch(wl["iq"], wl["offset"], wl["kind"], sps=4)
same = (got["toa"] == ref["toa"])
deb = np.abs(got["ebits"].astype(int) - ref["ebits"].astype(int))
deb[~same] = 0
print("bursts", n, "same toa", int(same.sum()), "bits differing by 1:", int((deb == 1).sum()), " by more:", int((deb > 1).sum()))
fm = [api.burst_format("bcch"), api.burst_format("dc6")]
for b, e in zip(*np.nonzero(deb > 1)):
    f = fm[wl["kind"][b]]
    # data symbol ordinal -> symbol position
    pos = np.concatenate([np.arange(p, p + l) for p, l in f.data])
    i = pos[e // 2]
    print(f"burst {b} kind {wl['kind'][b]} ebit {e} sym {i}: gpu {got['ebits'][b, e]} orc {ref['ebits'][b, e]}  "
          f"ssym gpu {got['ssyms'][b, i]!r} orc {ref['ssyms'][b, i]!r}")
