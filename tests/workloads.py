"""Seeded synthetic workloads shared by tests, smoke() and bench.py (host side)."""
from __future__ import annotations

import numpy as np


def bcch_ccch_mix(pkg, n, seed, esn0_db=(6.0, 10.0, 20.0), sps=4, toa_jitter=8, frac=True,
                  cfo_hz_std=50.0, gain_db_std=6.0, stride_align=16):
    """BCCH:CCCH = 1:6, windows 1016 / 976 samples, TOA jitter, CFO, AWGN.

    Layout: iq[burst][sample], each window starting on a multiple of `stride_align` samples.
    Returns dict(iq (flat complex64), offset (uint64), kind (uint8), l2 (n,24) sent payloads, toa).
    """
    synth = pkg.synth
    rng = np.random.default_rng(seed)
    kind = (np.arange(n) % 7 != 0).astype(np.uint8)          # 1 BCCH then 6 CCCH (gmr1_rx.c:873-878)
    l2 = rng.integers(0, 256, size=(n, 24), dtype=np.uint8)
    fmt = [pkg.api.burst_format("bcch"), pkg.api.burst_format("dc6")]
    win = [20 * sps, 10 * sps]
    lens = [234 * sps + win[0], 234 * sps + win[1]]
    stride = [-(-l // stride_align) * stride_align for l in lens]
    offset = np.zeros(n, np.uint64)
    sizes = np.where(kind == 0, stride[0], stride[1]).astype(np.uint64)
    offset[1:] = np.cumsum(sizes)[:-1]
    total = int(sizes.sum())
    iq = np.zeros(total, np.complex64)
    toa = np.zeros(n)
    esn0 = rng.choice(np.asarray(esn0_db, dtype=np.float64), size=n)
    for k in (0, 1):
        rows = np.nonzero(kind == k)[0]
        if rows.size == 0:
            continue
        ebits = synth.bcch_encode(l2[rows]) if k == 0 else synth.ccch_encode(l2[rows])
        sym = synth.map_symbols(fmt[k], ebits)
        bb = synth.synth_windows(fmt[k], sym, sps, win[k], rng, toa_jitter=toa_jitter, frac=frac,
                                 cfo_hz_std=cfo_hz_std, esn0_db=esn0[rows], gain_db_std=gain_db_std,
                                 stride=stride[k])
        idx = offset[rows][:, None].astype(np.int64) + np.arange(stride[k])[None, :]
        iq[idx] = bb.iq
        toa[rows] = bb.toa
    return dict(iq=iq, offset=offset, kind=kind, l2=l2, toa=toa, esn0=esn0, in_len=lens)
