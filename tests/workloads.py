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


def fcch_streams(pkg, n, seed, n_samples=93600, sps=4, snr_db=(0.0, 6.0), cfo_hz=2000.0):
    """n independent 1-s streams, AWGN + dual-chirp FCCH every 320 ms,
    SNR 0 / +6 dB, CFO uniform +-2 kHz.  Returns dict(iq (n, n_samples) complex64, offset, starts)."""
    synth = pkg.synth
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, n_samples * 2), dtype=np.float32).view(np.complex64)
    x *= np.float32(1.0 / np.sqrt(2.0))
    chirp = synth.fcch_dual_chirp(0.32, 117, sps)
    period = 7488 * sps
    first = rng.integers(0, period, size=n)
    snr = rng.choice(np.asarray(snr_db, dtype=np.float64), size=n)
    amp = np.sqrt(10.0 ** (snr / 10.0)).astype(np.float32)
    starts = []
    for i in range(n):
        st = []
        pos = int(first[i])
        while pos + chirp.size <= n_samples:
            x[i, pos:pos + chirp.size] += chirp * amp[i]
            st.append(pos)
            pos += period
        starts.append(st)
    cfo = rng.uniform(-cfo_hz, cfo_hz, size=n)
    step = (2 * np.pi * cfo / (synth.SYM_RATE * sps)).astype(np.float32)
    t = np.arange(n_samples, dtype=np.float32)
    for i0 in range(0, n, 64):          # chunked to bound temporaries
        ph = step[i0:i0 + 64, None] * t[None, :]
        rot = np.empty(ph.shape, np.complex64)
        rot.real = np.cos(ph)
        rot.imag = np.sin(ph)
        x[i0:i0 + 64] *= rot
    offset = (np.arange(n, dtype=np.uint64) * np.uint64(n_samples))
    return dict(iq=x, offset=offset, starts=starts, cfo=cfo, snr=snr, n_samples=n_samples)


def tch3_bursts(pkg, n, seed, m=0, sigma=40.0):
    """BASELINE.md config 5, l1-only variant: n NT3 speech bursts as int8 soft bits (212 each)."""
    rng = np.random.default_rng(seed)
    f0 = rng.integers(0, 256, (n, 10), dtype=np.uint8)
    f1 = rng.integers(0, 256, (n, 10), dtype=np.uint8)
    s = rng.integers(0, 2, (n, 4), dtype=np.uint8)
    bits = pkg.synth.tch3_encode(f0, f1, s, m)
    eb = 80.0 * (1.0 - 2.0 * bits.astype(np.float32))
    eb += rng.standard_normal(eb.shape, dtype=np.float32) * np.float32(sigma)
    eb = np.clip(np.rint(eb), -127, 127).astype(np.int8)
    return dict(ebits=eb, frame0=f0, frame1=f1, status=s, m=m)


def nt3_mix(pkg, n, seed, esn0_db=(6.0, 10.0, 20.0), sps=4, m=0, stride_align=16):
    """BASELINE.md config 5 from samples: n NT3 bursts (n a multiple of 40), 90 % speech (two random 80-bit frames
    each) and 10 % FACCH3 in groups of four consecutive bursts (sync sequence alternating per group), window 474
    samples, impairments as config 3 with the timing jitter the 7-lag window of rx_tch3 allows (gmr1_rx.c:549-550: win = 1.5 sps; +-1 sample + fraction), CFO, phase, AWGN, gain.  An NT3 burst has one short training sequence, so the demodulator cannot estimate a frequency
    error itself: as in rx_tch3 (gmr1_rx.c:509-512, -cd->freq_err from the BCCH tracking) the carrier offset is handed
    to it, here as `freq_shift` = minus the offset the generator applied (rad / symbol).

    Layout: iq[burst][480] complex64 (474 + padding to 16 samples).  Returns dict(iq flat, offset, kind (0 speech,
    1 FACCH3), speech (indices), facch (indices, groups of 4), frames (n_s, 2, 10), status (n_s, 4), l2 (n_f / 4, 10),
    bits_s (n_f / 4, 32), sync_id (n_f / 4), freq_shift (n,))."""
    assert n % 40 == 0
    synth = pkg.synth
    rng = np.random.default_rng(seed)
    kind = ((np.arange(n) % 40) >= 36).astype(np.uint8)
    speech = np.nonzero(kind == 0)[0]
    facch = np.nonzero(kind == 1)[0]
    fmt_s, fmt_f = pkg.api.burst_format("nt3_speech"), pkg.api.burst_format("nt3_facch")
    win = 6
    in_len = 117 * sps + win
    stride = -(-in_len // stride_align) * stride_align
    offset = np.arange(n, dtype=np.uint64) * np.uint64(stride)
    iq = np.zeros((n, stride), np.complex64)
    esn0 = rng.choice(np.asarray(esn0_db, dtype=np.float64), size=n)
    # speech
    frames = rng.integers(0, 256, (speech.size, 2, 10), dtype=np.uint8)
    status = rng.integers(0, 2, (speech.size, 4), dtype=np.uint8)
    eb = synth.tch3_encode(frames[:, 0], frames[:, 1], status, m)
    bb = synth.synth_windows(fmt_s, synth.map_symbols(fmt_s, eb), sps, win, rng, toa_jitter=1, frac=True, cfo_hz_std=50.0,
                             esn0_db=esn0[speech], gain_db_std=6.0, stride=stride)
    iq[speech] = bb.iq
    freq_shift = np.zeros(n, np.float32)
    freq_shift[speech] = -bb.cfo * sps
    # FACCH3: one 10-byte message over four bursts
    ng = facch.size // 4
    l2 = rng.integers(0, 256, (ng, 10), dtype=np.uint8)
    l2[:, 9] &= 0x0f                                          # 76 bits
    bits_s = rng.integers(0, 2, (ng, 32), dtype=np.uint8)
    ebf = synth.facch3_encode(l2, bits_s).reshape(ng * 4, 104)
    sync_id = (np.arange(ng) & 1).astype(np.int32)
    bb = synth.synth_windows(fmt_f, synth.map_symbols(fmt_f, ebf, np.repeat(sync_id, 4)), sps, win, rng, toa_jitter=1,
                             frac=True, cfo_hz_std=50.0, esn0_db=esn0[facch], gain_db_std=6.0, stride=stride)
    iq[facch] = bb.iq
    freq_shift[facch] = -bb.cfo * sps
    return dict(iq=iq.reshape(-1), offset=offset, kind=kind, speech=speech, facch=facch, frames=frames, status=status,
                l2=l2, bits_s=bits_s, sync_id=sync_id, freq_shift=freq_shift, esn0=esn0, in_len=in_len, stride=stride, m=m)


def bcch_carrier(pkg, seed, seconds=2.0, sps=4, **kw):
    """One BCCH carrier (one ARFCN): FCCH + SI1 BCCH + CCCH on the TDMA grid."""
    from importlib import import_module
    synth = import_module(pkg.__name__ + ".synth")
    rng = np.random.default_rng(seed)
    fb = pkg.api.burst_format("bcch")
    fd = pkg.api.burst_format("dc6")
    n = int(seconds * 23400 * sps)
    return synth.synth_bcch_carrier(fb, fd, n, sps, rng, **kw)


def match_records(records, sent):
    """(#BCCH records whose (fn, l2) equal a sent burst, #BCCH records, #CCCH matched by payload, #CCCH records,
    #BCCH records matched by payload alone)."""
    sb = {(s["fn"], bytes(s["l2"])) for s in sent if s["type"] == "bcch"}
    sc = {bytes(s["l2"]) for s in sent if s["type"] == "ccch"}
    rb = [r for r in records if r["type"] == 1]
    rc = [r for r in records if r["type"] == 2]
    mb = sum((int(r["fn"]), bytes(r["l2"])) in sb for r in rb)
    mc = sum(bytes(r["l2"]) in sc for r in rc)
    mp = sum(bytes(r["l2"]) in {l for _, l in sb} for r in rb)
    return mb, len(rb), mc, len(rc), mp


def bcch_tch_pair(pkg, seed, seconds=4.0, sps=4, stn=3, delay=2, tn=11, p=20, k_ass=20, kc=None,
                  cipher_after=None, esn0_db=25.0, cfo_hz=60.0, mix=(0.35, 0.35, 0.3), k_stop=None):
    """A BCCH carrier whose CCCH carries an IMMEDIATE ASSIGNMENT around frame index k_ass, and the
    traffic carrier it points to (timeslot tn, DKAB position p), time-aligned, same CFO.
    cipher_after: frames after the assignment from which the TCH is A5/1-ciphered with kc."""
    from importlib import import_module
    synth = import_module(pkg.__name__ + ".synth")
    rng = np.random.default_rng(seed)
    n = int(seconds * 23400 * sps)
    frame_len = 24 * 39 * sps
    t0 = int(rng.integers(0, frame_len))
    fn0 = int(rng.integers(0, 1 << 18))
    fb, fd = pkg.api.burst_format("bcch"), pkg.api.burst_format("dc6")
    fs, ff = pkg.api.burst_format("nt3_speech"), pkg.api.burst_format("nt3_facch")
    bcch, sent = synth.synth_bcch_carrier(fb, fd, n, sps, rng, stn=stn, delay=delay, fn0=fn0, t0=t0,
                                          esn0_db=esn0_db, cfo_hz=cfo_hz, imm_ass=[(k_ass, tn, p)])
    ia = [s for s in sent if s["type"] == "ccch" and s.get("imm_ass")]
    assert ia, "no CCCH frame carried the assignment"
    k_start = ia[0]["k"]
    tch, sent_t = synth.synth_tch3_carrier(fs, ff, n, sps, rng, t0=t0, fn0=fn0, k_start=k_start, tn=tn, p=p,
                                           kc=kc, cipher_from=None if cipher_after is None else k_start + cipher_after,
                                           esn0_db=esn0_db, cfo_hz=cfo_hz, mix=mix, k_stop=k_stop)
    return bcch, tch, sent, sent_t


def wideband_capture(pkg, seed, seconds=2.5, samp_rate=2.0e6, carriers=((3, {}), (17, {}), (60, {})), sps=4):
    """A wideband capture (wideband container): each (channel, kwargs) is a BCCH
    carrier sent with root-raised-cosine pulses on ARFCN raster position `channel` (k x 31.25 kHz from the
    centre, k >= 32 below it), resampled 93.6 k -> samp_rate and summed.  Returns (wide, {channel: sent})."""
    from importlib import import_module
    from scipy.signal import resample_poly
    synth = import_module(pkg.__name__ + ".synth")
    rng = np.random.default_rng(seed)
    n_nb = int(seconds * 23400 * sps)
    up, down = 2500, 117                       # 2.0e6 / 93.6e3
    assert abs(samp_rate / (23400 * sps) - up / down) < 1e-9
    n_w = n_nb * up // down
    wide = np.zeros(n_w, np.complex64)
    sents = {}
    fb, fd = pkg.api.burst_format("bcch"), pkg.api.burst_format("dc6")
    t = np.arange(n_w, dtype=np.float64)
    for ch, kw in carriers:
        kw = dict(kw)
        kw.setdefault("esn0_db", 25.0)
        nb, sent = synth.synth_bcch_carrier(fb, fd, n_nb, sps, rng, pulse="rrc", span=8, **kw)
        w = resample_poly(nb.astype(np.complex128), up, down)[:n_w]
        k = ch if ch < 32 else ch - 64
        wide += (w * np.exp(2j * np.pi * (k * 31250.0 / samp_rate) * t)).astype(np.complex64)
        sents[ch] = sent
    return wide, sents


def bcch_tch_csd_triple(pkg, orc, seed, seconds=6.0, sps=4, stn=3, delay=2, tn=11, p=20, tn9=5, k_ass=15,
                        k_cmd=25, kc=None, esn0_db=25.0, cfo_hz=40.0, mix9=(0.3, 0.6)):
    """BCCH carrier with an IMM.ASS, the TCH3 carrier it points to -- whose FACCH3 messages from frame k_cmd
    on are ASSIGNMENT COMMAND 1 to timeslot tn9 -- and the CSD carrier with NT9 bursts on that timeslot:
    FACCH9 (sync sequence 0) or TCH9 9k6 (sync sequence 1), always A5/1-ciphered with kc (gmr1_rx.c:276-353).
    mix9 = (P(FACCH9), P(TCH9)) per frame.  The NT9 encoders are the oracle's (test data only)."""
    from importlib import import_module
    synth = import_module(pkg.__name__ + ".synth")
    rng = np.random.default_rng(seed)
    n = int(seconds * 23400 * sps)
    frame_len = 24 * 39 * sps
    t0 = int(rng.integers(0, frame_len))
    fn0 = int(rng.integers(0, 1 << 18))
    if kc is None:
        kc = np.zeros(8, np.uint8)
    fb, fd = pkg.api.burst_format("bcch"), pkg.api.burst_format("dc6")
    fs, ff = pkg.api.burst_format("nt3_speech"), pkg.api.burst_format("nt3_facch")
    f9 = pkg.api.burst_format("nt9")
    bcch, sent = synth.synth_bcch_carrier(fb, fd, n, sps, rng, stn=stn, delay=delay, fn0=fn0, t0=t0,
                                          esn0_db=esn0_db, cfo_hz=cfo_hz, imm_ass=[(k_ass, tn, p)])
    ia = [s for s in sent if s["type"] == "ccch" and s.get("imm_ass")]
    assert ia
    k_start = ia[0]["k"]
    tch, sent_t = synth.synth_tch3_carrier(fs, ff, n, sps, rng, t0=t0, fn0=fn0, k_start=k_start, tn=tn, p=p,
                                           kc=None, esn0_db=esn0_db, cfo_hz=cfo_hz, mix=(0.2, 0.2, 0.6),
                                           ass_cmd=(k_start + k_cmd, tn9))
    # CSD carrier
    sigma = np.sqrt(10.0 ** (-esn0_db / 10.0) / 2.0)
    csd = (rng.standard_normal((n, 2)) * sigma).astype(np.float32).view(np.complex64).reshape(-1)
    n_frames = (n - t0) // frame_len - 1
    il = orc.Interleaver()
    orc.lib().orc_interleaver_init(orc.C.byref(il), orc.C.c_int(3), orc.C.c_int(648))
    sent9 = []
    span = 5
    for k in range(k_start, n_frames):
        fn = fn0 + k
        u = rng.random()
        if u >= mix9[0] + mix9[1]:
            continue
        ciph = synth.a5_1(kc, [fn], 658)[0]
        sacch = rng.integers(0, 2, 10, dtype=np.uint8)
        status = rng.integers(0, 2, 4, dtype=np.uint8)
        if u < mix9[0]:
            l2 = rng.integers(0, 256, 38, dtype=np.uint8)
            l2[37] &= 0x0F
            e = orc.facch9_encode(l2, sacch, status, ciph)
            sid = 0
            kind = "facch9"
        else:
            l2 = rng.integers(0, 256, 60, dtype=np.uint8)
            e = np.zeros(662, np.uint8)
            orc.lib().orc_tch9_encode(e.ctypes.data_as(orc.C.c_void_p), l2.ctypes.data_as(orc.C.c_void_p),
                                      orc.C.c_int(2), sacch.ctypes.data_as(orc.C.c_void_p),
                                      status.ctypes.data_as(orc.C.c_void_p), ciph.ctypes.data_as(orc.C.c_void_p),
                                      orc.C.byref(il))
            sid = 1
            kind = "tch9"
        body = synth.shape_bursts(synth.map_symbols(f9, e[None, :], sync_id=sid), sps, 0.0, span)[0]
        pos = t0 + k * frame_len + tn9 * 39 * sps - span * sps
        if pos >= 0 and pos + body.size <= n:
            csd[pos:pos + body.size] += body
            sent9.append(dict(type=kind, fn=fn, k=k, l2=l2))
    if cfo_hz:
        ph = (2 * np.pi * cfo_hz / (23400 * sps)) * np.arange(n, dtype=np.float64)
        csd *= np.exp(1j * ph).astype(np.complex64)
    return bcch, tch, csd, kc, sent, sent_t, sent9
