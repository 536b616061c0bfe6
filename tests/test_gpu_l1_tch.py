"""GPU parity tests of the FACCH3 / TCH3 layer-1 decoders vs the CPU oracle (bit-exact)."""
import numpy as np
import pytest

import workloads

pytestmark = pytest.mark.gpu


def _soft(bits, amp=127):
    return (amp * (1 - 2 * bits.astype(np.int16))).astype(np.int8)


def _variants(rng, bits):
    clean = _soft(bits)
    noisy = np.clip(55 * (1 - 2 * bits.astype(np.int16)) + rng.normal(0, 45, bits.shape), -128, 127).astype(np.int8)
    noisy[rng.random(bits.shape) < 0.05] = 0
    junk = rng.integers(-128, 128, size=bits.shape).astype(np.int8)
    coarse = (rng.integers(-2, 3, size=bits.shape) * 50).astype(np.int8)
    return (("clean", clean), ("noisy", noisy), ("junk", junk), ("coarse", coarse))


def test_facch3_bit_exact(gpu_api, orc, pkg, decoder):
    rng = np.random.default_rng(31)
    n = 403                                     # ragged: not a multiple of 4 frames
    l2 = rng.integers(0, 256, (n, 10), dtype=np.uint8)
    l2[:, 9] &= 0x0F
    s = rng.integers(0, 2, (n, 32), dtype=np.uint8)
    e = pkg.synth.facch3_encode(l2, s).reshape(n, 416)
    for tag, eb in _variants(rng, e):
        g = gpu_api.facch3_decode_batch(eb)
        o = orc.facch3_decode(eb)
        assert np.array_equal(g[3], o[3]), f"{tag}: conv_rv"
        assert np.array_equal(g[2], o[2]), f"{tag}: crc"
        assert np.array_equal(g[0], o[0]), f"{tag}: l2"
        assert np.array_equal(g[1], o[1]), f"{tag}: status bits"
        if tag == "clean":
            assert np.array_equal(g[0], l2) and np.array_equal(g[1], s) and not g[2].any()
    # the reference's single call
    out, sb, rv, conv = gpu_api.facch3_decode(_soft(e[3]))
    assert rv == 0 and np.array_equal(out, l2[3]) and np.array_equal(sb, s[3]) and conv == 0


def test_tch3_bit_exact(gpu_api, orc, pkg, decoder):
    rng = np.random.default_rng(32)
    n = 301
    f0 = rng.integers(0, 256, (n, 10), dtype=np.uint8)
    f1 = rng.integers(0, 256, (n, 10), dtype=np.uint8)
    s = rng.integers(0, 2, (n, 4), dtype=np.uint8)
    for m in (0, 1):
        e = pkg.synth.tch3_encode(f0, f1, s, m)
        for tag, eb in _variants(rng, e):
            g = gpu_api.tch3_decode_batch(eb, m)
            o = orc.tch3_decode(eb, m)
            for idx, what in enumerate(("frame0", "frame1", "status", "conv0", "conv1")):
                assert np.array_equal(g[idx], o[idx]), f"m={m} {tag}: {what}"
            if tag == "clean":
                assert np.array_equal(g[0], f0) and np.array_equal(g[1], f1) and np.array_equal(g[2], s)
    a0, a1, sb, c0, c1 = gpu_api.tch3_decode(_soft(pkg.synth.tch3_encode(f0[:1], f1[:1], s[:1], 0)[0]), 0)
    assert np.array_equal(a0, f0[0]) and np.array_equal(a1, f1[0]) and np.array_equal(sb, s[0]) and c0 == 0 == c1


def test_nt3_mix_from_samples_matches_oracle(gpu_api, orc, pkg, decoder):
    """BASELINE configs[4] from samples (the bench's `--workload nt3`, small): 90 % NT3 speech + 10 % FACCH3 groups,
    window 474, the carrier offset handed over as freq_shift like rx_tch3 does.  Demodulated soft bits within 1 LSB of
    the oracle's, decoded frames / messages identical to the oracle's on the same soft bits, and what was sent comes
    back wherever the channel allows (speech class-1 bits; FACCH3 groups whose CRC passes)."""
    import workloads
    wl = workloads.nt3_mix(pkg, 800, seed=51)
    iq = wl["iq"].reshape(-1, wl["stride"])
    sp, fa = wl["speech"], wl["facch"]
    d = gpu_api.demod_batch("nt3_speech", wl["iq"], wl["offset"][sp], 474, sps=4, freq_shift=wl["freq_shift"][sp],
                            want_ssyms=False)
    assert not d["rv"].any()
    for k in range(0, sp.size, 7):
        r = orc.demod("nt3_speech", iq[sp[k], :474], 4, float(wl["freq_shift"][sp[k]]))
        assert np.max(np.abs(r["ebits"].astype(int) - d["ebits"][k].astype(int))) <= 1
        assert abs(r["toa"] - d["toa"][k]) < 0.02
    f0, f1, st, c0, c1 = gpu_api.tch3_decode_batch(d["ebits"], m=0)
    r0, r1, rs, rc0, rc1 = orc.tch3_decode(d["ebits"], 0)
    assert np.array_equal(f0, r0) and np.array_equal(f1, r1) and np.array_equal(st, rs)
    assert np.array_equal(c0, rc0) and np.array_equal(c1, rc1)
    hi = wl["esn0"][sp] >= 10.0 if "esn0" in wl else np.ones(sp.size, bool)
    ok = (f0[:, :6] == wl["frames"][:, 0, :6]).all(axis=1) & (f1[:, :6] == wl["frames"][:, 1, :6]).all(axis=1)
    assert ok[hi].mean() > 0.9

    df = gpu_api.demod_batch("nt3_facch", wl["iq"], wl["offset"][fa], 474, sps=4, freq_shift=wl["freq_shift"][fa],
                             want_ssyms=False)
    assert not df["rv"].any()
    for k in range(0, fa.size, 3):
        r = orc.demod("nt3_facch", iq[fa[k], :474], 4, float(wl["freq_shift"][fa[k]]))
        assert r["sync_id"] == df["sync_id"][k]
        assert np.max(np.abs(r["ebits"].astype(int) - df["ebits"][k].astype(int))) <= 1
    l2, bs, crc, conv = gpu_api.facch3_decode_batch(df["ebits"].reshape(-1, 416))
    rl2, rbs, rcrc, rconv = orc.facch3_decode(df["ebits"].reshape(-1, 416))
    assert np.array_equal(crc, rcrc) and np.array_equal(conv, rconv) and np.array_equal(l2, rl2) and np.array_equal(bs, rbs)
    good = crc == 0
    assert good.any() and np.array_equal(l2[good], wl["l2"][good])


def test_demod_large_batch_four_bursts_per_wave(gpu_api, orc, pkg):
    """Batches above 4096 bursts of a simple format take k_rx4g (four bursts per wave); smaller ones k_rx (one per wave).
    Both against the oracle, and against each other, on NT3 speech (one 6-symbol sync chunk, 7 lags) and DC6 (three
    chunks, 41 lags)."""
    import workloads
    wl = workloads.nt3_mix(pkg, 8000, seed=52)
    iq = wl["iq"].reshape(-1, wl["stride"])
    sp = wl["speech"]
    assert sp.size > 4096
    big = gpu_api.demod_batch("nt3_speech", wl["iq"], wl["offset"][sp], 474, sps=4, freq_shift=wl["freq_shift"][sp])
    small = gpu_api.demod_batch("nt3_speech", wl["iq"], wl["offset"][sp[:2000]], 474, sps=4, freq_shift=wl["freq_shift"][sp[:2000]])
    assert not big["rv"].any() and not (big["sync_id"] != 0).any()
    assert np.max(np.abs(big["ebits"][:2000].astype(int) - small["ebits"].astype(int))) <= 1
    assert np.max(np.abs(big["toa"][:2000] - small["toa"])) < 0.02
    assert np.max(np.abs(big["ssyms"][:2000] - small["ssyms"])) < 1e-4
    for k in list(range(0, sp.size, 97)) + [sp.size - 1, sp.size - 2, sp.size - 3]:
        r = orc.demod("nt3_speech", iq[sp[k], :474], 4, float(wl["freq_shift"][sp[k]]))
        assert r["rv"] == 0
        assert np.max(np.abs(r["ebits"].astype(int) - big["ebits"][k].astype(int))) <= 1, k
        assert abs(r["toa"] - big["toa"][k]) < 0.02
        assert np.max(np.abs(r["ssyms"] - big["ssyms"][k])) < 1e-4
    # a batch that is not a multiple of four, and one window of silence in it (no sync found: rv = -1, zeros out)
    m = 4099
    iq2 = wl["iq"].copy().reshape(-1, wl["stride"])
    iq2[sp[4098]] = 0
    odd = gpu_api.demod_batch("nt3_speech", iq2, wl["offset"][sp[:m]], 474, sps=4, freq_shift=wl["freq_shift"][sp[:m]])
    assert odd["rv"][4098] == -1 and not odd["ebits"][4098].any() and not odd["rv"][:4098].any()
    assert np.array_equal(odd["ebits"][:4098], big["ebits"][:4098])

    mix = workloads.bcch_ccch_mix(pkg, 7000, seed=53)
    cc = np.nonzero(mix["kind"] == 1)[0]
    assert cc.size > 4096
    d = gpu_api.demod_batch("dc6", mix["iq"], mix["offset"][cc], 976, sps=4)
    for k in range(0, cc.size, 211):
        o = int(mix["offset"][cc[k]])
        r = orc.demod("dc6", mix["iq"][o:o + 976], 4)
        assert r["rv"] == d["rv"][k] == 0
        assert np.max(np.abs(r["ebits"].astype(int) - d["ebits"][k].astype(int))) <= 1
        assert abs(r["toa"] - d["toa"][k]) < 0.02 and abs(r["freq_err"] - d["freq_err"][k]) < 1e-5


@pytest.mark.parametrize("name,win", [("dc2", 12), ("bcch", 80), ("nt3_speech", 6), ("nt3_facch", 6)])
def test_demod_four_per_wave_equals_one_per_wave(gpu_api, orc, pkg, name, win):
    """The same bursts through both demodulation kernels: 4097+ bursts in one call (k_rx4g) and in calls of <= 4096
    (k_rx): decisions identical (rv, sync_id), toa within a bisection step, soft symbols within 1e-4, soft bits within
    1 LSB except the rare midpoint symbols; and a sample against the oracle."""
    synth = pkg.synth
    rng = np.random.default_rng(len(name) + win)
    fmt = pkg.api.burst_format(name)
    info = gpu_api.burst_info(name)
    n = 4500
    eb = rng.integers(0, 2, (n, info.ebits), dtype=np.uint8)
    jit = max(0, min(8, win // 2 - 2))
    # formats with two training sequences (NT3 FACCH: the kernel's two-sequence variant): both in the batch
    sid = rng.integers(0, len(fmt.sync), n) if len(fmt.sync) > 1 else 0
    bb = synth.synth_windows(fmt, synth.map_symbols(fmt, eb, sync_id=sid), 4, win, rng, toa_jitter=jit, frac=True, cfo_hz_std=20.0,
                             esn0_db=rng.choice([8.0, 15.0], n), gain_db_std=3.0)
    in_len = bb.in_len
    off = np.arange(n, dtype=np.uint64) * np.uint64(bb.stride)
    big = gpu_api.demod_batch(name, bb.iq, off, in_len, sps=4)
    parts = [gpu_api.demod_batch(name, bb.iq, off[i:i + 2250], in_len, sps=4) for i in (0, 2250)]
    small = {k: np.concatenate([p[k] for p in parts]) for k in ("rv", "sync_id", "toa", "freq_err", "ebits", "ssyms")}
    assert np.array_equal(big["rv"], small["rv"]) and np.array_equal(big["sync_id"], small["sync_id"])
    assert not big["rv"].any()
    if len(fmt.sync) > 1:
        # the sequence that was sent is the one that is found (the second one is ranked on the sum of both correlations,
        # the reference's quirk: it wins whenever it is there, and mostly when it is not - what matters is GPU == oracle)
        assert (big["sync_id"][np.asarray(sid) == 1] == 1).all() and set(np.unique(big["sync_id"])) <= {0, 1}
    assert np.max(np.abs(big["toa"] - small["toa"])) <= 16 / 1024 + 1e-6
    same_pick = np.rint(big["toa"]) == np.rint(small["toa"])
    assert same_pick.mean() > 0.995
    assert np.max(np.abs(big["freq_err"] - small["freq_err"])) < 1e-5
    d = np.abs(big["ebits"].astype(int) - small["ebits"].astype(int))[same_pick]
    assert (d.max(axis=1) <= 1).mean() > 0.999
    ds = np.abs(big["ssyms"] - small["ssyms"])[same_pick]
    ds = np.minimum(ds, 2 ** info.nbits - ds)                    # soft symbols live on a circle of 2^nbits
    assert ds.max() < 1e-4
    # the payload bits come back (hard decisions) at these signal levels
    # (with two training sequences only where the second was sent: the reference's ranking takes the second one anyway,
    # and a burst sent with the first is then read against the wrong phase reference - DESIGN.md, reference quirks)
    sent_ok = np.asarray(sid) == len(fmt.sync) - 1 if len(fmt.sync) > 1 else np.ones(n, bool)
    assert ((big["ebits"][sent_ok] < 0) == eb[sent_ok].astype(bool)).mean() > 0.97
    for k in range(0, n, 450):
        r = orc.demod(name, bb.iq[k, :in_len], 4)
        assert r["sync_id"] == big["sync_id"][k]
        assert r["rv"] == 0 and abs(r["toa"] - big["toa"][k]) <= 16 / 1024 + 1e-6
        if np.rint(r["toa"]) == np.rint(big["toa"][k]):
            assert np.max(np.abs(r["ebits"].astype(int) - big["ebits"][k].astype(int))) <= 1


@pytest.mark.gpu
def test_tch3_rx_one_launch_equals_demod_then_decode(gpu_api, orc, pkg, decoder):
    """gmr1_hip_tch3_rx_batch (rx_tch3's burst step: demodulate, then decode) against the two separate calls: identical
    outputs for a batch large enough for the one-launch kernel (soft bits never leave LDS; not a multiple of four; one
    window of silence: rv = -1, zero soft bits, decoded all the same), for a small batch (two launches), with a cipher
    stream and in both multiplexing modes; and the frames against the oracle's decoder on the GPU's soft bits."""
    import workloads
    wl = workloads.nt3_mix(pkg, 8000, seed=61)
    sp = wl["speech"]
    iq = wl["iq"].copy().reshape(-1, wl["stride"])
    n = 4099
    assert sp.size >= n
    iq[sp[4098]] = 0
    off, fs = wl["offset"][sp[:n]], wl["freq_shift"][sp[:n]]
    rng = np.random.default_rng(7)
    ciph = rng.integers(0, 2, (n, 208), dtype=np.uint8)
    for m, cp, cnt in ((0, None, n), (1, ciph, n), (0, ciph, 1500), (1, None, 1500)):
        one = gpu_api.tch3_rx_batch(iq, off[:cnt], 474, sps=4, freq_shift=fs[:cnt], m=m, ciph=None if cp is None else cp[:cnt])
        d = gpu_api.demod_batch("nt3_speech", iq, off[:cnt], 474, sps=4, freq_shift=fs[:cnt], want_ssyms=False)
        f0, f1, st, c0, c1 = gpu_api.tch3_decode_batch(d["ebits"], m=m, ciph=None if cp is None else cp[:cnt])
        assert np.array_equal(one["rv"], d["rv"]) and np.array_equal(one["sync_id"], d["sync_id"])
        assert np.array_equal(one["toa"], d["toa"]) and np.array_equal(one["ebits"], d["ebits"])
        assert np.array_equal(one["frame0"], f0) and np.array_equal(one["frame1"], f1) and np.array_equal(one["bits_s"], st)
        assert np.array_equal(one["conv0"], c0) and np.array_equal(one["conv1"], c1)
        if cnt == n:
            assert one["rv"][4098] == -1 and not one["ebits"][4098].any() and not one["rv"][:4098].any()
        k = np.arange(0, cnt, 37)
        r0, r1, rs, rc0, rc1 = orc.tch3_decode(one["ebits"][k], m=m, ciph=None if cp is None else cp[k])
        assert np.array_equal(one["frame0"][k], r0) and np.array_equal(one["frame1"][k], r1) and np.array_equal(one["bits_s"][k], rs)
        assert np.array_equal(one["conv0"][k], rc0) and np.array_equal(one["conv1"][k], rc1)
    # without the soft-bit output
    lean = gpu_api.tch3_rx_batch(iq, off, 474, sps=4, freq_shift=fs, m=0, want_ebits=False)
    full = gpu_api.tch3_rx_batch(iq, off, 474, sps=4, freq_shift=fs, m=0)
    assert np.array_equal(lean["frame0"], full["frame0"]) and np.array_equal(lean["frame1"], full["frame1"])


def test_facch3_pass_rate_is_the_reference_sync_ranking_not_the_decoder(gpu_api, orc, pkg):
    """`bench.py --workload nt3` reports that only ~59 % of the FACCH3 groups pass their CRC, at any SNR.  That is the
    reference's sync search, reproduced on purpose: _sync_find zeroes its correlation accumulator once and never between
    training sequences (pi4cxpsk.c:206-237), so for the two-sequence NT3 FACCH format the second sequence is ranked on
    |c0| + |c1| and always wins -- every burst SENT with sequence 0 is demodulated against the wrong phase reference.
    The genie check that separates this from a bug shared by the oracle and the product: with the SENT sequence as the
    format's only one (a caller-defined burst description) nearly every group passes at 20 dB, on both sides; with the
    reference's ranking both sides pick sequence 1 throughout and lose the groups sent with sequence 0."""
    import ctypes as C
    import oracle_lib
    wl = workloads.nt3_mix(pkg, n=3200, seed=77, esn0_db=(20.0,))
    fac = wl["facch"].reshape(-1, 4)
    ng = fac.shape[0]
    in_len, sps = wl["in_len"], 4
    iq = wl["iq"].reshape(-1, wl["stride"])[:, :in_len]

    def caller_format(sid):                      # the product's: struct gmr1_pi4cxpsk_burst with one training sequence
        c = gpu_api.CallerBurst("nt3_facch")
        if sid:
            c.burst.sync[0] = c.burst.sync[1]
        c.burst.sync[1] = type(c.burst.sync[1])()
        return c

    def oracle_format(sid):                      # the oracle's: struct orc_burst with one training sequence
        b = oracle_lib.Burst()
        C.memmove(C.byref(b), oracle_lib.burst("nt3_facch"), C.sizeof(b))
        if sid:
            C.memmove(C.byref(b.sync[0]), C.byref(b.sync[1]), C.sizeof(b.sync[0]))
            b.n_sync_chunks[0] = b.n_sync_chunks[1]
        b.n_sync = 1
        return C.pointer(b)

    cf, of = [caller_format(0), caller_format(1)], [oracle_format(0), oracle_format(1)]
    eb = {k: np.zeros((ng, 4, 104), np.int8) for k in ("g_ref", "g_genie", "o_ref", "o_genie")}
    sid_g, sid_o = np.zeros((ng, 4), int), np.zeros((ng, 4), int)
    for g in range(ng):
        sent = int(wl["sync_id"][g])
        for q in range(4):
            i = fac[g, q]
            fsh = float(wl["freq_shift"][i])
            a = gpu_api.pi4cxpsk_demod("nt3_facch", iq[i], sps, fsh)
            b = gpu_api.pi4cxpsk_demod(cf[sent], iq[i], sps, fsh)
            c = orc.demod("nt3_facch", iq[i], sps, fsh)
            d = orc.demod(of[sent], iq[i], sps, fsh)
            assert a["rv"] == b["rv"] == c["rv"] == d["rv"] == 0
            eb["g_ref"][g, q], eb["g_genie"][g, q], eb["o_ref"][g, q], eb["o_genie"][g, q] = a["ebits"], b["ebits"], c["ebits"], d["ebits"]
            sid_g[g, q], sid_o[g, q] = a["sync_id"], c["sync_id"]
    crc = {k: (gpu_api.facch3_decode_batch(v) if k[0] == "g" else orc.facch3_decode(v)) for k, v in eb.items()}
    rate = {k: float((v[2] == 0).mean()) for k, v in crc.items()}
    print("FACCH3 groups passing their CRC at 20 dB:", rate, "; sequence 1 picked on", float((sid_g == 1).mean()), "of the bursts")
    # the reference's ranking: sequence 1 nearly always, so about half the groups (those sent with sequence 0) are lost
    assert np.array_equal(sid_g, sid_o)
    assert (sid_g == 1).mean() > 0.9
    sent0 = wl["sync_id"][:ng] == 0
    assert (crc["g_ref"][2][sent0] == 0).mean() < 0.5 and (crc["g_ref"][2][~sent0] == 0).mean() > 0.95
    assert 0.4 < rate["g_ref"] < 0.75 and abs(rate["g_ref"] - rate["o_ref"]) < 0.05
    # the sent sequence forced: the algorithm itself loses next to nothing
    assert rate["g_genie"] > 0.97 and rate["o_genie"] > 0.97
    good = crc["g_genie"][2] == 0
    assert np.array_equal(crc["g_genie"][0][good], wl["l2"][:ng][good])
