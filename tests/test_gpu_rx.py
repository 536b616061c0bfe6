"""GPU parity tests (run on an MI355X): HIP path vs the CPU oracle, through the C ABI."""
import numpy as np
import pytest

import workloads

pytestmark = pytest.mark.gpu


def sps_of_label(label):
    return int(label[3:]) if label.startswith("sps") else 4


def _compare_fused(got, ref, wl, *, label):
    n = wl["kind"].size
    assert np.array_equal(got["rv"], ref["rv"]), f"{label}: rv differs"
    # --- float stage: soft symbols within 1e-4 (north_star tolerance) except where the integer
    #     sample pick round(toa) legitimately flipped because toa sits on an x.5 boundary
    dtoa = np.abs(got["toa"] - ref["toa"])
    flip = np.round(got["toa"]) != np.round(ref["toa"])
    # early/late bisection can take the other branch on a near tie, so toa may also differ by
    # a few steps of 1/1024 without flipping the pick; both must stay rare and small
    assert dtoa.max() < 16.0 / 1024.0, f"{label}: toa differs by {dtoa.max()}"
    # (DESIGN.md section 6 measures 0 ... 0.25 % over sps 2 ... 16: twice that is the bound, so a regression shows)
    assert flip.mean() <= 0.005, f"{label}: {flip.sum()} / {n} sample-pick flips"
    # at >= 4 samples per symbol everything behind the timing depends on the pick round(toa) alone (pi4cxpsk.c:292-295):
    # soft symbols, frequency error and soft bits are compared on EVERY burst whose pick did not flip; bursts whose toa
    # differs by a bisection step or two without flipping must be rare
    near = (dtoa != 0) & ~flip
    print(f"{label}: toa differs without a flip on {near.mean():.4f} of the bursts, picks flipped on {flip.mean():.4f}")
    assert near.mean() < 0.005, f"{label}: toa differs on {near.mean():.4f} of the bursts"
    same = ~flip
    dss = np.abs(got["ssyms"][same] - ref["ssyms"][same])
    # phase wraps at +-2 (QPSK soft symbol range): compare modulo 4
    dss = np.minimum(dss, np.abs(dss - 4.0))
    assert dss.max() < 1e-4, f"{label}: soft symbols differ by {dss.max()}"
    dfe = np.abs(got["freq_err"][same] - ref["freq_err"][same])
    assert dfe.max() < 1e-5, f"{label}: freq_err differs by {dfe.max()}"
    # --- soft bits: identical or +-1 LSB, and rarely
    deb = np.abs(got["ebits"][same].astype(int) - ref["ebits"][same].astype(int))
    # ... except for a symbol whose phase sits on the midpoint between two constellation points within the soft-symbol
    # tolerance: nearest point and neighbour swap roles, which flips the sign of the weakest possible soft bit
    # (|value| = 63) and nothing else.  Those are identified one by one through the oracle's soft symbol, and counted.
    n_mid = 0
    if deb.max() > 1:
        ge, re_, rs = got["ebits"][same], ref["ebits"][same], ref["ssyms"][same]
        kinds = wl["kind"][same]
        from __graft_entry__ import load_package
        fmts = [load_package().api.burst_format("bcch"), load_package().api.burst_format("dc6")]
        pos = [np.concatenate([np.arange(p, p + l) for p, l in f.data]) for f in fmts]
        for b, e in zip(*np.nonzero(deb > 1)):
            sym = pos[kinds[b]][e // 2]
            frac = abs(abs(float(rs[b, sym])) % 1.0 - 0.5)
            assert frac < 1e-4, f"{label}: burst {b} bit {e}: {ge[b, e]} vs {re_[b, e]}, soft symbol {rs[b, sym]}"
            assert abs(abs(int(ge[b, e])) - 63) <= 1 and abs(abs(int(re_[b, e])) - 63) <= 1, (label, b, e)
            n_mid += 1
        assert n_mid <= max(1, int(2e-5 * deb.size)), f"{label}: {n_mid} midpoint symbols in {deb.size} soft bits"
    assert ((deb != 0) & (deb <= 1)).mean() < 1e-3, f"{label}: {(deb != 0).mean():.2e} of soft bits differ"
    # --- integer chain: wherever the soft bits are identical everything downstream is bit-exact
    # (a burst's ~430 soft bits are all identical for about 85 % of the bursts: the phase of a late symbol is an fp32
    # number of ~30 turns on both sides, i.e. quantised to ~1e-3 of a soft-bit step, and the two sides round it in
    # different places, so about 4e-4 of the soft bits sit on a step's edge and come out 1 LSB apart)
    eq = same & np.all(got["ebits"] == ref["ebits"], axis=1)
    # documented: 85 % at sps 4; the bound is that minus three standard deviations of the sample
    want = 0.85 if sps_of_label(label) == 4 else 0.80
    assert eq.mean() > want - 3.0 * np.sqrt(want * (1 - want) / n), f"{label}: only {eq.mean():.3f} of the bursts have all soft bits identical"
    assert np.array_equal(got["l2"][eq], ref["l2"][eq]), f"{label}: L2 differs on identical soft bits"
    assert np.array_equal(got["crc"][eq], ref["crc"][eq])
    assert np.array_equal(got["conv"][eq], ref["conv"][eq])
    # --- payloads: every burst whose CRC passes on either side decodes to the same bytes
    ok = (got["crc"] == 0) | (ref["crc"] == 0)
    assert np.array_equal(got["crc"][ok], ref["crc"][ok]), f"{label}: CRC verdicts differ"
    assert np.array_equal(got["l2"][ok], ref["l2"][ok]), f"{label}: decoded payloads differ"
    good = got["crc"] == 0
    assert np.array_equal(got["l2"][good], wl["l2"][good]), f"{label}: payload is not what was sent"
    return dict(flips=int(flip.sum()), dss=float(dss.max()), eb_diff=float((deb != 0).mean()),
                crc_fail=int((got["crc"] != 0).sum()), all_soft_bits_identical=float(eq.mean()))


def test_l1_bcch_ccch_bit_exact(gpu_api, orc, pkg, decoder):
    """Hard-decision l1 chain: same soft bits in -> identical L2 / crc / conv (bit-exact)."""
    rng = np.random.default_rng(11)
    n = 1003   # not a multiple of 4: exercises the ragged last wavefront
    l2 = rng.integers(0, 256, size=(n, 24), dtype=np.uint8)
    for name, enc, dec_g, dec_o, neb in (
            ("bcch", pkg.synth.bcch_encode, gpu_api.bcch_decode_batch, orc.bcch_decode, 424),
            ("ccch", pkg.synth.ccch_encode, gpu_api.ccch_decode_batch, orc.ccch_decode, 432)):
        bits = enc(l2).astype(np.int16)
        clean = (127 * (1 - 2 * bits)).astype(np.int8)
        # (a) clean, (b) noisy soft values incl. erasures and -128, (c) pure noise (CRC fails, ties)
        noisy = np.clip(60 * (1 - 2 * bits) + rng.normal(0, 50, bits.shape), -128, 127).astype(np.int8)
        noisy[rng.random(bits.shape) < 0.05] = 0
        junk = rng.integers(-128, 128, size=bits.shape).astype(np.int8)
        coarse = (rng.integers(-2, 3, size=bits.shape) * 50).astype(np.int8)   # many exact metric ties
        for tag, eb in (("clean", clean), ("noisy", noisy), ("junk", junk), ("coarse", coarse)):
            g = dec_g(eb)
            o = dec_o(eb)
            assert np.array_equal(g[2], o[2]), f"{name}/{tag}: conv_rv differs"
            assert np.array_equal(g[1], o[1]), f"{name}/{tag}: crc differs"
            assert np.array_equal(g[0], o[0]), f"{name}/{tag}: L2 differs"
            if tag == "clean":
                assert np.array_equal(g[0], l2) and not g[1].any()


def test_l1_legacy_single_call(gpu_api, orc, pkg, decoder):
    """gmr1_bcch_decode / gmr1_ccch_decode: the reference's own one-burst calls."""
    rng = np.random.default_rng(12)
    l2 = rng.integers(0, 256, size=(3, 24), dtype=np.uint8)
    for i in range(3):
        eb = (100 * (1 - 2 * pkg.synth.bcch_encode(l2[i:i + 1])[0].astype(np.int16))).astype(np.int8)
        out, crc, conv = gpu_api.bcch_decode(eb)
        o = orc.bcch_decode(eb[None])
        assert crc == 0 and np.array_equal(out, l2[i]) and conv == o[2][0]
        eb = (100 * (1 - 2 * pkg.synth.ccch_encode(l2[i:i + 1])[0].astype(np.int16))).astype(np.int8)
        eb[7] = -eb[7]
        out, crc, conv = gpu_api.ccch_decode(eb)
        o = orc.ccch_decode(eb[None])
        assert crc == 0 and np.array_equal(out, l2[i]) and conv == o[2][0]


def test_fused_rx_parity_small(gpu_api, orc, pkg, decoder):
    wl = workloads.bcch_ccch_mix(pkg, n=2001, seed=3)
    got = gpu_api.rx_bcch_ccch_batch(wl["iq"], wl["offset"], wl["kind"], sps=4)
    ref = orc.demod_decode_batch(wl["iq"], wl["offset"], wl["kind"], sps=4)
    st = _compare_fused(got, ref, wl, label="mix")
    print("fused parity:", st)
    assert (got["crc"] == 0).mean() > 0.9


def test_fused_rx_bursts_at_the_window_edges(gpu_api, orc, pkg):
    """Bursts anywhere in the search window, its first and last lags included (the estimator itself pulls the outermost
    ones a lag inward: picks d = round(toa) of 0 ... 39 in the CCCH window's 41 lags): the fourth burst of a wave -- whose
    window the kernel keeps in registers and hands to pass 2 through LDS instead of reading it again -- must see exactly
    what a re-read would have at either end.  The oracle is the checker, burst by burst."""
    wl = workloads.bcch_ccch_mix(pkg, n=6001, seed=41, esn0_db=(8.0, 20.0), toa_jitter=20)
    got = gpu_api.rx_bcch_ccch_batch(wl["iq"], wl["offset"], wl["kind"], sps=4)
    ref = orc.demod_decode_batch(wl["iq"], wl["offset"], wl["kind"], sps=4)
    assert np.array_equal(got["rv"], ref["rv"])
    d_ref = np.round(ref["toa"]).astype(int)
    fourth = (np.arange(wl["kind"].size) % 4) == 3
    # the edge picks occur, on fourth bursts too
    ccch = wl["kind"] == 1
    assert (d_ref[ccch & fourth] <= 1).sum() > 10 and (d_ref[ccch & fourth] >= 38).sum() > 10 and d_ref.min() == 0
    dtoa = np.abs(got["toa"] - ref["toa"])
    flip = np.round(got["toa"]) != np.round(ref["toa"])
    assert dtoa.max() < 16.0 / 1024.0 and flip.mean() <= 0.01
    near = (dtoa != 0) & ~flip
    print(f"window edges: toa differs without a flip on {near.mean():.4f} of the bursts, picks flipped on {flip.mean():.4f}")
    assert near.mean() < 0.01
    same = ~flip
    dss = np.abs(got["ssyms"][same] - ref["ssyms"][same])
    dss = np.minimum(dss, np.abs(dss - 4.0))
    assert dss.max() < 1e-4
    deb = np.abs(got["ebits"][same].astype(int) - ref["ebits"][same].astype(int))
    assert (deb > 1).mean() < 2e-5 and ((deb != 0)).mean() < 1e-3
    eq = same & np.all(got["ebits"] == ref["ebits"], axis=1)
    assert eq.mean() > 0.75
    for k in ("l2", "crc", "conv"):
        assert np.array_equal(got[k][eq], ref[k][eq]), k
    # and the bursts at the edges are in that set in proportion (nothing edge-specific hides in the excluded ones)
    edge = (d_ref <= 1) | (ccch & (d_ref >= 38))
    assert eq[edge & fourth].mean() > 0.6
    ok = (got["crc"] == 0) | (ref["crc"] == 0)
    assert np.array_equal(got["l2"][ok], ref["l2"][ok]) and np.array_equal(got["crc"][ok], ref["crc"][ok])


def test_fused_rx_clean_exact_payload(gpu_api, orc, pkg, decoder):
    """Config 1 flavour: noiseless bursts, integer TOA -> every payload recovered, toa == 40 / 20."""
    wl = workloads.bcch_ccch_mix(pkg, n=70, seed=1, esn0_db=(200.0,), toa_jitter=0, frac=False,
                                 cfo_hz_std=0.0, gain_db_std=0.0)
    got = gpu_api.rx_bcch_ccch_batch(wl["iq"], wl["offset"], wl["kind"], sps=4)
    ref = orc.demod_decode_batch(wl["iq"], wl["offset"], wl["kind"], sps=4)
    assert not got["crc"].any() and not got["rv"].any()
    assert np.array_equal(got["l2"], wl["l2"])
    assert np.array_equal(got["l2"], ref["l2"])
    assert np.array_equal(np.round(got["toa"]), np.where(wl["kind"] == 0, 40, 20))
    assert np.array_equal(got["conv"], ref["conv"])


def test_fused_rx_degenerate_inputs(gpu_api, orc, pkg):
    """All-zero and constant windows: the reference finds no sync (rv = -1); ragged n."""
    n = 5
    kind = np.array([0, 1, 0, 1, 1], np.uint8)
    offset = (np.arange(n) * 1024).astype(np.uint64)
    iq = np.zeros(n * 1024, np.complex64)
    iq[1024:2048] = 3.0 + 1.0j          # constant -> zero variance
    wl = workloads.bcch_ccch_mix(pkg, n=7, seed=5)
    iq[2048:2048 + 1016] = wl["iq"][:1016]          # one real BCCH burst in slot 2
    got = gpu_api.rx_bcch_ccch_batch(iq, offset, kind, sps=4)
    ref = orc.demod_decode_batch(iq, offset, kind, sps=4)
    assert list(ref["rv"]) == [-1, -1, 0, -1, -1]
    assert np.array_equal(got["rv"], ref["rv"])
    assert np.array_equal(got["crc"], ref["crc"])
    assert np.array_equal(got["l2"], ref["l2"])
    assert np.array_equal(got["l2"][2], wl["l2"][0])


def test_demod_generic_burst_types(gpu_api, orc, pkg):
    """Demod-only path on other burst formats (multi-sync-sequence, BPSK, long)."""
    rng = np.random.default_rng(21)
    sps = 4
    for name, win in (("nt3_speech", 6), ("nt3_facch", 6), ("dc2", 40), ("nt6", 24), ("nt9", 24),
                      ("sdcch", 24), ("dc12", 40), ("rach", 40)):
        fmt = pkg.api.burst_format(name)
        n = 24
        ebits = rng.integers(0, 2, size=(n, fmt.ebits), dtype=np.uint8)
        sid = rng.integers(0, len(fmt.sync), size=n)
        sym = pkg.synth.map_symbols(fmt, ebits, sync_id=sid)
        bb = pkg.synth.synth_windows(fmt, sym, sps, win, rng, toa_jitter=min(2, win // 4), frac=True,
                                     cfo_hz_std=20.0, esn0_db=15.0)
        in_len = bb.in_len
        offset = (np.arange(n) * bb.stride).astype(np.uint64)
        got = gpu_api.demod_batch(name, bb.iq, offset, in_len, sps=sps)
        hard = (got["ebits"] < 0).astype(np.uint8)
        for i in range(n):
            o = orc.demod(name, bb.iq[i, :in_len], sps)
            assert got["rv"][i] == o["rv"] == 0
            assert got["sync_id"][i] == o["sync_id"], name
            assert abs(got["toa"][i] - o["toa"]) < 16 / 1024, name
            if round(float(got["toa"][i])) == round(o["toa"]) and got["toa"][i] == o["toa"]:
                d = np.abs(got["ssyms"][i] - o["ssyms"])
                span = 2.0 ** fmt.nbits
                d = np.minimum(d, np.abs(d - span))
                assert d.max() < 1e-4, (name, d.max())
                assert np.abs(got["ebits"][i].astype(int) - o["ebits"].astype(int)).max() <= 1
        # Reference quirk (pi4cxpsk.c:207-237): the correlation accumulator is not
        # cleared between sync sequences, so later sequences are ranked (and timed) on the sum of all
        # earlier correlations.  It is reproduced for parity (checked against the oracle above); the
        # sent bits are therefore only guaranteed to come back for single-sequence formats.
        if len(fmt.sync) == 1:
            assert (hard != ebits).mean() < 0.02, name


def test_legacy_pi4cxpsk_demod_call(gpu_api, orc, pkg):
    """gmr1_pi4cxpsk_demod(&gmr1_bcch_burst, cxvec, ...) through the exported struct address."""
    wl = workloads.bcch_ccch_mix(pkg, n=7, seed=9, esn0_db=(12.0,))
    for i in range(7):
        name = "dc6" if wl["kind"][i] else "bcch"
        a = int(wl["offset"][i])
        iq = wl["iq"][a:a + wl["in_len"][int(wl["kind"][i])]]
        g = gpu_api.pi4cxpsk_demod(name, iq, 4, 0.0)
        o = orc.demod(name, iq, 4, 0.0)
        assert g["rv"] == o["rv"] == 0
        assert g["sync_id"] == o["sync_id"]
        assert abs(g["toa"] - o["toa"]) < 16 / 1024
        if g["toa"] == o["toa"]:
            assert np.abs(g["ebits"].astype(int) - o["ebits"].astype(int)).max() <= 1
            assert abs(g["freq_err"] - o["freq_err"]) < 1e-5


def test_detect_and_mod_order(gpu_api, orc, pkg):
    """gmr1_pi4cxpsk_detect between the two NT3 burst types (rx_tch3, gmr1_rx.c:584) and
    gmr1_pi4cxpsk_mod_order, batch and single-call forms, vs the oracle."""
    rng = np.random.default_rng(41)
    sps, win, n = 4, 6, 60
    f_sp, f_fa = pkg.api.burst_format("nt3_speech"), pkg.api.burst_format("nt3_facch")
    in_len = 117 * sps + win
    iq = np.zeros((n, in_len), np.complex64)
    truth = np.zeros(n, int)
    for i in range(n):
        which = int(rng.integers(0, 2))          # 0 = facch (first in the list), 1 = speech
        fmt = f_sp if which else f_fa
        bits = rng.integers(0, 2, (1, fmt.ebits), dtype=np.uint8)
        sym = pkg.synth.map_symbols(fmt, bits, sync_id=int(rng.integers(0, len(fmt.sync))))
        bb = pkg.synth.synth_windows(fmt, sym, sps, win, rng, toa_jitter=1, frac=True, cfo_hz_std=30.0,
                                     esn0_db=float(rng.choice([8.0, 15.0])))
        iq[i] = bb.iq[0, :in_len]
        truth[i] = which
    offset = (np.arange(n) * in_len).astype(np.uint64)
    for e_toa in (3.0, None):
        got = gpu_api.detect_batch(["nt3_facch", "nt3_speech"], iq, offset, in_len, sps=sps, e_toa=e_toa)
        for i in range(n):
            o = orc.detect(["nt3_facch", "nt3_speech"], -1.0 if e_toa is None else e_toa, iq[i], sps)
            assert got["rv"][i] == o["rv"] == 0
            assert got["bt_id"][i] == o["bt_id"], (i, e_toa)
            assert got["sync_id"][i] == o["sync_id"], (i, e_toa)
            assert abs(got["toa"][i] - o["toa"]) < 16 / 1024
    assert (got["bt_id"] == truth).mean() > 0.9
    d = gpu_api.pi4cxpsk_detect(["nt3_facch", "nt3_speech"], 3.0, iq[0], sps)
    o = orc.detect(["nt3_facch", "nt3_speech"], 3.0, iq[0], sps)
    assert d["rv"] == 0 and d["bt_id"] == o["bt_id"] and d["sync_id"] == o["sync_id"]
    # caller-defined burst descriptions (the reference takes any struct gmr1_pi4cxpsk_burst **, pi4cxpsk.c:617-682):
    # copies of the two formats in the caller's memory, alone and mixed with a built-in one, decide like the built-ins;
    # one with a training symbol changed is a different format and loses against the right one
    c_fa, c_sp = gpu_api.CallerBurst("nt3_facch"), gpu_api.CallerBurst("nt3_speech")
    for i in range(0, n, 7):
        o = orc.detect(["nt3_facch", "nt3_speech"], 3.0, iq[i], sps)
        for cand in ([c_fa, c_sp], [c_fa, "nt3_speech"], ["nt3_facch", c_sp]):
            d = gpu_api.pi4cxpsk_detect(cand, 3.0, iq[i], sps)
            assert (d["rv"], d["bt_id"], d["sync_id"]) == (0, o["bt_id"], o["sync_id"]), (i, cand)
            assert abs(d["toa"] - o["toa"]) < 16 / 1024
    i_sp = int(np.nonzero(truth == 1)[0][0])
    odd = gpu_api.CallerBurst("nt3_speech")
    for k in range(1, odd.burst.sync[0][0].len, 2):
        odd.burst.sync[0][0].syms[k] ^= 2            # every other training symbol turned by 180 degrees
    assert gpu_api.pi4cxpsk_detect([odd, c_sp], 3.0, iq[i_sp], sps)["bt_id"] == 1
    assert gpu_api.pi4cxpsk_detect([c_sp, odd], 3.0, iq[i_sp], sps)["bt_id"] == 0
    # lists longer than four (the reference takes any NULL-terminated list): batch form with built-in ids, the
    # reference's call with built-in and caller-defined descriptions mixed; against the oracle on the same list, and the
    # winner's position is the one in the caller's list
    long_list = ["nt3_facch", "nt3_speech", "nt3_facch", "nt3_speech", "nt3_speech", "nt3_facch", "nt3_speech"]
    got = gpu_api.detect_batch(long_list, iq, offset, in_len, sps=sps, e_toa=3.0)
    for i in range(n):
        o = orc.detect(long_list, 3.0, iq[i], sps)
        assert (got["rv"][i], got["bt_id"][i], got["sync_id"][i]) == (0, o["bt_id"], o["sync_id"]), i
        assert got["bt_id"][i] in (0, 1)                 # a later copy never beats the first (strict >)
    odd_first = [odd, odd, odd, odd, odd, c_sp, "nt3_facch"]
    assert gpu_api.pi4cxpsk_detect(odd_first, 3.0, iq[i_sp], sps)["bt_id"] == 5
    mixed = [c_fa, odd, "nt3_facch", odd, c_sp, "nt3_speech"]
    for i in range(0, n, 9):
        o = orc.detect(["nt3_facch", "nt3_speech"], 3.0, iq[i], sps)
        d = gpu_api.pi4cxpsk_detect(mixed, 3.0, iq[i], sps)
        assert d["rv"] == 0 and d["bt_id"] == (0 if o["bt_id"] == 0 else 4) and d["sync_id"] == o["sync_id"], (i, d, o)
    order = gpu_api.mod_order_batch(iq, offset, in_len, sps=sps)
    for i in range(n):
        assert order[i] == orc.mod_order(iq[i], sps), i
    assert (order == np.where(truth == 1, 4, 2)).mean() >= 0.8      # the estimator itself errs at 8 dB
    assert gpu_api.pi4cxpsk_mod_order(iq[1], sps) == order[1]


def test_fused_rx_full_size_properties(gpu_api, orc, pkg):
    """BASELINE.json configs[2] at full size (100 000 bursts): size-independent properties --
    every CRC-passing burst returns exactly the payload that was sent, the pass rate is what the
    channel allows, two runs are bit-identical, and a strided sample agrees with the oracle."""
    n = 100_000
    wl = workloads.bcch_ccch_mix(pkg, n=n, seed=3)
    a = gpu_api.rx_bcch_ccch_batch(wl["iq"], wl["offset"], wl["kind"], sps=4, want_ebits=False, want_ssyms=False)
    b = gpu_api.rx_bcch_ccch_batch(wl["iq"], wl["offset"], wl["kind"], sps=4, want_ebits=False, want_ssyms=False)
    for k in ("l2", "crc", "conv", "toa", "freq_err", "rv"):
        assert np.array_equal(a[k], b[k]), f"run-to-run difference in {k}"
    good = a["crc"] == 0
    assert not a["rv"].any()
    assert good.mean() > 0.95
    assert np.array_equal(a["l2"][good], wl["l2"][good])
    assert np.abs(a["toa"][good] - wl["toa"][good]).max() < 2.5       # TOA estimate tracks the truth
    # a checksum of checksums over the payloads, against the generator's
    assert int(a["l2"][good].astype(np.uint64).sum()) == int(wl["l2"][good].astype(np.uint64).sum())
    # strided sample vs the oracle (CRC verdicts and payloads)
    idx = np.arange(0, n, 53)
    sub_off = wl["offset"][idx]
    ref = orc.demod_decode_batch(wl["iq"], sub_off, wl["kind"][idx], sps=4, want_ebits=False, want_ssyms=False)
    assert np.array_equal(ref["crc"], a["crc"][idx])
    ok = ref["crc"] == 0
    assert np.array_equal(ref["l2"][ok], a["l2"][idx][ok])


def test_demod_low_oversampling(gpu_api, orc, pkg):
    """sps < 4: the reference's sinc fractional-delay branch (pi4cxpsk.c:298-343), and sps 1."""
    rng = np.random.default_rng(51)
    for sps, name, win in ((2, "bcch", 20), (2, "nt3_speech", 4), (3, "dc6", 12), (1, "bcch", 6), (8, "dc2", 32),
                           (12, "bcch", 48), (16, "dc6", 64)):          # windows beyond 2048 samples: the 64-samples-per-lane body
        fmt = pkg.api.burst_format(name)
        n = 16
        ebits = rng.integers(0, 2, size=(n, fmt.ebits), dtype=np.uint8)
        sym = pkg.synth.map_symbols(fmt, ebits)
        bb = pkg.synth.synth_windows(fmt, sym, sps, win, rng, toa_jitter=1 if win >= 8 else 0, frac=sps > 1,
                                     cfo_hz_std=20.0, esn0_db=18.0)
        offset = (np.arange(n) * bb.stride).astype(np.uint64)
        got = gpu_api.demod_batch(name, bb.iq, offset, bb.in_len, sps=sps)
        nfrac = 0
        for i in range(n):
            o = orc.demod(name, bb.iq[i, :bb.in_len], sps)
            assert got["rv"][i] == o["rv"] == 0, (sps, name)
            assert abs(got["toa"][i] - o["toa"]) < 16 / 1024, (sps, name)
            if got["toa"][i] == o["toa"]:
                dss = np.abs(got["ssyms"][i] - o["ssyms"])
                span = 2.0 ** fmt.nbits
                dss = np.minimum(dss, np.abs(dss - span))
                assert dss.max() < 1e-4, (sps, name, dss.max())
                assert np.abs(got["ebits"][i].astype(int) - o["ebits"].astype(int)).max() <= 1
                nfrac += abs(o["toa"] - round(o["toa"])) > 0.1
        if sps in (2, 3):
            assert nfrac >= 3, "the fractional-delay branch was not exercised"
        hard = (got["ebits"] < 0).astype(np.uint8)
        assert (hard != ebits).mean() < 0.03, (sps, name)


def test_demod_and_detect_wide_search_windows(gpu_api, orc, pkg):
    """Search windows beyond 256 lags (the reference has no limit, pi4cxpsk.c:184-201): the correlation accumulator is sized
    by the call.  Demodulation of DC2 with 400 lags and NT3 speech with 700, and detection between the two NT3 formats."""
    rng = np.random.default_rng(77)
    for name, win in (("dc2", 400), ("nt3_speech", 700)):
        fmt = pkg.api.burst_format(name)
        n = 12
        ebits = rng.integers(0, 2, size=(n, fmt.ebits), dtype=np.uint8)
        bb = pkg.synth.synth_windows(fmt, pkg.synth.map_symbols(fmt, ebits), 4, win, rng, toa_jitter=win // 3, frac=True,
                                     cfo_hz_std=20.0, esn0_db=20.0)
        assert bb.in_len - fmt.length * 4 + 1 > 256
        offset = (np.arange(n) * bb.stride).astype(np.uint64)
        got = gpu_api.demod_batch(name, bb.iq, offset, bb.in_len, sps=4)
        for i in range(n):
            o = orc.demod(name, bb.iq[i, :bb.in_len], 4)
            assert got["rv"][i] == o["rv"] == 0, name
            assert abs(got["toa"][i] - o["toa"]) < 16 / 1024, name
            if got["toa"][i] == o["toa"]:
                assert np.abs(got["ebits"][i].astype(int) - o["ebits"].astype(int)).max() <= 1
        assert ((got["ebits"] < 0).astype(np.uint8) != ebits).mean() < 0.02
        if name == "nt3_speech":
            d = gpu_api.detect_batch(["nt3_speech", "nt3_facch"], bb.iq.reshape(-1), offset, bb.in_len, sps=4)
            for i in range(n):
                o = orc.detect(["nt3_speech", "nt3_facch"], -1.0, bb.iq[i, :bb.in_len], 4)
                assert d["rv"][i] == o["rv"] and d["bt_id"][i] == o["bt_id"] and d["sync_id"][i] == o["sync_id"]


@pytest.mark.parametrize("sps", [2, 3, 5, 8, 10, 16])
def test_fused_rx_other_oversampling(gpu_api, orc, pkg, sps, decoder):
    """The fused path at sps != 4 (generic k_rx4 instantiation; windows of 234 sps + 20 sps / 10 sps samples; below 4 and
    above 8 samples per symbol the one-burst-at-a-time body, whose windows go up to 4 096 samples)."""
    wl = workloads.bcch_ccch_mix(pkg, n=403, seed=7, sps=sps, toa_jitter=2 * sps)
    got = gpu_api.rx_bcch_ccch_batch(wl["iq"], wl["offset"], wl["kind"], sps=sps)
    ref = orc.demod_decode_batch(wl["iq"], wl["offset"], wl["kind"], sps=sps)
    st = _compare_fused(got, ref, wl, label=f"sps{sps}")
    print("fused parity:", st)
    assert (got["crc"] == 0).mean() > 0.9


def test_empty_batches_are_no_ops(gpu_api):
    """n = 0 everywhere: nothing is launched, nothing is touched, the call succeeds."""
    z64 = np.zeros(0, np.uint64)
    x = np.zeros(2048, np.complex64)
    r = gpu_api.rx_bcch_ccch_batch(x, z64, np.zeros(0, np.uint8), sps=4)
    assert r["l2"].shape == (0, 24) and r["crc"].size == 0
    d = gpu_api.demod_batch("bcch", x, z64, 1016, sps=4)
    assert d["rv"].size == 0
    toa, rv = gpu_api.fcch_rough_batch(x, z64, 2048, sps=4)
    assert toa.size == 0 and rv.size == 0
    assert gpu_api.dkab_demod_batch(x, z64, 474, 0, sps=4)[0].size == 0
    assert gpu_api.a5_batch(1, np.zeros(8, np.uint8), np.zeros(0, np.uint32), 208).shape == (0, 208)
    rec, status, chains, found = gpu_api.rx_run(x, [], [], sps=4)
    assert found == 0 and len(rec) == 0
    assert gpu_api.channelize(np.zeros(6400, np.complex64), 2.0e6, []).shape[0] == 0


def _to_planar_numpy(iq, sps, plane_stride):
    """Sample s of the flat array at planes[(s % sps) * plane_stride + s // sps] (include/gmr1_hip.h)."""
    out = np.zeros(sps * plane_stride, np.complex64)
    s = np.arange(iq.size)
    out[(s % sps) * plane_stride + s // sps] = iq
    return out


@pytest.mark.parametrize("sps,n", [(4, 4099), (1, 1000), (3, 777), (16, 70000), (7, 5)])
def test_iq_to_planar_layout(gpu_api, sps, n):
    import torch
    rng = np.random.default_rng(sps * 1000 + n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    P = -(-n // sps) + 3                                     # a stride with slack: the slack is left alone
    d_in = torch.from_numpy(x.view(np.float32)).cuda()
    d_out = torch.full((sps * P * 2,), 7.0, dtype=torch.float32, device="cuda")
    gpu_api.iq_to_planar_dev(None, sps, n, d_in.data_ptr(), d_out.data_ptr(), P)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(np.complex64)
    want = np.full(sps * P, 7.0 + 7.0j, np.complex64)
    s = np.arange(n)
    want[(s % sps) * P + s // sps] = x
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,align", [(6000, 16), (6001, 1), (300, 1)])
def test_fused_rx_planar_layout_bit_identical(gpu_api, pkg, decoder, n, align):
    """gmr1_hip_rx_bcch_ccch_batch_planar_dev: the polyphase-planar sample layout changes addresses only -- every output
    (payload, CRC, conv, toa, freq_err, rv, all soft bits, all soft symbols) equals the interleaved call's bit for bit, for
    windows starting anywhere (align 1: offsets of every residue mod 4), in the four-bursts-per-wave shape (n > 4096)
    and the one-burst-per-wave shape, in both decoder modes."""
    import torch
    wl = workloads.bcch_ccch_mix(pkg, n=n, seed=41 + n, stride_align=align)
    if align == 1:
        # re-pack with 0..4 samples of slack in front of every window: offsets of every residue mod 4
        lens = np.where(wl["kind"] == 0, 1016, 976)
        pad = np.arange(n) % 5
        new_off = np.cumsum(np.concatenate([[0], (lens + pad)[:-1]])) + pad
        iq2 = np.zeros(int(new_off[-1] + lens[-1]) + 4, np.complex64)
        for i in range(n):
            iq2[new_off[i]:new_off[i] + lens[i]] = wl["iq"][int(wl["offset"][i]):int(wl["offset"][i]) + lens[i]]
        wl["iq"], wl["offset"] = iq2, new_off.astype(np.uint64)
        assert len(set(int(o) & 3 for o in wl["offset"])) == 4
    total = wl["iq"].size
    P = -(-total // 4)
    d_iq = torch.from_numpy(wl["iq"].view(np.float32)).cuda()
    d_pl = torch.zeros(4 * P * 2, dtype=torch.float32, device="cuda")
    gpu_api.iq_to_planar_dev(None, 4, total, d_iq.data_ptr(), d_pl.data_ptr(), P)
    d_off = torch.from_numpy(wl["offset"].astype(np.int64)).cuda()
    d_kind = torch.from_numpy(wl["kind"]).cuda()
    # with and without a caller-supplied frequency shift per burst (rx_ccch hands -freq_err over, gmr1_rx.c:821-823): the
    # rotated sync reference is then computed per burst instead of read from the zero-shift table
    fs_host = (np.random.default_rng(n).standard_normal(n) * 0.01).astype(np.float32)
    d_fs = torch.from_numpy(fs_host).cuda()
    for shift in (None, d_fs):
        _planar_vs_interleaved(gpu_api, wl, n, P, d_iq, d_pl, d_off, d_kind, shift)


def _planar_vs_interleaved(gpu_api, wl, n, P, d_iq, d_pl, d_off, d_kind, d_fs):
    import torch

    def run(planar):
        o = dict(l2=torch.zeros((n, 24), dtype=torch.uint8, device="cuda"), crc=torch.zeros(n, dtype=torch.int32, device="cuda"),
                 conv=torch.zeros(n, dtype=torch.int32, device="cuda"), toa=torch.zeros(n, dtype=torch.float32, device="cuda"),
                 fe=torch.zeros(n, dtype=torch.float32, device="cuda"), eb=torch.zeros((n, 432), dtype=torch.int8, device="cuda"),
                 ss=torch.zeros((n, 234), dtype=torch.float32, device="cuda"), rv=torch.zeros(n, dtype=torch.int32, device="cuda"))
        tail = (d_off.data_ptr(), d_kind.data_ptr(), d_fs.data_ptr() if d_fs is not None else None, o["l2"].data_ptr(),
                o["crc"].data_ptr(), o["conv"].data_ptr(),
                o["toa"].data_ptr(), o["fe"].data_ptr(), o["eb"].data_ptr(), o["ss"].data_ptr(), o["rv"].data_ptr())
        if planar:
            gpu_api.rx_bcch_ccch_batch_planar_dev(None, n, 4, d_pl.data_ptr(), P, *tail)
        else:
            gpu_api.rx_bcch_ccch_batch_dev(None, n, 4, d_iq.data_ptr(), *tail)
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in o.items()}

    a, b = run(False), run(True)
    assert (a["crc"] == 0).mean() > 0.9
    good = a["crc"] == 0
    if d_fs is None:
        assert np.array_equal(a["l2"][good], wl["l2"][good])
    for k in a:
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), f"{k} differs between the two sample layouts"


def test_fused_rx_planar_layout_refusals(gpu_api):
    import torch
    z = torch.zeros(64, dtype=torch.float32, device="cuda")
    with pytest.raises(RuntimeError):         # another oversampling: the planar kernel is built for 4 samples per symbol
        gpu_api.rx_bcch_ccch_batch_planar_dev(None, 1, 8, z.data_ptr(), 8, z.data_ptr(), z.data_ptr(), None, z.data_ptr(),
                                              z.data_ptr(), z.data_ptr(), None, None, None, None, z.data_ptr())
    with pytest.raises(RuntimeError):         # no stride
        gpu_api.rx_bcch_ccch_batch_planar_dev(None, 1, 4, z.data_ptr(), 0, z.data_ptr(), z.data_ptr(), None, z.data_ptr(),
                                              z.data_ptr(), z.data_ptr(), None, None, None, None, z.data_ptr())
    with pytest.raises(RuntimeError):         # planes shorter than the samples
        gpu_api.iq_to_planar_dev(None, 4, 100, z.data_ptr(), z.data_ptr(), 24)


def test_demod_taps_are_the_reference_debug_signals(gpu_api, orc, pkg):
    """gmr1_hip_demod_taps: the four vectors the reference dumps under ENABLE_DEBUG_SIGNAL (sdr/defs.h:35-39), rebuilt
    from the production demodulation, against a float64 restatement of pi4cxpsk.c's own steps (corr :207-237,
    burst :539-545, align :280-348, final :574-582) -- and its ordinary outputs against the batch entry's."""
    rng = np.random.default_rng(77)
    for sps, name, win, fsh in ((4, "bcch", 20, 0.0), (4, "dc6", 20, 0.05), (4, "nt3_speech", 6, -0.02), (4, "rach", 40, 0.0),
                                (8, "dc2", 32, 0.01), (2, "bcch", 20, 0.0), (16, "dc6", 64, 0.0)):
        fmt = pkg.api.burst_format(name)
        ebits = rng.integers(0, 2, size=(4, fmt.ebits), dtype=np.uint8)
        sym = pkg.synth.map_symbols(fmt, ebits)
        bb = pkg.synth.synth_windows(fmt, sym, sps, win, rng, toa_jitter=1, frac=sps > 1, cfo_hz_std=20.0, esn0_db=15.0)
        n_frac = 0
        for i in range(4):
            x = bb.iq[i, :bb.in_len]
            t = gpu_api.demod_taps(name, x, sps=sps, freq_shift=fsh)
            b = gpu_api.demod_batch(name, x, np.zeros(1, np.uint64), bb.in_len, sps=sps, freq_shift=np.float32([fsh]))
            assert t["rv"] == b["rv"][0] == 0
            assert t["toa"] == b["toa"][0] and t["sync_id"] == b["sync_id"][0] and t["freq_err"] == b["freq_err"][0]
            assert np.array_equal(t["ssyms"], b["ssyms"][0]) and np.array_equal(t["ebits"], b["ebits"][0]), (sps, name)
            o = orc.demod(name, x, sps, freq_shift=fsh) if fsh else orc.demod(name, x, sps)
            assert abs(t["toa"] - o["toa"]) < 16 / 1024
            # burst: normalised over the whole window, then de-rotated
            x64 = x.astype(np.complex128)
            nrm = (x64 - x64.mean()) / np.sqrt((np.abs(x64 - x64.mean()) ** 2).mean())
            fs = (fsh - fmt.rotation) / sps
            burst = nrm * np.exp(1j * fs * np.arange(bb.in_len))
            assert np.abs(t["burst"] - burst).max() < 2e-4 * (1 + bb.in_len * abs(fs) / 50), (sps, name)
            # corr: magnitudes summed over every chunk of every training sequence (the accumulator is never cleared)
            w = bb.in_len - fmt.length * sps + 1
            corr = np.zeros(w)
            for seq in fmt.sync:
                for pos, syms in seq:
                    ref = np.exp(2j * np.pi * np.asarray(syms) / 2 ** fmt.nbits)
                    idx = pos * sps + np.arange(w)[:, None] + sps * np.arange(len(syms))[None, :]
                    corr += np.abs((np.conj(ref)[None, :] * burst[idx]).sum(axis=1))
            assert t["corr"].shape == (w,)
            assert np.abs(t["corr"] - corr).max() < 1e-3 * corr.max(), (sps, name)
            assert abs(int(np.argmax(t["corr"])) - t["toa"]) <= 1.0
            # align: one sample per symbol at the rounded timing (sps >= 4), else through the 21-tap fractional delay
            d = int(np.round(t["toa"]))
            frac = float(t["toa"]) - d
            j = np.arange(fmt.length) * sps + d
            ok = (j >= 0) & (j < bb.in_len)
            if sps >= 4 or abs(frac) <= 0.1:
                align = np.where(ok, burst[np.clip(j, 0, bb.in_len - 1)], 0)
            else:
                n_frac += 1
                k = np.arange(21)
                xx = np.pi * ((k - 10) + frac)
                taps = np.where(np.abs(xx) >= 0.01, np.sin(xx) / np.where(xx == 0, 1, xx), 1.0)
                pad = np.concatenate([np.zeros(32), burst, np.zeros(32)])
                align = np.array([(taps * pad[32 + jj + 10 - k]).sum() if o_ else 0 for jj, o_ in zip(j, ok)])
            assert np.abs(t["align"] - align).max() < 1e-3, (sps, name, np.abs(t["align"] - align).max())
            # final: align after the fine-frequency and carrier rotations -- same magnitudes, phases = the soft symbols
            assert np.abs(np.abs(t["final"]) - np.abs(t["align"])).max() < 1e-5
            ph = np.angle(t["final"]) * 2 ** fmt.nbits / (2 * np.pi)
            dph = np.abs(ph - t["ssyms"])
            dph = np.minimum(dph, np.abs(dph - 2 ** fmt.nbits))
            assert dph[np.abs(t["final"]) > 1e-3].max() < 1e-3
            #   ... and they ARE the reference's rotations: align[i] e^{-j (ffe i + psi)} with psi fitted on the sync symbols
            rot = t["align"].astype(np.complex128) * np.exp(-1j * float(t["freq_err"]) * np.arange(fmt.length))
            sq = fmt.sync[int(t["sync_id"])]
            ph0 = sum((np.conj(np.exp(2j * np.pi * np.asarray(s) / 2 ** fmt.nbits)) * rot[p:p + len(s)]).sum() for p, s in sq)
            fin = rot * np.exp(-1j * np.angle(ph0))
            assert np.abs(t["final"] - fin).max() < 2e-3 * np.abs(fin).max(), (sps, name)
        if sps == 2:
            assert n_frac >= 1, "the fractional-delay branch was not exercised"


def test_demod_taps_refusals_and_no_power(gpu_api, pkg):
    x = np.zeros(234 * 4 + 20, np.complex64)
    t = gpu_api.demod_taps("bcch", x, sps=4)
    assert t["rv"] == -1 and not t["align"].any() and not t["final"].any() and not t["ssyms"].any()
    with pytest.raises(Exception):
        gpu_api.demod_taps("bcch", x[:100], sps=4)
    with pytest.raises(Exception):
        gpu_api.demod_taps("bcch", x, sps=0)


def test_legacy_calls_through_the_resident_server(gpu_api, pkg):
    """gmr1_pi4cxpsk_demod + gmr1_bcch_decode / gmr1_ccch_decode at 4 samples per symbol are answered by a resident one-wave
    kernel that takes requests from a mailbox in pinned memory (rx_server_kernels.inc).  Every answer equals the batch entry
    point's for the same burst -- across pauses longer than the server's idle time (it ends and is started again), switches
    of the Viterbi decoder (a new server generation), calls of other kinds in between, a device-wide synchronisation while
    it is resident, and more back-to-back calls than one server lifetime holds."""
    import time
    import torch
    n = 64
    wl = workloads.bcch_ccch_mix(pkg, n=n, seed=91, esn0_db=(7.0, 12.0))
    ref = {}
    for dec in (gpu_api.CONV_ACC, gpu_api.CONV_GENERIC):
        with gpu_api.conv_decoder(dec):
            ref[dec] = gpu_api.rx_bcch_ccch_batch(wl["iq"], wl["offset"], wl["kind"], sps=4)

    def one(i, dec):
        k = int(wl["kind"][i])
        name = "dc6" if k else "bcch"
        a = int(wl["offset"][i])
        iq = wl["iq"][a:a + wl["in_len"][k]]
        g = gpu_api.pi4cxpsk_demod(name, iq, 4, 0.0)
        r = ref[dec]
        assert g["rv"] == r["rv"][i], i
        if g["rv"]:
            return
        assert g["toa"] == r["toa"][i] and g["freq_err"] == r["freq_err"][i], i
        assert np.array_equal(g["ebits"], r["ebits"][i][:g["ebits"].size]), i
        l2, crc, conv = (gpu_api.ccch_decode if k else gpu_api.bcch_decode)(g["ebits"])
        assert crc == r["crc"][i] and conv == r["conv"][i], i
        if crc == 0:
            assert np.array_equal(l2, r["l2"][i][:l2.size]), i

    with gpu_api.conv_decoder(gpu_api.CONV_ACC):
        for i in range(8):
            one(i, gpu_api.CONV_ACC)
        time.sleep(0.02)                                   # the server has ended by now: the next call starts one
        for i in range(8, 12):
            one(i, gpu_api.CONV_ACC)
            time.sleep(0.002)
        torch.cuda.synchronize()                           # returns once the resident server has ended (its idle time)
        # other calls in between: another format (a launch per call), a batch call
        fmt = pkg.api.burst_format("nt3_speech")
        rng = np.random.default_rng(5)
        eb = rng.integers(0, 2, size=(1, fmt.ebits), dtype=np.uint8)
        bb = pkg.synth.synth_windows(fmt, pkg.synth.map_symbols(fmt, eb), 4, 6, rng, toa_jitter=1, frac=True, esn0_db=15.0)
        for i in range(12, 20):
            one(i, gpu_api.CONV_ACC)
            assert gpu_api.pi4cxpsk_demod("nt3_speech", bb.iq[0, :bb.in_len], 4, 0.0)["rv"] == 0
            gpu_api.rx_bcch_ccch_batch(wl["iq"], wl["offset"][:3], wl["kind"][:3], sps=4)
    for rep in range(3):                                   # the decoder mode changes under a live server
        for dec in (gpu_api.CONV_GENERIC, gpu_api.CONV_ACC):
            with gpu_api.conv_decoder(dec):
                for i in range(20 + 4 * rep, 24 + 4 * rep):
                    one(i, dec)
    # 0.7 s of calls back to back: longer than a server lives
    t0 = time.perf_counter()
    m = 0
    with gpu_api.conv_decoder(gpu_api.CONV_ACC):
        while time.perf_counter() - t0 < 0.7:
            one(m % n, gpu_api.CONV_ACC)
            m += 1
    assert m > 2000
    torch.cuda.synchronize()
