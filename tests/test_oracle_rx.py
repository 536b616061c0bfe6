"""CPU tests of the oracle's receive control loop (oracle/orc_rx.c, reference src/gmr1_rx.c:605-895)
on synthetic BCCH carriers (one ARFCN at a time).

The reference holds no capture or expected output for gmr1_rx, so what is checked is
that the loop does what the reference's design implies: it acquires the carrier from the FCCH, tracks
it, and every BCCH/CCCH frame it reports carries a payload that was actually transmitted, with the
frame number / timeslot taken from the SI1 it decoded (gmr1_rx.c:786-791, 803-848)."""
import numpy as np
import pytest

import workloads


@pytest.mark.parametrize("seed,stn,delay,cfo", [(1, 3, 2, 120.0), (2, 0, 0, -300.0), (3, 17, 5, 0.0)])
def test_rx_loop_recovers_transmitted_frames(orc, pkg, seed, stn, delay, cfo):
    iq, sent = workloads.bcch_carrier(pkg, seed, seconds=3.0, stn=stn, delay=delay, cfo_hz=cfo, esn0_db=15.0)
    rv, rec, n_chains = orc.rx_run(iq, sps=4, arfcn=7)
    assert rv == 0 and n_chains == 1
    mb, nb, mc, nc, mp = workloads.match_records(rec, sent)
    n_b = sum(s["type"] == "bcch" for s in sent)
    n_c = sum(s["type"] == "ccch" for s in sent)
    assert nb >= n_b - 3 and mp == nb
    assert mb >= nb - 3                            # frames before the first SI1 carry an unaligned fn
    assert nc >= 0.85 * n_c and mc >= nc - 1
    assert np.all(rec["arfcn"] == 7) and np.all(rec["chain"] == 0) and np.all(rec["crc"] == 0)
    # after the first SI1 the timeslot is the transmitted one
    b = rec[rec["type"] == 1]
    assert np.all(b["tn"][-3:] == stn)
    assert np.all(np.diff(b["fn"].astype(np.int64))[-3:] % 8 == 0)


def test_rx_loop_noise_only_reports_nothing(orc):
    rng = np.random.default_rng(5)
    iq = (rng.standard_normal((2 * 93600, 2), dtype=np.float32).view(np.complex64).reshape(-1))
    rv, rec, n_chains = orc.rx_run(iq, sps=4)
    assert len(rec) == 0


def test_rx_loop_short_capture_is_an_error(orc):
    iq = np.zeros(20000, np.complex64)
    rv, rec, n_chains = orc.rx_run(iq, sps=4)
    assert rv < 0 and len(rec) == 0


def test_a5_two_implementations_agree(orc, pkg):
    """The reference holds no A5 vectors; the oracle's C restatement (a5.c) and the generator's numpy
    one are written independently and must produce the same keystream."""
    import importlib
    synth = importlib.import_module(pkg.__name__ + ".synth")
    rng = np.random.default_rng(4)
    key = rng.integers(0, 256, 8, dtype=np.uint8)
    fns = np.array([0, 1, 2, 0x3F, 0x7C0, 0xF800, 0x70000, 0x7FFFF, 123456])
    ks = synth.a5_1(key, fns, 208)
    for i, fn in enumerate(fns):
        dl, ul = orc.a5(1, key, int(fn), 208)
        assert np.array_equal(dl, ks[i])
        assert ul.any() and not np.array_equal(dl, ul)
    z, _ = orc.a5(0, key, 7, 32)
    assert not z.any()


def test_rx_loop_follows_tch3_assignment(orc, pkg):
    """IMMEDIATE ASSIGNMENT on the CCCH -> DKAB / speech / FACCH3 on the traffic carrier (gmr1_rx.c:531-600),
    ciphered part of the way with A5/1."""
    kc = np.array([1, 2, 3, 4, 5, 6, 7, 8], np.uint8)
    bcch, tch, sent, sent_t = workloads.bcch_tch_pair(pkg, 5, seconds=5.0, kc=kc, cipher_after=30)
    rv, rec, n_chains = orc.rx_run_tch(bcch, tch, kc=kc)
    assert rv == 0 and n_chains == 1
    speech = {(s["fn"], bytes(s["frame0"]) + bytes(s["frame1"])) for s in sent_t if s["type"] == "speech"}
    facch = {(s["fn"], bytes(s["l2"])) for s in sent_t if s["type"] == "facch3"}
    rs = rec[rec["type"] == 0x10]
    rf = rec[rec["type"] == 0x12]
    # unciphered speech always decodes; ciphered speech only once a ciphered FACCH3 has switched ciph on
    got = {(int(r["fn"]), bytes(r["l2"][:20])) for r in rs}
    plain = {k for k, s in zip([(s["fn"], bytes(s["frame0"]) + bytes(s["frame1"])) for s in sent_t if s["type"] == "speech"],
                               [s for s in sent_t if s["type"] == "speech"]) if not s["ciph"]}
    assert plain <= got
    assert len(got & speech) >= 0.9 * len(speech)
    assert len({(int(r["fn"]), bytes(r["l2"][:10])) for r in rf} & facch) >= 1
    assert np.all(rs["len"] == 20) and np.all(rf["len"] == 10) and np.all(rec["tn"][rec["type"] >= 0x10] == 11)
    # without the traffic carrier nothing but BCCH / CCCH comes back, and they are the same frames
    rv2, rec2, _ = orc.rx_run_tch(bcch, None)
    assert np.array_equal(rec2, rec[rec["type"] < 0x10])


def test_rx_loop_follows_tch9_assignment(orc, pkg):
    """ASSIGNMENT COMMAND 1 on the FACCH3 -> NT9 bursts of the CSD carrier (gmr1_rx.c:262-353): TCH9 9k6 blocks
    come out two bursts after they went in (depth-3 inter-burst interleaver), always deciphered with A5/1."""
    kc = np.arange(8, dtype=np.uint8)
    bcch, tch, csd, kc, sent, sent_t, sent9 = workloads.bcch_tch_csd_triple(pkg, orc, 2, seconds=5.5, kc=kc, mix9=(0.0, 1.0))
    rv, rec, big, n_chains = orc.rx_run_full(bcch, tch, csd, kc=kc)
    assert rv == 0 and n_chains == 1 and len(big) > 50
    assert np.all(big["type"] == 0x18) and np.all(big["len"] == 60) and np.all(big["tn"] == 5)
    by_fn = {s["fn"]: bytes(s["l2"]) for s in sent9}
    hits = sum(by_fn.get(int(r["fn"]) - 2) == bytes(r["l2"][:60]) for r in big)
    assert hits > 0.5 * len(big)
    # the ordinary records do not change when the CSD carrier is taken away
    rv2, rec2, big2, _ = orc.rx_run_full(bcch, tch, None, kc=kc)
    assert len(big2) == 0 and np.array_equal(rec2, rec)
