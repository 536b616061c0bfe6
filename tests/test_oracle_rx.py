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
