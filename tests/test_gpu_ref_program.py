"""The reference's application, UNCHANGED, RUNNING on the GPU library: src/gmr1_rx.c + src/gsmtap.c (reference
src/gmr1_rx.c:900-991 main, :605-744 acquisition, :746-895 frame loop; src/gsmtap.c:43-71) linked against
libgmr1_hip.so and the test-only stand-ins for its eleven libosmocore / libosmo-dsp calls (tests/c/tp_shim_gmr1_rx.c:
file loading, a message buffer, the GSMTAP "socket" appending to a file).  Built in the container by
tests/ref_rx_program.py into oracle/_ref/gmr1_rx_hip, which travels to the GPU box.

What is asserted: the GSMTAP messages the program sends -- every gmr1_fcch_* / gmr1_pi4cxpsk_demod / gmr1_bcch_decode /
gmr1_ccch_decode call of its loop answered by the HIP library, one burst per call -- are, message for message,
(1) the records of ONE batched gmr1_hip_rx_run call on the same capture and (2) the records of the CPU oracle's loop;
and with a traffic carrier and a key on the command line, gmr1_hip_rx_run_tch's."""
import os

import numpy as np
import pytest

import ref_rx_program
import workloads

pytestmark = pytest.mark.gpu

SPS = 4


def _exe():
    exe = ref_rx_program.build()
    if not exe or not os.path.exists(exe):
        pytest.skip("oracle/_ref/gmr1_rx_hip was not built (needs /root/reference at build time)")
    return exe


def _rec_key(rec):
    return [(int(r["type"]), int(r["fn"]), int(r["tn"]), bytes(r["l2"][:int(r["len"])])) for r in rec]


def test_unchanged_gmr1_rx_runs_on_the_library_and_sends_the_batched_loops_records(gpu_api, orc, pkg):
    _exe()
    x, sent = workloads.bcch_carrier(pkg, 4711, seconds=10.0, sps=SPS, stn=5, delay=3, cfo_hz=140.0, esn0_db=14.0)
    rc, msgs, err = ref_rx_program.run(x, sps=SPS)
    assert rc == 0, err[-2000:]
    assert "Primary FCCH found" in err
    assert all(m[4] == (2, 0x0a) for m in msgs)              # GSMTAP_VERSION, GSMTAP_TYPE_GMR1_UM
    prog = [m[:4] for m in msgs]
    assert len(prog) > 150, len(prog)                        # 250 frames: ~31 BCCH + most CCCH slots

    rec, status, chains, found = gpu_api.rx_run(x, [0], [x.size], sps=SPS)
    assert status[0] == 0 and found == len(rec)
    assert prog == _rec_key(rec), "the unchanged program's GSMTAP messages differ from gmr1_hip_rx_run's records"

    orv, orec, och = orc.rx_run(x, sps=SPS, arfcn=0)
    assert orv == 0 and och == chains[0]
    assert prog == _rec_key(orec), "the unchanged program's GSMTAP messages differ from the oracle loop's records"

    # and they are what the generator transmitted
    mb, nb, mc, nc, mp = workloads.match_records(rec, sent)
    assert nb >= 28 and mp == nb and mc >= nc - 1


def test_unchanged_gmr1_rx_two_transmitters_and_hostile_input(gpu_api, orc, pkg):
    """More than one FCCH chain (fcch_multi_process walks them one after the other, gmr1_rx.c:704-741) and a capture
    with no carrier at all (the program reports the failed acquisition, gmr1_rx.c:962-966)."""
    _exe()
    a, _ = workloads.bcch_carrier(pkg, 21, seconds=4.0, sps=SPS, stn=2, delay=3, cfo_hz=60.0, esn0_db=18.0, t0=1000)
    b, _ = workloads.bcch_carrier(pkg, 22, seconds=4.0, sps=SPS, stn=2, delay=3, cfo_hz=90.0, esn0_db=18.0, t0=1000 + 11 * 39 * SPS)
    x = (a + 0.8 * b).astype(np.complex64)
    rc, msgs, err = ref_rx_program.run(x, sps=SPS)
    assert rc == 0, err[-2000:]
    rec, status, chains, found = gpu_api.rx_run(x, [0], [x.size], sps=SPS)
    assert chains[0] > 1
    assert [m[:4] for m in msgs] == _rec_key(rec)
    orv, orec, och = orc.rx_run(x, sps=SPS, arfcn=0)
    assert [m[:4] for m in msgs] == _rec_key(orec)

    rng = np.random.default_rng(5)
    noise = rng.standard_normal((93600 * 2, 2), dtype=np.float32).view(np.complex64).reshape(-1)
    rc, msgs, err = ref_rx_program.run(noise, sps=SPS)
    rec, status, chains, found = gpu_api.rx_run(noise, [0], [noise.size], sps=SPS)
    orv, orec, och = orc.rx_run(noise, sps=SPS, arfcn=0)
    assert (rc == 0) == (status[0] == 0) == (orv == 0)
    assert [m[:4] for m in msgs] == _rec_key(rec) == _rec_key(orec)
