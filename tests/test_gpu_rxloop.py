"""GPU parity of the batched receive loop (gmr1_hip_rx_run*, reference src/gmr1_rx.c:605-895)
against the oracle's restatement of the same loop, carrier by carrier."""
import numpy as np
import pytest

import workloads

pytestmark = pytest.mark.gpu

SPS = 4


def _key(rec):
    return [(int(r["arfcn"]), int(r["chain"]), int(r["type"]), int(r["fn"]), int(r["tn"]), bytes(r["l2"]))
            for r in rec]


def _capture(pkg):
    """Carriers of different length, timeslot, SI1 delay, CFO, SNR -- plus the degenerate ones."""
    specs = [
        dict(seed=11, seconds=3.0, stn=3, delay=2, cfo_hz=120.0, esn0_db=15.0),
        dict(seed=12, seconds=2.5, stn=0, delay=0, cfo_hz=-300.0, esn0_db=12.0),
        dict(seed=13, seconds=3.5, stn=17, delay=5, cfo_hz=0.0, esn0_db=20.0),
        dict(seed=14, seconds=3.0, stn=9, delay=7, cfo_hz=250.0, esn0_db=9.0),
        dict(seed=15, seconds=2.0, stn=21, delay=1, cfo_hz=-80.0, esn0_db=7.0, p_idle=0.5),
    ]
    streams, sents = [], []
    for sp in specs:
        sp = dict(sp)
        seed = sp.pop("seed")
        seconds = sp.pop("seconds")
        x, sent = workloads.bcch_carrier(pkg, seed, seconds=seconds, sps=SPS, **sp)
        streams.append(x)
        sents.append(sent)
    # two transmitters on one carrier, different frame timing: more than one FCCH chain
    a, sa = workloads.bcch_carrier(pkg, 21, seconds=3.0, sps=SPS, stn=2, delay=3, cfo_hz=60.0, esn0_db=18.0, t0=1000)
    b, sb = workloads.bcch_carrier(pkg, 22, seconds=3.0, sps=SPS, stn=2, delay=3, cfo_hz=90.0, esn0_db=18.0, t0=1000 + 11 * 39 * SPS)
    streams.append((a + 0.8 * b).astype(np.complex64))
    sents.append(sa + sb)
    rng = np.random.default_rng(99)
    streams.append(rng.standard_normal((2 * 93600, 2), dtype=np.float32).view(np.complex64).reshape(-1))  # noise only
    sents.append([])
    streams.append(np.zeros(20000, np.complex64))                                                         # too short
    sents.append([])
    return streams, sents


def test_rx_loop_matches_oracle_per_carrier(gpu_api, orc, pkg, decoder):
    streams, sents = _capture(pkg)
    length = np.array([s.size for s in streams], np.uint64)
    offset = np.concatenate([[0], np.cumsum(length)[:-1]]).astype(np.uint64)
    iq = np.concatenate(streams)
    arfcn = np.array([100 + 3 * i for i in range(len(streams))], np.uint16)
    rec, status, chains, found = gpu_api.rx_run(iq, offset, length, sps=SPS, arfcn=arfcn)
    assert found == len(rec)
    n_multi = 0
    for i, x in enumerate(streams):
        orv, orec, och = orc.rx_run(x, sps=SPS, arfcn=int(arfcn[i]))
        mine = rec[rec["arfcn"] == arfcn[i]]
        assert (status[i] == 0) == (orv == 0), (i, status[i], orv)
        if orv:
            assert len(mine) == 0
            continue
        assert chains[i] == och, (i, chains[i], och)
        assert _key(mine) == _key(orec), f"carrier {i}: decoded frames differ from the oracle's"
        if len(orec):
            # the Viterbi metric of a frame is a function of its soft bits, and those may differ from the oracle's by
            # one LSB on about 4e-4 of the bits (tests/test_gpu_rx.py: an fp32 phase of ~30 turns quantises at ~1e-3 of
            # a soft-bit step on both sides): a burst of ~430 soft bits has all of them identical about 85 % of the time,
            # and one LSB moves a bit's cost ((v -+ 127)^2 >> 9) by at most 1 -- so most metrics are equal and none is
            # off by more than a few units
            d = np.abs(mine["conv"].astype(int) - orec["conv"].astype(int))
            assert (d == 0).mean() >= 0.75, (i, (d == 0).mean())
            assert d.max() <= 3, (i, d.max())
        n_multi += och > 1
        # and they are what was transmitted
        mb, nb, mc, nc, mp = workloads.match_records(mine, sents[i])
        assert mp == nb and mc >= nc - 1
    assert status[-1] < 0 and len(rec[rec["arfcn"] == arfcn[-2]]) == 0
    assert n_multi >= 1, "the two-transmitter carrier should be followed as more than one chain"
    # carriers come back in order, chains in order within a carrier
    assert np.all(np.diff(rec["arfcn"].astype(int)) >= 0)


def test_rx_loop_device_resident_and_record_limit(gpu_api, orc, pkg):
    import torch
    x, sent = workloads.bcch_carrier(pkg, 31, seconds=2.5, sps=SPS, stn=5, delay=4, cfo_hz=40.0)
    n = x.size
    both = np.concatenate([x, x])
    t = torch.from_numpy(both.view(np.float32)).cuda()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        rec, status, chains, found = gpu_api.rx_run_dev(st.cuda_stream, t.data_ptr(), [0, n], [n, n], sps=SPS)
    assert list(status) == [0, 0] and found == len(rec)
    a, b = rec[rec["arfcn"] == 0], rec[rec["arfcn"] == 1]
    assert len(a) and [k[1:] for k in _key(a)] == [k[1:] for k in _key(b)]
    orv, orec, och = orc.rx_run(x, sps=SPS, arfcn=0)
    assert _key(a) == _key(orec)
    # max_records smaller than what is found: count still reported, storage truncated
    rec2, _, _, found2 = gpu_api.rx_run_dev(None, t.data_ptr(), [0, n], [n, n], sps=SPS, max_records=5)
    assert found2 == found and len(rec2) == 5 and _key(rec2) == _key(rec[:5])


def test_rx_loop_more_chains_than_compute_units(gpu_api, orc, pkg):
    """300 carriers = more work-groups than the 256 CUs hold at once (each walks its whole capture): late
    work-groups start when early ones finish; every carrier must still yield exactly its own frames."""
    import torch
    distinct = [workloads.bcch_carrier(pkg, 90 + a, seconds=1.6, sps=SPS, stn=(7 * a) % 24, delay=a + 1, cfo_hz=25.0 * a)[0]
                for a in range(3)]
    n = distinct[0].size
    A = 300
    t = torch.from_numpy(np.concatenate(distinct).view(np.float32)).cuda()
    big = torch.cat([t] * (A // 3)).contiguous()
    offset = np.arange(A, dtype=np.uint64) * np.uint64(n)
    length = np.full(A, n, np.uint64)
    rec, status, chains, found = gpu_api.rx_run_dev(None, big.data_ptr(), offset, length, sps=SPS, max_records=1 << 17)
    assert not status.any() and found == len(rec)
    refs = [_key(orc.rx_run(distinct[a], sps=SPS, arfcn=0)[1]) for a in range(3)]
    assert all(len(r) > 20 for r in refs)
    # records come back carrier by carrier
    bounds = np.searchsorted(rec["arfcn"], np.arange(A + 1))
    for a in range(A):
        got = [k[1:] for k in _key(rec[bounds[a]:bounds[a + 1]])]
        assert got == [k[1:] for k in refs[a % 3]], a


def test_rx_loop_carriers_of_very_different_lengths(gpu_api, orc, pkg):
    """The loop walks its chains in time slices (k_rx_chain launched several times, the CCCH batch of one slice under the
    next): a chain that reaches the end of its capture in the first slice sits out the others, a long one uses them all,
    and one too short to hold a frame yields nothing - each carrier's records are the oracle's."""
    secs = [0.9, 7.0, 0.25, 2.0]
    xs = [workloads.bcch_carrier(pkg, 120 + i, seconds=sec, sps=SPS, stn=(3 * i) % 24, delay=i, cfo_hz=35.0 * i)[0]
          for i, sec in enumerate(secs)]
    length = np.array([x.size for x in xs], np.uint64)
    offset = np.concatenate([[0], np.cumsum(length)[:-1]]).astype(np.uint64)
    rec, status, chains, found = gpu_api.rx_run(np.concatenate(xs), offset, length, sps=SPS, arfcn=np.arange(4, dtype=np.uint16))
    assert found == len(rec)
    n_tot = 0
    for i, x in enumerate(xs):
        orv, orec, och = orc.rx_run(x, sps=SPS, arfcn=i)
        assert (status[i] == 0) == (orv == 0), (i, status[i], orv)
        mine = rec[rec["arfcn"] == i]
        if orv:
            assert len(mine) == 0
            continue
        assert chains[i] == och
        assert _key(mine) == _key(orec), f"carrier {i}"
        n_tot += len(orec)
    assert n_tot > 150


def test_rx_loop_rejects_bad_arguments(gpu_api):
    x = np.zeros(1000, np.complex64)
    with pytest.raises(Exception):
        gpu_api.rx_run(x, [0], [2000], sps=SPS)            # carrier runs past the buffer
    for bad in (0, 17):
        with pytest.raises(Exception):
            gpu_api.rx_run(x, [0], [1000], sps=bad)        # sps outside the reference's 1..16 (gmr1_rx.c:919-922)
    rec, status, chains, found = gpu_api.rx_run(x, [0], [1000], sps=SPS)
    assert found == 0 and status[0] < 0


def test_rx_capture_sharded_single_rank(gpu_api, orc, pkg):
    """The config-4 runner on a world of one: scatter (local), receive loop, gather."""
    import torch
    import torch.distributed as dist
    from importlib import import_module
    shard = import_module(pkg.__name__ + ".shard")
    n = int(2.5 * 23400 * SPS)
    xs = [workloads.bcch_carrier(pkg, 40 + a, seconds=2.5, sps=SPS, stn=a, delay=a % 8, cfo_hz=30.0 * a)[0]
          for a in range(3)]
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1)
    try:
        slices = [torch.from_numpy(x).cuda() for x in xs]
        rec = shard.rx_capture_sharded(gpu_api, slices, 3, n, sps=SPS, device=None)
    finally:
        dist.destroy_process_group()
    # the order one receive loop over all carriers gives: carrier, chain, order of emission
    ref = np.concatenate([orc.rx_run(xs[a], sps=SPS, arfcn=a)[1] for a in range(3)])
    assert _key(rec) == _key(ref)


def test_native_sharded_entry_world_of_one(gpu_api, orc, pkg):
    """gmr1_hip_rx_run_sharded (gmr1_hip_shard.h) on a communicator of one rank: RCCL is found and initialised
    (ncclGetUniqueId, ncclCommInitRank), the exchanges degenerate, and what comes back is gmr1_hip_rx_run_dev's output --
    records, status and chain counts -- with the caller's carrier labels, also when two carriers share a label."""
    import torch
    n = int(2.5 * 23400 * SPS)
    xs = [workloads.bcch_carrier(pkg, 50 + a, seconds=2.5, sps=SPS, stn=2 * a, delay=a % 8, cfo_hz=-25.0 * a)[0]
          for a in range(3)]
    iq = torch.from_numpy(np.concatenate(xs).view(np.float32)).cuda()
    offset = np.arange(3, dtype=np.uint64) * np.uint64(n)
    length = np.full(3, n, np.uint64)
    st = torch.cuda.current_stream().cuda_stream
    sh = gpu_api.Shard(gpu_api.Shard.unique_id(), 0, 1)
    try:
        for labels in ([700, 701, 702], [9, 9, 11]):
            rec, status, chains, timing = sh.rx_run(st, iq.data_ptr(), offset, length, sps=SPS, arfcn=labels)
            one, st1, ch1, _ = gpu_api.rx_run_dev(st, iq.data_ptr(), offset, length, sps=SPS, arfcn=np.array(labels, np.uint16))
            assert _key(rec) == _key(one) and len(rec) > 0
            assert np.array_equal(status, st1) and np.array_equal(chains, ch1)
            assert (timing >= 0).all() and timing[1] > 0
            # the samples already where they are processed (gmr1_hip_rx_run_sharded_resident): the same records
            rec2, st2, ch2, _ = sh.rx_run(st, iq.data_ptr(), offset, length, sps=SPS, arfcn=labels, resident=True)
            assert _key(rec2) == _key(one) and np.array_equal(st2, st1) and np.array_equal(ch2, ch1)
        # arguments every rank can check are refused before anything collective starts
        for bad in (0, 17):
            with pytest.raises(gpu_api.Gmr1HipError, match="-22"):
                sh.rx_run(st, iq.data_ptr(), offset, length, sps=bad)
        with pytest.raises(gpu_api.Gmr1HipError, match="-22"):
            sh.rx_run(st, 0, offset, length, sps=SPS)                      # root without samples
    finally:
        sh.close()
    ref = np.concatenate([orc.rx_run(xs[a], sps=SPS, arfcn=[9, 9, 11][a])[1] for a in range(3)])
    assert _key(rec) == _key(ref)


def _key_n(rec):
    return [(int(r["arfcn"]), int(r["chain"]), int(r["type"]), int(r["fn"]), int(r["tn"]), int(r["len"]),
             bytes(r["l2"][:int(r["len"])])) for r in rec]


def test_rx_loop_tch3_follow_up_matches_oracle(gpu_api, orc, pkg, decoder):
    """IMM.ASS -> DKAB / speech / FACCH3 on the traffic carrier, plain and A5/1-ciphered (gmr1_rx.c:355-600):
    the record sequence is the oracle's, carrier by carrier."""
    rng = np.random.default_rng(77)
    cases = [
        dict(seed=5, seconds=5.0, kc=np.array([1, 2, 3, 4, 5, 6, 7, 8], np.uint8), cipher_after=30),
        dict(seed=6, seconds=4.0, kc=None, cipher_after=None, tn=4, p=7, stn=1, delay=6),
        dict(seed=7, seconds=4.5, kc=rng.integers(0, 256, 8, dtype=np.uint8), cipher_after=0, tn=20, p=33,
             mix=(0.2, 0.3, 0.5)),
        dict(seed=8, seconds=4.0, kc=None, cipher_after=None, tn=9, p=12, k_stop=45),      # the call ends: weak DKABs -> END
    ]
    bc, tc, kcs, sents = [], [], [], []
    for cs in cases:
        cs = dict(cs)
        seed = cs.pop("seed")
        b, t, s_b, s_t = workloads.bcch_tch_pair(pkg, seed, **cs)
        bc.append(b); tc.append(t); sents.append(s_t)
        kcs.append(cs["kc"] if cs["kc"] is not None else np.zeros(8, np.uint8))
    # one carrier without a traffic capture worth speaking of: noise
    b, t, _, _ = workloads.bcch_tch_pair(pkg, 9, seconds=3.0)
    bc.append(b); tc.append((rng.standard_normal((b.size, 2)) * 0.05).astype(np.float32).view(np.complex64).reshape(-1))
    kcs.append(np.zeros(8, np.uint8)); sents.append([])
    length = np.array([x.size for x in bc], np.uint64)
    offset = np.concatenate([[0], np.cumsum(length)[:-1]]).astype(np.uint64)
    rec, status, chains, found = gpu_api.rx_run_tch(np.concatenate(bc), np.concatenate(tc), offset, length, sps=SPS,
                                                    kc=np.stack(kcs))
    assert found == len(rec) and not status.any()
    n_tch = 0
    for i in range(len(bc)):
        orv, orec, och = orc.rx_run_tch(bc[i], tc[i], sps=SPS, arfcn=i, kc=kcs[i])
        mine = rec[rec["arfcn"] == i]
        assert orv == 0 and chains[i] == och
        assert _key_n(mine) == _key_n(orec), f"carrier {i}: records differ from the oracle's"
        n_tch += int(np.sum(orec["type"] >= 0x10))
        sp = {(s["fn"], bytes(s["frame0"]) + bytes(s["frame1"])) for s in sents[i] if s["type"] == "speech" and not s["ciph"]}
        got = {(int(r["fn"]), bytes(r["l2"][:20])) for r in mine[mine["type"] == 0x10]}
        assert sp <= got
    assert n_tch > 100
    # without the traffic carriers: exactly the BCCH / CCCH records
    rec0, _, _, _ = gpu_api.rx_run(np.concatenate(bc), offset, length, sps=SPS)
    assert _key_n(rec0) == _key_n(rec[rec["type"] < 0x10])


def test_rx_loop_at_sps_8(gpu_api, orc, pkg):
    """The whole receive loop at another oversampling (gmr1_rx takes sps from the command line, gmr1_rx.c:917)."""
    x, sent = workloads.bcch_carrier(pkg, 61, seconds=2.5, sps=8, stn=6, delay=3, cfo_hz=100.0)
    rec, status, chains, found = gpu_api.rx_run(x, [0], [x.size], sps=8)
    orv, orec, och = orc.rx_run(x, sps=8, arfcn=0)
    assert status[0] == orv == 0 and chains[0] == och
    assert _key(rec) == _key(orec)
    mb, nb, mc, nc, mp = workloads.match_records(rec, sent)
    assert nb >= 5 and mp == nb and mc >= nc - 1


@pytest.mark.parametrize("sps", [2, 3, 10, 16])
def test_rx_loop_outside_4_to_8_samples_per_symbol(gpu_api, orc, pkg, sps, decoder):
    """gmr1_rx accepts 1..16 samples per symbol (gmr1_rx.c:919-922).  Below 4 the demodulator delays the burst by a
    fraction of a sample (pi4cxpsk.c:298-343), above 8 a BCCH window passes 2048 samples: the loop then runs the one-burst
    generic body (k_rx_chain<..., ONE>, k_rx for the CCCH lists).  Records identical to the oracle's loop."""
    x, sent = workloads.bcch_carrier(pkg, 80 + sps, seconds=2.5, sps=sps, stn=4, delay=2, cfo_hz=70.0)
    rec, status, chains, found = gpu_api.rx_run(x, [0], [x.size], sps=sps)
    orv, orec, och = orc.rx_run(x, sps=sps, arfcn=0)
    assert status[0] == orv == 0 and chains[0] == och
    assert _key(rec) == _key(orec)
    mb, nb, mc, nc, mp = workloads.match_records(rec, sent)
    assert nb >= 5 and mp == nb and mc >= nc - 1


@pytest.mark.parametrize("sps", [2, 10])
def test_rx_loop_tch3_follow_up_outside_4_to_8(gpu_api, orc, pkg, sps):
    """The TCH3 follow-up (IMM.ASS -> DKAB / speech / FACCH3, ciphered) at 2 and 10 samples per symbol: records are the oracle's."""
    kc = np.array([9, 8, 7, 6, 5, 4, 3, 2], np.uint8)
    b, t, s_b, s_t = workloads.bcch_tch_pair(pkg, 15 + sps, seconds=4.0, sps=sps, kc=kc, cipher_after=25)
    rec, status, chains, found = gpu_api.rx_run_tch(b, t, [0], [b.size], sps=sps, kc=kc[None, :])
    orv, orec, och = orc.rx_run_tch(b, t, sps=sps, arfcn=0, kc=kc)
    assert status[0] == orv == 0 and chains[0] == och
    assert _key_n(rec) == _key_n(orec)
    assert int(np.sum(orec["type"] >= 0x10)) > 20


def _key_big(rec):
    return [(int(r["arfcn"]), int(r["chain"]), int(r["type"]), int(r["fn"]), int(r["tn"]), int(r["len"]),
             bytes(r["l2"][:int(r["len"])]), int(r["conv"])) for r in rec]


def test_rx_loop_tch9_follow_up_matches_oracle(gpu_api, orc, pkg, decoder):
    """The whole application: IMM.ASS -> TCH3; ASSIGNMENT COMMAND 1 on its FACCH3 -> NT9 bursts on the CSD carrier,
    FACCH9 / TCH9 9k6, A5/1 (gmr1_rx.c:262-353).  Records and big records are the oracle's, carrier by carrier."""
    kc1 = np.arange(8, dtype=np.uint8)
    cases = [dict(seed=2, kc=kc1, mix9=(0.0, 1.0)), dict(seed=3, kc=None, mix9=(0.3, 0.6)),
             dict(seed=4, kc=kc1, mix9=(0.2, 0.5), tn9=17)]
    bc, tc, cc, kcs, sents9 = [], [], [], [], []
    for cs in cases:
        cs = dict(cs)
        seed = cs.pop("seed")
        b, t, c, kc, _, _, s9 = workloads.bcch_tch_csd_triple(pkg, orc, seed, seconds=5.5, **cs)
        bc.append(b); tc.append(t); cc.append(c); kcs.append(kc); sents9.append(s9)
    length = np.array([x.size for x in bc], np.uint64)
    offset = np.concatenate([[0], np.cumsum(length)[:-1]]).astype(np.uint64)
    rec, big, status, chains = gpu_api.rx_run_full(np.concatenate(bc), np.concatenate(tc), np.concatenate(cc),
                                                   offset, length, sps=SPS, kc=np.stack(kcs))
    assert not status.any()
    n_big = 0
    for i in range(len(bc)):
        orv, orec, obig, och = orc.rx_run_full(bc[i], tc[i], cc[i], sps=SPS, arfcn=i, kc=kcs[i])
        assert orv == 0 and chains[i] == och
        assert _key_n(rec[rec["arfcn"] == i]) == _key_n(orec), f"carrier {i}: records differ"
        assert _key_big(big[big["arfcn"] == i]) == _key_big(obig), f"carrier {i}: NT9 records differ"
        n_big += len(obig)
    assert n_big > 100
    # TCH9-only carrier: a block comes out two bursts after it went in (depth-3 inter-burst interleaver)
    sent = {s["fn"]: bytes(s["l2"]) for s in sents9[0] if s["type"] == "tch9"}
    mine = big[(big["arfcn"] == 0) & (big["type"] == 0x18)]
    hits = sum(sent.get(int(r["fn"]) - 2) == bytes(r["l2"][:60]) for r in mine)
    assert hits > 0.5 * len(mine)
    # without the CSD carrier: the same ordinary records, no big ones
    rec2, big2, _, _ = gpu_api.rx_run_full(np.concatenate(bc), np.concatenate(tc), None, offset, length, sps=SPS,
                                           kc=np.stack(kcs))
    assert len(big2) == 0 and _key_n(rec2) == _key_n(rec)


@pytest.mark.timeout(120)
def test_rx_loop_survives_hostile_samples(gpu_api, pkg):
    """NaN / Inf / huge values in the capture: every loop in the path has a data-independent trip count, so
    the call returns (with whatever it could decode) instead of hanging or faulting."""
    x, sent = workloads.bcch_carrier(pkg, 71, seconds=2.5, sps=SPS, stn=3, delay=2)
    rng = np.random.default_rng(0)
    bad = x.copy()
    idx = rng.choice(bad.size, 400, replace=False)
    bad[idx[:100]] = np.nan
    bad[idx[100:200]] = np.inf
    bad[idx[200:300]] = 1e30
    bad[idx[300:]] = -np.inf + 1j * np.nan
    allnan = np.full(x.size, np.nan + 1j * np.nan, np.complex64)
    streams = [bad, allnan, x]
    length = np.array([s.size for s in streams], np.uint64)
    offset = np.concatenate([[0], np.cumsum(length)[:-1]]).astype(np.uint64)
    rec, status, chains, found = gpu_api.rx_run(np.concatenate(streams), offset, length, sps=SPS)
    clean = rec[rec["arfcn"] == 2]
    mb, nb, mc, nc, mp = workloads.match_records(clean, sent)
    assert nb >= 5 and mp == nb                     # the clean carrier next to them is untouched
    assert len(rec[rec["arfcn"] == 1]) == 0
    # the fused kernel on its own
    n = 64
    wl = workloads.bcch_ccch_mix(pkg, n=n, seed=2)
    iq = wl["iq"].copy()
    iq[rng.choice(iq.size, 2000, replace=False)] = np.nan
    got = gpu_api.rx_bcch_ccch_batch(iq, wl["offset"], wl["kind"], sps=SPS)
    assert got["rv"].shape == (n,)


@pytest.mark.timeout(300)
def test_rx_loop_forced_mis_speculation(gpu_api, orc, pkg, decoder):
    """The loop's front stage runs two rounds ahead of the decoder on the assumption that the BCCH burst will pass its CRC
    and that its SI1 will leave the TDMA position alone (k_rx_chain_pipe).  Carriers on which that assumption fails again
    and again - a third or more of the bursts failing the CRC, bursts that are not there, a first SI1 that arrives after
    many other-SI bursts, SI1s that re-label the timeslot or the frame count in mid-capture, stretches of strong noise -
    must give exactly the oracle's records: a squashed round is redone from the true state."""
    seconds = 12.0
    specs = [
        dict(seed=301, stn=4, delay=1, cfo_hz=70.0, esn0_db=4.0),                               # many CRC failures
        dict(seed=302, stn=7, delay=3, cfo_hz=-150.0, esn0_db=3.3),                             # more of them
        dict(seed=303, stn=11, delay=6, cfo_hz=30.0, esn0_db=14.0, absent_bcch=range(2, 36, 3)),   # every third burst missing
        dict(seed=304, stn=19, delay=2, cfo_hz=200.0, esn0_db=14.0, other_first=14),            # the first SI1 arrives late
        dict(seed=305, stn=5, delay=4, cfo_hz=-40.0, esn0_db=16.0, si1_lie={12: (4, 9), 13: (4, 9)}),   # SI1 re-labels the timeslot
        dict(seed=306, stn=2, delay=0, cfo_hz=10.0, esn0_db=16.0, si1_lie={9: (3, 2)}),         # SI1 changes the frame count
        dict(seed=307, stn=13, delay=5, cfo_hz=0.0, esn0_db=5.0, absent_bcch=(5, 6, 7, 20), other_first=6),
    ]
    streams, sents = [], []
    for sp in specs:
        sp = dict(sp)
        x, sent = workloads.bcch_carrier(pkg, sp.pop("seed"), seconds=seconds, sps=SPS, **sp)
        streams.append(x)
        sents.append(sent)
    # hostile samples on top of a weak carrier: stretches of the capture replaced by strong noise, at BCCH bursts and between
    rng = np.random.default_rng(5)
    x = streams[0].copy()
    for pos in rng.integers(0, x.size - 6000, 25):
        x[pos:pos + 5000] = rng.standard_normal((5000, 2), dtype=np.float32).view(np.complex64).reshape(-1) * np.float32(3.0)
    streams.append(x)
    sents.append(sents[0])
    length = np.array([s.size for s in streams], np.uint64)
    offset = np.concatenate([[0], np.cumsum(length)[:-1]]).astype(np.uint64)
    arfcn = np.arange(len(streams), dtype=np.uint16) + 40
    rec, status, chains, found = gpu_api.rx_run(np.concatenate(streams), offset, length, sps=SPS, arfcn=arfcn)
    assert found == len(rec)
    fail_frac = []
    for i, x in enumerate(streams):
        orv, orec, och = orc.rx_run(x, sps=SPS, arfcn=int(arfcn[i]))
        mine = rec[rec["arfcn"] == arfcn[i]]
        assert (status[i] == 0) == (orv == 0), (i, status[i], orv)
        assert chains[i] == och
        assert _key(mine) == _key(orec), f"carrier {i}: records differ from the oracle's"
        n_sent = sum(1 for s_ in sents[i] if s_["type"] == "bcch")
        n_got = int((orec["type"] == 1).sum()) if len(orec) else 0
        fail_frac.append(1.0 - n_got / max(n_sent, 1))
    print("BCCH bursts sent but not emitted, per carrier:", " ".join("%.2f" % f for f in fail_frac))
    # the assumption did fail often enough to mean something (carrier 2: a third of its bursts were never sent)
    assert fail_frac[0] >= 0.2 and fail_frac[1] >= 0.2 and fail_frac[7] >= 0.2, fail_frac
    assert all(len(rec[rec["arfcn"] == a]) > 100 for a in arfcn[2:7])


@pytest.mark.timeout(900)
def test_rx_loop_production_length_64_carriers_60_s(gpu_api, orc, pkg, decoder):
    """BASELINE configs[3] at its size: 64 carriers x 60 s (eight distinct, tiled -- as bench.py --workload rx runs it), one
    gmr1_hip_rx_run_dev call.  187 feedback rounds per chain, the four slice hand-overs at production length, a minute of
    float freq_err accumulation (gmr1_rx.c:782-789): EVERY distinct carrier's record sequence is the oracle loop's
    (gmr1_rx.c:852-895), every tile of a carrier equals it, in both decoder modes."""
    import torch
    A, seconds, distinct = 64, 60.0, 8
    host = [workloads.bcch_carrier(pkg, 700 + a, seconds=seconds, sps=SPS, stn=(5 * a) % 24, delay=a % 8,
                                   cfo_hz=40.0 * (a - 3), esn0_db=10.0 + a)[0] for a in range(distinct)]
    ns = host[0].size
    base = torch.from_numpy(np.concatenate(host).view(np.float32)).cuda()
    iq = torch.cat([base] * (A // distinct)).contiguous()
    offset = np.arange(A, dtype=np.uint64) * np.uint64(ns)
    length = np.full(A, ns, np.uint64)
    rec, status, chains, found = gpu_api.rx_run_dev(None, iq.data_ptr(), offset, length, sps=SPS, max_records=A * 4096)
    assert not status.any() and found == len(rec)
    bounds = np.searchsorted(rec["arfcn"], np.arange(A + 1))
    total = 0
    for a in range(distinct):
        orv, orec, och = orc.rx_run(host[a], sps=SPS, arfcn=a)
        assert orv == 0 and len(orec) > 1000                   # a carrier-minute: 187 BCCH + the CCCH bursts that decode
        want = [k[1:] for k in _key(orec)]
        for t in range(a, A, distinct):
            got = [k[1:] for k in _key(rec[bounds[t]:bounds[t + 1]])]
            assert got == want, f"carrier {t} (tile of {a}): records differ from the oracle's loop"
            assert chains[t] == och
        total += len(orec)
    assert found == total * (A // distinct)


@pytest.mark.timeout(900)
def test_rx_loop_tch3_follow_up_production_length(gpu_api, orc, pkg, decoder):
    """gmr1_hip_rx_run_tch on one 60-s BCCH + traffic carrier pair (IMM.ASS early, ciphering switched on later): the
    record sequence -- BCCH / CCCH, DKAB, speech, FACCH3 over a minute of a call -- is the oracle's (gmr1_rx.c:355-600)."""
    kc = np.array([9, 8, 7, 6, 5, 4, 3, 2], np.uint8)
    b, t, s_b, s_t = workloads.bcch_tch_pair(pkg, 61, seconds=60.0, kc=kc, cipher_after=400, k_ass=25)
    rec, status, chains, found = gpu_api.rx_run_tch(b, t, [0], [b.size], sps=SPS, kc=kc[None, :])
    orv, orec, och = orc.rx_run_tch(b, t, sps=SPS, arfcn=0, kc=kc)
    assert orv == 0 and not status.any() and chains[0] == och
    assert int(np.sum(orec["type"] >= 0x10)) > 500
    assert _key_n(rec) == _key_n(orec)


def test_rx_loop_record_buffer_in_pinned_and_device_memory(gpu_api, orc, pkg):
    """gmr1_hip_rx_run_dev closes the records up on the device and copies them once: into pageable host memory through the
    library's pinned block, straight into a pinned or a DEVICE buffer.  The same records every way, also when the buffer
    holds fewer than there are."""
    import torch
    xs = [workloads.bcch_carrier(pkg, 41 + a, seconds=2.2, sps=SPS, stn=4 * a, delay=a, cfo_hz=70.0 * a)[0] for a in range(3)]
    n = xs[0].size
    t = torch.from_numpy(np.concatenate(xs).view(np.float32)).cuda()
    offset, length = [0, n, 2 * n], [n, n, n]
    rec, status, chains, found = gpu_api.rx_run_dev(None, t.data_ptr(), offset, length, sps=SPS)       # pageable
    assert found == len(rec) > 60 and not status.any()
    want = np.concatenate([orc.rx_run(x, sps=SPS, arfcn=a)[1] for a, x in enumerate(xs)])
    assert _key(rec) == _key(want)
    isz = gpu_api.RX_RECORD.itemsize
    for cap in (found + 10, found, 17):
        pin = torch.zeros(cap * isz, dtype=torch.uint8, pin_memory=True)
        dev = torch.zeros(cap * isz, dtype=torch.uint8, device="cuda")
        for buf in (pin, dev):
            got, st2, ch2 = gpu_api.rx_run_dev_raw(None, t.data_ptr(), offset, length, buf.data_ptr(), cap, sps=SPS)
            torch.cuda.synchronize()
            assert got == found and np.array_equal(st2, status) and np.array_equal(ch2, chains)
            back = buf.cpu().numpy().view(gpu_api.RX_RECORD)
            k = min(cap, found)
            assert back[:k].tobytes() == rec[:k].tobytes()
            assert not back[k:].view(np.uint8).any()                 # nothing written past what fits


def test_measurement_aids(gpu_api, pkg):
    """gmr1_hip_clock_probe_dev reads a plausible shader clock on the device; gmr1_hip_rx_run_last_timing returns the phases of
    the calling thread's last receive-loop call, and they add up to less than the call took."""
    import time
    import torch
    mhz, wall = gpu_api.clock_probe_dev(torch.cuda.current_stream().cuda_stream, 200)
    assert 300.0 < mhz < 3500.0 and 10.0 <= wall <= 1000.0, (mhz, wall)
    with pytest.raises(Exception):
        gpu_api.clock_probe_dev(None, 0)
    x, _ = workloads.bcch_carrier(pkg, 31, seconds=2.5, sps=SPS, stn=5, delay=4, cfo_hz=40.0)
    t0 = time.perf_counter()
    rec, status, chains, found = gpu_api.rx_run(x, [0], [x.size], sps=SPS)
    dt_ms = (time.perf_counter() - t0) * 1e3
    ph = gpu_api.rx_run_last_timing()
    assert found > 10 and not status.any()
    assert all(v >= 0.0 for v in ph.values()), ph
    assert ph["acquisition_ms"] > 0.01 and ph["chain_ms"] > 0.01
    assert sum(ph.values()) <= dt_ms * 1.05, (ph, dt_ms)
