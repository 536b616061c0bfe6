"""The decoder-choice sensitivity (decision D1 vs D1b, tests/test_oracle_d1b.py) on the benchmark workloads at full
size, from the soft bits the GPU demodulator produced: configs[2] (100 000 BCCH / CCCH bursts) and configs[4]'s NT3 mix
(100 000 bursts: 90 000 speech + 2 500 FACCH3 groups).  The GPU decodes with D1 (gmr1_hip_set_conv_decoder(GMR1_HIP_CONV_GENERIC);
bit-exact with the oracle's D1, other tests); the oracle decodes the SAME soft bits with D1b and the differences are
counted.  The product decodes with D1b too (GMR1_HIP_CONV_ACC, its default since round 4), and at these sizes it must
return exactly what the oracle's D1b returns -- from the fused kernel and from the stand-alone layer-1 kernels."""
import numpy as np
import pytest

import workloads

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_bench_bursts_100k_decoder_choice(gpu_api, orc, pkg):
    wl = workloads.bcch_ccch_mix(pkg, n=100_000, seed=3)              # bench.py's configs[2] workload (rank 0)
    with gpu_api.conv_decoder(gpu_api.CONV_GENERIC):                  # D1: the generic decoder (not the default any more)
        got = gpu_api.rx_bcch_ccch_batch(wl["iq"], wl["offset"], wl["kind"], sps=4, want_ssyms=False)
    found = got["rv"] == 0
    l2 = np.zeros((wl["kind"].size, 24), np.uint8)
    crc = np.full(wl["kind"].size, -1, np.int32)
    with orc.conv_mode(1):
        for k, dec, neb in ((0, orc.bcch_decode, 424), (1, orc.ccch_decode, 432)):
            rows = np.nonzero((wl["kind"] == k) & found)[0]
            o = dec(got["ebits"][rows][:, :neb])
            l2[rows], crc[rows] = o[0], o[1]
    # the product in the accelerated decoder's mode: the fused kernel (same demodulator, so the same soft bits) and the
    # stand-alone layer-1 kernel on those soft bits both return the oracle's D1b frames, verdicts, and conv_rv = 0
    with gpu_api.conv_decoder(gpu_api.CONV_ACC):
        acc = gpu_api.rx_bcch_ccch_batch(wl["iq"], wl["offset"], wl["kind"], sps=4, want_ssyms=False)
        assert np.array_equal(acc["ebits"], got["ebits"]) and np.array_equal(acc["rv"], got["rv"])
        assert np.array_equal(acc["crc"][found], crc[found]) and np.array_equal(acc["l2"][found], l2[found])
        assert not acc["conv"].any()
        for k, dec, neb in ((0, gpu_api.bcch_decode_batch, 424), (1, gpu_api.ccch_decode_batch, 432)):
            rows = np.nonzero((wl["kind"] == k) & found)[0]
            g = dec(got["ebits"][rows][:, :neb])
            assert np.array_equal(g[0], l2[rows]) and np.array_equal(g[1], crc[rows]) and not g[2].any()
    pa, pb = (got["crc"] == 0) & found, (crc == 0) & found
    both = pa & pb
    clash = int((both & (got["l2"] != l2).any(axis=1)).sum())
    only_d1, only_d1b = int((pa & ~pb).sum()), int((~pa & pb).sum())
    print(f"configs[2] at 100k: pass under both {int(both.sum())}, only D1 {only_d1}, only D1b {only_d1b}, "
          f"both pass with different bits {clash}")
    assert clash == 0                                            # a frame that passes under both is the same frame
    assert np.array_equal(l2[pb], wl["l2"][pb])                  # and what passes is what was sent
    assert only_d1 + only_d1b <= 300                             # <= 0.3 % of the bursts change verdict (Es/N0 6 dB third)
    assert both.sum() > 90_000


@pytest.mark.timeout(900)
def test_bench_nt3_100k_decoder_choice(gpu_api, orc, pkg):
    wl = workloads.nt3_mix(pkg, 100_000, seed=5)                       # bench.py --workload nt3's distinct bursts
    sp, fa = wl["speech"], wl["facch"]
    ds = gpu_api.demod_batch("nt3_speech", wl["iq"], wl["offset"][sp], 474, sps=4, freq_shift=wl["freq_shift"][sp],
                             want_ssyms=False)
    df = gpu_api.demod_batch("nt3_facch", wl["iq"], wl["offset"][fa], 474, sps=4, freq_shift=wl["freq_shift"][fa],
                             want_ssyms=False)
    with gpu_api.conv_decoder(gpu_api.CONV_GENERIC):                  # D1
        g_fr = gpu_api.tch3_decode_batch(ds["ebits"], 0)
        g_fa = gpu_api.facch3_decode_batch(df["ebits"].reshape(-1, 4, 104))
    with orc.conv_mode(1):
        o_fr = orc.tch3_decode(ds["ebits"], 0)
        o_fa = orc.facch3_decode(df["ebits"].reshape(-1, 4, 104))
    with gpu_api.conv_decoder(gpu_api.CONV_ACC):
        a_fr = gpu_api.tch3_decode_batch(ds["ebits"], 0)
        a_fa = gpu_api.facch3_decode_batch(df["ebits"].reshape(-1, 4, 104))
    for k in (0, 1):
        assert np.array_equal(a_fr[k], o_fr[k]), "TCH3 frames under the accelerated decoder differ from the oracle's D1b"
    assert np.array_equal(a_fa[0], o_fa[0]) and np.array_equal(a_fa[2], o_fa[2])
    # speech: no CRC -- class-1 bits (6 bytes per frame) against what was sent, under each decoder
    one_sided = differ = recovered = 0
    for k in (0, 1):
        ra = (g_fr[k][:, :6] == wl["frames"][:, k, :6]).all(axis=1)
        rb = (o_fr[k][:, :6] == wl["frames"][:, k, :6]).all(axis=1)
        one_sided += int((ra != rb).sum())
        recovered += int((ra & rb).sum())
        differ += int((g_fr[k][:, :6] != o_fr[k][:, :6]).any(axis=1).sum())
        assert np.array_equal(g_fr[k][:, 6:], o_fr[k][:, 6:])
    pa, pb = g_fa[2] == 0, o_fa[2] == 0
    clash = int((pa & pb & (g_fa[0] != o_fa[0]).any(axis=1)).sum())
    print(f"configs[4] NT3 at 100k: speech frames returned as sent by both {recovered} of {2 * sp.size}, by one only "
          f"{one_sided}, decoded differently {differ}; FACCH3 groups: pass under both {int((pa & pb).sum())} of {pa.size}, "
          f"only D1 {int((pa & ~pb).sum())}, only D1b {int((~pa & pb).sum())}, clash {clash}")
    assert clash == 0
    assert one_sided <= 0.01 * 2 * sp.size
    assert int((pa != pb).sum()) <= 0.01 * pa.size
