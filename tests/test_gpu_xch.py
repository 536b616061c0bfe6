"""GPU parity of the two layer-1 decoders gmr1_rx does not call against the
oracle: xCH over DC12 (reference src/l1/xch_dc12.c:81-108, K = 9 tail-biting, 256 states) and RACH
(reference src/l1/rach.c:127-200)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _soft(bits, rng, amp, sigma):
    """hard bits -> noisy soft bits (+ = 0), clipped like a demodulator's output"""
    x = (1.0 - 2.0 * bits.astype(np.float64)) * amp + rng.normal(0.0, sigma, bits.shape)
    return np.clip(np.rint(x), -127, 127).astype(np.int8)


def _xch_batch(pkg, n, seed):
    synth = importlib.import_module(pkg.__name__ + ".synth")
    rng = np.random.default_rng(seed)
    l2 = rng.integers(0, 256, size=(n, 24), dtype=np.uint8)
    e = synth.xch_dc12_encode(l2)
    sb = np.empty((n, 432), np.int8)
    q = n // 4
    sb[:q] = _soft(e[:q], rng, 64, 30)            # clean
    sb[q:2 * q] = _soft(e[q:2 * q], rng, 48, 48)  # around the threshold
    sb[2 * q:3 * q] = _soft(e[2 * q:3 * q], rng, 30, 60)   # mostly failing
    sb[3 * q:] = rng.integers(-128, 128, size=(n - 3 * q, 432)).astype(np.int8)   # no signal, incl. -128
    sb[0] = 0                                      # all erasures: every metric ties
    sb[1] = 127
    sb[2] = -128
    return l2, sb


def test_xch_dc12_matches_oracle(gpu_api, orc, pkg, decoder):
    l2, sb = _xch_batch(pkg, 600, 11)
    g_l2, g_crc, g_conv = gpu_api.xch_dc12_decode_batch(sb)
    n_pass = 0
    for i in range(sb.shape[0]):
        o_l2, o_crc, o_conv = orc.xch_dc12_decode(sb[i])
        assert g_crc[i] == o_crc, (i, g_crc[i], o_crc)
        assert g_conv[i] == o_conv, (i, g_conv[i], o_conv)
        assert np.array_equal(g_l2[i], o_l2), i
        if o_crc == 0 and i >= 3:
            n_pass += 1
            assert np.array_equal(g_l2[i], l2[i])
    assert n_pass > 150
    # the reference's own call
    r_l2, r_crc, r_conv = gpu_api.xch_dc12_decode(sb[5])
    assert r_crc == g_crc[5] and r_conv == g_conv[5] and np.array_equal(r_l2, g_l2[5])


def test_rach_matches_oracle(gpu_api, orc, pkg, decoder):
    synth = importlib.import_module(pkg.__name__ + ".synth")
    rng = np.random.default_rng(12)
    n = 803                                        # not a multiple of the four bursts per wavefront
    rach = rng.integers(0, 256, size=(n, 18), dtype=np.uint8)
    rach[:, 17] &= 7
    mask = rng.integers(0, 256, n).astype(np.uint8)
    e = np.concatenate([synth.rach_encode(rach[i:i + 1], int(mask[i])) for i in range(n)])
    sb = np.empty((n, 494), np.int8)
    q = n // 4
    sb[:q] = _soft(e[:q], rng, 64, 25)
    sb[q:2 * q] = _soft(e[q:2 * q], rng, 48, 40)
    sb[2 * q:3 * q] = _soft(e[2 * q:3 * q], rng, 30, 50)
    sb[3 * q:] = rng.integers(-128, 128, size=(n - 3 * q, 494)).astype(np.int8)
    sb[0] = 0
    sb[1] = -128
    use = mask.copy()
    use[q // 2:q] ^= 0x5A                          # decoded with the wrong SB mask: CRC8 must fail, CRC12 not
    g_rach, g_rv, g_conv, g_crc = gpu_api.rach_decode_batch(sb, use)
    n_pass = n_wrong = 0
    for i in range(n):
        o_rach, o_rv, o_conv, o_crc = orc.rach_decode(sb[i], int(use[i]))
        assert g_rv[i] == o_rv and tuple(g_crc[i]) == o_crc, (i, g_rv[i], o_rv, g_crc[i], o_crc)
        assert g_conv[i] == o_conv, (i, g_conv[i], o_conv)
        assert np.array_equal(g_rach[i], o_rach), i
        if o_rv == 0 and i >= 2:
            n_pass += 1
            assert np.array_equal(g_rach[i], rach[i])
        if q // 2 <= i < q and o_crc == (1, 0):
            n_wrong += 1
    assert n_pass > 200 and n_wrong > 0.8 * (q - q // 2)
    r = gpu_api.rach_decode(sb[7], int(use[7]))
    assert r[1] == g_rv[7] and r[2] == g_conv[7] and r[3] == tuple(g_crc[7]) and np.array_equal(r[0], g_rach[7])


@pytest.mark.parametrize("name", ["dc12", "rach"])
def test_burst_to_payload_end_to_end(gpu_api, orc, pkg, name):
    """A DC12 / RACH burst on the air -> gmr1_hip_demod_batch -> layer-1 decode, against the oracle's
    demodulator + decoder on the same samples, and against what was sent."""
    synth = importlib.import_module(pkg.__name__ + ".synth")
    rng = np.random.default_rng(31)
    sps, win, n = 4, 40, 40
    fmt = pkg.api.burst_format(name)
    if name == "dc12":
        sent = rng.integers(0, 256, size=(n, 24), dtype=np.uint8)
        ebits = synth.xch_dc12_encode(sent)
    else:
        sent = rng.integers(0, 256, size=(n, 18), dtype=np.uint8)
        sent[:, 17] &= 7
        ebits = synth.rach_encode(sent, 0xA7)
    assert ebits.shape[1] == fmt.ebits
    sym = synth.map_symbols(fmt, ebits)
    bb = synth.synth_windows(fmt, sym, sps, win, rng, toa_jitter=4, frac=True, cfo_hz_std=20.0, esn0_db=8.0)
    offset = (np.arange(n) * bb.stride).astype(np.uint64)
    dm = gpu_api.demod_batch(name, bb.iq, offset, bb.in_len, sps=sps)
    assert not dm["rv"].any()
    if name == "dc12":
        l2, crc, conv = gpu_api.xch_dc12_decode_batch(dm["ebits"])
    else:
        l2, crc, conv, _ = gpu_api.rach_decode_batch(dm["ebits"], 0xA7)
    assert (crc == 0).mean() > 0.9
    n_same = 0
    for i in range(n):
        o = orc.demod(name, bb.iq[i, :bb.in_len], sps)
        if name == "dc12":
            o_l2, o_crc, _ = orc.xch_dc12_decode(o["ebits"])
        else:
            o_l2, o_crc, _, _ = orc.rach_decode(o["ebits"], 0xA7)
        if crc[i] == 0:
            assert np.array_equal(l2[i], sent[i]), i
        # soft bits may differ by 1 LSB between the two demodulators; the decoded payload does not
        if crc[i] == 0 and o_crc == 0:
            assert np.array_equal(l2[i], o_l2)
            n_same += 1
    assert n_same > 0.85 * n


def test_xch_rach_reject_bad_arguments(gpu_api):
    with pytest.raises(Exception):
        gpu_api.xch_dc12_decode_batch(np.zeros((3, 431), np.int8))
    with pytest.raises(Exception):
        gpu_api.rach_decode_batch(np.zeros((3, 400), np.int8), 0)
    l2, crc, conv = gpu_api.xch_dc12_decode_batch(np.zeros((0, 432), np.int8))
    assert l2.shape == (0, 24)
