"""GPU parity of the NT9 burst decoders (FACCH9, TCH9 in its three modes) against the oracle
(oracle/orc_nt9.c; reference src/l1/facch9.c, tch9.c): bit-exact L2, CRC verdict and conv_rv for the same
soft bits -- clean, noisy, erased, pure noise, all-equal (tie-heavy) inputs, with and without deciphering."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _soft(rng, hard, kind):
    sb = (127 * (1 - 2 * hard.astype(np.int16))).astype(np.int16)
    if kind == "clean":
        return sb.astype(np.int8)
    if kind == "noisy":
        v = sb * 0.35 + rng.normal(0, 40.0, sb.shape)
        return np.clip(np.round(v), -127, 127).astype(np.int8)
    if kind == "erased":
        out = sb.copy()
        out[rng.random(sb.shape) < 0.25] = 0
        return out.astype(np.int8)
    if kind == "noise":
        return rng.integers(-128, 128, sb.shape).astype(np.int8)
    if kind == "ties":
        return np.full(sb.shape, 5, np.int8)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["clean", "noisy", "erased", "noise", "ties"])
def test_facch9_bit_exact(gpu_api, orc, kind, decoder):
    rng = np.random.default_rng(3)
    n = 37
    hard = np.zeros((n, 662), np.uint8)
    l2 = rng.integers(0, 256, size=(n, 38), dtype=np.uint8)
    l2[:, 37] &= 0x0F
    ciph = rng.integers(0, 2, size=(n, 658), dtype=np.uint8)
    for i in range(n):
        hard[i] = orc.facch9_encode(l2[i], rng.integers(0, 2, 10), rng.integers(0, 2, 4), ciph[i])
    sb = _soft(rng, hard, kind)
    for use_c in (True, False):
        c = ciph if use_c else None
        g = gpu_api.facch9_decode_batch(sb, c)
        for i in range(n):
            o = orc.facch9_decode(sb[i], None if c is None else c[i])
            assert np.array_equal(g[0][i], o[0]), (kind, use_c, i)
            assert np.array_equal(g[1][i], o[1]) and np.array_equal(g[2][i], o[2])
            assert (g[3][i] != 0) == (o[3] != 0) and g[4][i] == o[4], (kind, use_c, i, g[4][i], o[4])
        if kind == "clean" and use_c:
            assert not g[3].any() and np.array_equal(g[0], l2)
    one = gpu_api.facch9_decode(sb[0], ciph[0])
    ref = orc.facch9_decode(sb[0], ciph[0])
    assert np.array_equal(one[0], ref[0]) and (one[3] != 0) == (ref[3] != 0) and one[4] == ref[4]


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("kind", ["clean", "noisy", "erased", "noise", "ties"])
def test_tch9_bit_exact(gpu_api, orc, mode, kind, decoder):
    rng = np.random.default_rng(10 * mode + 1)
    n_chan, seq = 3, 7
    nb = orc.TCH9_BYTES[mode]
    eb = np.zeros((n_chan * seq, 662), np.int8)
    ciph = rng.integers(0, 2, size=(n_chan * seq, 658), dtype=np.uint8)
    sent = []
    for ch in range(n_chan):
        l2 = rng.integers(0, 256, size=(seq, nb), dtype=np.uint8)
        sent.append(l2)
        hard = orc.tch9_encode_seq(l2, mode, rng.integers(0, 2, (seq, 10)), rng.integers(0, 2, (seq, 4)),
                                   ciph[ch * seq:(ch + 1) * seq])
        eb[ch * seq:(ch + 1) * seq] = _soft(rng, hard, kind)
    for use_c in (True, False):
        c = ciph if use_c else None
        g = gpu_api.tch9_decode_batch(eb, mode, seq, c)
        for ch in range(n_chan):
            sl = slice(ch * seq, (ch + 1) * seq)
            o = orc.tch9_decode_seq(eb[sl], mode, None if c is None else c[sl])
            assert np.array_equal(g[0][sl], o[0]), (mode, kind, use_c, ch)
            assert np.array_equal(g[1][sl], o[1]) and np.array_equal(g[2][sl], o[2])
            assert np.array_equal(g[3][sl], o[3]), (mode, kind, use_c, ch, g[3][sl], o[3])
            if kind == "clean" and use_c:
                assert np.array_equal(g[0][sl][2:], sent[ch][:-2])


@pytest.mark.parametrize("mode", [0, 2])
def test_tch9_stateful_reference_call(gpu_api, orc, mode, decoder):
    """gmr1_interleaver_init + gmr1_tch9_decode burst by burst, as gmr1_rx.c:273 / :333 call them, against the
    oracle's stateful decoder (tch9.c:139-175 + interleave.c:163-186)."""
    rng = np.random.default_rng(40 + mode)
    seq, nb = 9, orc.TCH9_BYTES[mode]
    l2 = rng.integers(0, 256, size=(seq, nb), dtype=np.uint8)
    ciph = rng.integers(0, 2, size=(seq, 658), dtype=np.uint8)
    hard = orc.tch9_encode_seq(l2, mode, rng.integers(0, 2, (seq, 10)), rng.integers(0, 2, (seq, 4)), ciph)
    eb = _soft(rng, hard, "noisy")
    o = orc.tch9_decode_seq(eb, mode, ciph)
    ch = gpu_api.Tch9Channel(mode)
    for i in range(seq):
        g_l2, g_sa, g_st, g_conv = ch.decode(eb[i], ciph[i])
        assert np.array_equal(g_l2, o[0][i]) and g_conv == o[3][i], (mode, i)
        assert np.array_equal(g_sa, o[1][i]) and np.array_equal(g_st, o[2][i])
    ch.close()
    # without a key stream, and a geometry GMR-1 does not use
    ch = gpu_api.Tch9Channel(mode)
    o = orc.tch9_decode_seq(eb[:4], mode, None)
    for i in range(4):
        assert np.array_equal(ch.decode(eb[i])[0], o[0][i])
    ch.close()
    with pytest.raises(Exception):
        gpu_api.Tch9Channel(mode, N=4, K=648)


def test_nt9_argument_checks(gpu_api):
    eb = np.zeros((6, 662), np.int8)
    with pytest.raises(Exception):
        gpu_api.tch9_decode_batch(eb, 3, 3)            # no such mode
    l2, sa, stt, conv = gpu_api.tch9_decode_batch(eb[:0], 1, 1)
    assert l2.shape == (0, 30)
