"""CPU tests of the oracle's NT9 codecs (oracle/orc_nt9.c; reference src/l1/facch9.c, tch9.c, punct.c,
interleave.c): known answers derivable from the reference's own code, and encode -> decode round trips."""
import numpy as np
import pytest


def test_tch9_puncturing_arrays(orc):
    # 9k6 -> {1, 5} u {10 + 6k, 13 + 6k : k < 158} u {963, 967}: 320 of 968 coded bits
    import json
    import os
    p96 = orc.tch9_punct(2)
    with open(os.path.join(os.path.dirname(__file__), "golden", "known_answers.json")) as f:
        want = json.load(f)["tch9_9k6_punctured"]
    assert list(p96) == want and len(want) == 320
    # every mode leaves exactly the 648 bits of an NT9 burst
    for mode, (n_out, n_p) in enumerate(((148 * 5, None), (244 * 3, None), (484 * 2, 320))):
        p = orc.tch9_punct(mode)
        assert n_out - len(p) == 648
        assert np.all(np.diff(p) > 0) and p[0] >= 0 and p[-1] < n_out


def test_facch9_round_trip_and_demux(orc):
    rng = np.random.default_rng(1)
    for trial in range(6):
        l2 = rng.integers(0, 256, 38, dtype=np.uint8)
        l2[37] &= 0x0F                                   # 300 bits
        sacch = rng.integers(0, 2, 10, dtype=np.uint8)
        status = rng.integers(0, 2, 4, dtype=np.uint8)
        ciph = rng.integers(0, 2, 658, dtype=np.uint8) if trial & 1 else None
        e = orc.facch9_encode(l2, sacch, status, ciph)
        assert e.shape == (662,) and np.array_equal(e[52:56], status)
        sb = (127 * (1 - 2 * e.astype(np.int16))).astype(np.int8)
        out, sa, stt, crc, conv = orc.facch9_decode(sb, ciph)
        assert crc == 0 and conv == 0 and np.array_equal(out, l2)
        assert np.array_equal(sa < 0, sacch.astype(bool)) and np.array_equal(stt < 0, status.astype(bool))
        # a wrong key stream breaks the CRC
        if ciph is not None:
            assert orc.facch9_decode(sb, 1 - ciph)[3] != 0
        # noise: still decodes with a few flipped soft bits
        sb2 = sb.copy()
        sb2[rng.choice(662, 25, replace=False)] *= -1
        assert np.array_equal(orc.facch9_decode(sb2, ciph)[0], l2)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_tch9_round_trip_with_inter_burst_interleaver(orc, mode):
    rng = np.random.default_rng(10 + mode)
    n, nb = 12, orc.TCH9_BYTES[mode]
    l2 = rng.integers(0, 256, size=(n, nb), dtype=np.uint8)
    sacch = rng.integers(0, 2, size=(n, 10), dtype=np.uint8)
    status = rng.integers(0, 2, size=(n, 4), dtype=np.uint8)
    ciph = rng.integers(0, 2, size=(n, 658), dtype=np.uint8)
    e = orc.tch9_encode_seq(l2, mode, sacch, status, ciph)
    sb = (127 * (1 - 2 * e.astype(np.int16))).astype(np.int8)
    out, sa, stt, conv = orc.tch9_decode_seq(sb, mode, ciph)
    # depth-3 diagonal interleaving: block i comes out of burst i + 2 (interleave.c:132-186)
    assert np.array_equal(out[2:], l2[:-2]) and not conv[2:].any()
    assert np.array_equal(sa < 0, sacch.astype(bool)) and np.array_equal(stt < 0, status.astype(bool))
