"""CPU tests of the oracle's xCH-over-DC12 and RACH codecs (oracle/orc_xch.c; reference src/l1/xch_dc12.c,
rach.c): known answers derivable from the reference's own code, agreement with the independent numpy
encoders of the signal generator, encode -> decode round trips."""
import importlib
import json
import os

import numpy as np


def _golden():
    with open(os.path.join(os.path.dirname(__file__), "golden", "known_answers.json")) as f:
        return json.load(f)


def test_k9_13_code_rows_and_puncturing(pkg):
    synth = importlib.import_module(pkg.__name__ + ".synth")
    spot = _golden()["conv_spot"]["k9_13"]
    assert list(synth.K9_13) == spot["polys"]
    for s, (o0, o1) in spot["next_output_rows"].items():
        for b, want in ((0, o0), (1, o1)):
            reg = (int(s) << 1) | b
            got = sum((bin(reg & g).count("1") & 1) << (2 - j) for j, g in enumerate(synth.K9_13))
            assert got == want, (s, b)
    # P(12;13) over 208 x 3 coded bits leaves exactly the 432 bits of a DC12 burst
    assert int(np.tile(synth.P1213, 16).sum()) == 432 and _golden()["xch_dc12_punctured_first13"] == \
        [int(i) for i in np.flatnonzero(np.tile(synth.P1213, 16) == 0)[:13]]


def test_xch_dc12_round_trip(orc, pkg):
    synth = importlib.import_module(pkg.__name__ + ".synth")
    rng = np.random.default_rng(3)
    for trial in range(8):
        l2 = rng.integers(0, 256, 24, dtype=np.uint8)
        e = orc.xch_dc12_encode(l2)
        assert np.array_equal(e, synth.xch_dc12_encode(l2[None])[0])      # two independent encoders
        sb = (127 * (1 - 2 * e.astype(np.int16))).astype(np.int8)
        out, crc, conv = orc.xch_dc12_decode(sb)
        assert crc == 0 and conv == 0 and np.array_equal(out, l2)
        sb2 = sb.copy()
        sb2[rng.choice(432, 10, replace=False)] *= -1                      # a few flipped soft bits
        out, crc, conv = orc.xch_dc12_decode(sb2)
        assert crc == 0 and conv > 0 and np.array_equal(out, l2)
    # no signal: the CRC says so
    assert orc.xch_dc12_decode(rng.integers(-127, 128, 432).astype(np.int8))[1] != 0


def test_rach_round_trip_and_sb_mask(orc, pkg):
    synth = importlib.import_module(pkg.__name__ + ".synth")
    rng = np.random.default_rng(4)
    for trial in range(8):
        rach = rng.integers(0, 256, 18, dtype=np.uint8)
        rach[17] &= 7                                                      # 16 + 123 bits
        mask = int(rng.integers(1, 256))
        e = orc.rach_encode(rach, mask)
        assert np.array_equal(e, synth.rach_encode(rach[None], mask)[0])
        # the class-1 part is sent twice (rach.c:112-114), scrambled differently
        sb = (127 * (1 - 2 * e.astype(np.int16))).astype(np.int8)
        out, rv, conv, crc = orc.rach_decode(sb, mask)
        assert rv == 0 and crc == (0, 0) and conv == 0 and np.array_equal(out, rach)
        # a wrong SB mask fails CRC8 only; the payload still comes out
        out, rv, conv, crc = orc.rach_decode(sb, mask ^ 0x21)
        assert rv == 1 and crc == (1, 0) and np.array_equal(out, rach)
        # rach.c:176-184 checks CRC8 as received first: a burst sent with mask 0 passes under any mask
        e0 = orc.rach_encode(rach, 0)
        sb0 = (127 * (1 - 2 * e0.astype(np.int16))).astype(np.int8)
        assert orc.rach_decode(sb0, mask)[1] == 0
        sb2 = sb.copy()
        sb2[rng.choice(494, 10, replace=False)] *= -1
        out, rv, conv, crc = orc.rach_decode(sb2, mask)
        assert rv == 0 and conv == 0 and np.array_equal(out, rach)       # the accelerated decoder (default) returns no metric
        with orc.conv_mode(0):
            out, rv, conv, crc = orc.rach_decode(sb2, mask)
        assert rv == 0 and conv > 0 and np.array_equal(out, rach)        # the generic one returns the corrected errors' cost
