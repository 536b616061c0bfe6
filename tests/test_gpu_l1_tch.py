"""GPU parity tests of the FACCH3 / TCH3 layer-1 decoders vs the CPU oracle (bit-exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _soft(bits, amp=127):
    return (amp * (1 - 2 * bits.astype(np.int16))).astype(np.int8)


def _variants(rng, bits):
    clean = _soft(bits)
    noisy = np.clip(55 * (1 - 2 * bits.astype(np.int16)) + rng.normal(0, 45, bits.shape), -128, 127).astype(np.int8)
    noisy[rng.random(bits.shape) < 0.05] = 0
    junk = rng.integers(-128, 128, size=bits.shape).astype(np.int8)
    coarse = (rng.integers(-2, 3, size=bits.shape) * 50).astype(np.int8)
    return (("clean", clean), ("noisy", noisy), ("junk", junk), ("coarse", coarse))


def test_facch3_bit_exact(gpu_api, orc, pkg):
    rng = np.random.default_rng(31)
    n = 403                                     # ragged: not a multiple of 4 frames
    l2 = rng.integers(0, 256, (n, 10), dtype=np.uint8)
    l2[:, 9] &= 0x0F
    s = rng.integers(0, 2, (n, 32), dtype=np.uint8)
    e = pkg.synth.facch3_encode(l2, s).reshape(n, 416)
    for tag, eb in _variants(rng, e):
        g = gpu_api.facch3_decode_batch(eb)
        o = orc.facch3_decode(eb)
        assert np.array_equal(g[3], o[3]), f"{tag}: conv_rv"
        assert np.array_equal(g[2], o[2]), f"{tag}: crc"
        assert np.array_equal(g[0], o[0]), f"{tag}: l2"
        assert np.array_equal(g[1], o[1]), f"{tag}: status bits"
        if tag == "clean":
            assert np.array_equal(g[0], l2) and np.array_equal(g[1], s) and not g[2].any()
    # the reference's single call
    out, sb, rv, conv = gpu_api.facch3_decode(_soft(e[3]))
    assert rv == 0 and np.array_equal(out, l2[3]) and np.array_equal(sb, s[3]) and conv == 0


def test_tch3_bit_exact(gpu_api, orc, pkg):
    rng = np.random.default_rng(32)
    n = 301
    f0 = rng.integers(0, 256, (n, 10), dtype=np.uint8)
    f1 = rng.integers(0, 256, (n, 10), dtype=np.uint8)
    s = rng.integers(0, 2, (n, 4), dtype=np.uint8)
    for m in (0, 1):
        e = pkg.synth.tch3_encode(f0, f1, s, m)
        for tag, eb in _variants(rng, e):
            g = gpu_api.tch3_decode_batch(eb, m)
            o = orc.tch3_decode(eb, m)
            for idx, what in enumerate(("frame0", "frame1", "status", "conv0", "conv1")):
                assert np.array_equal(g[idx], o[idx]), f"m={m} {tag}: {what}"
            if tag == "clean":
                assert np.array_equal(g[0], f0) and np.array_equal(g[1], f1) and np.array_equal(g[2], s)
    a0, a1, sb, c0, c1 = gpu_api.tch3_decode(_soft(pkg.synth.tch3_encode(f0[:1], f1[:1], s[:1], 0)[0]), 0)
    assert np.array_equal(a0, f0[0]) and np.array_equal(a1, f1[0]) and np.array_equal(sb, s[0]) and c0 == 0 == c1
