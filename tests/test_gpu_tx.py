"""Transmit direction on the GPU (tx_kernels.hip through the C ABI): every channel encoder and the modulator
against the CPU oracle, bit for bit; then what only the two directions together can show -- encode -> modulate
-> demodulate -> decode round trips entirely on the GPU, at the benchmark's full batch size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = (1, 5, 1003)          # one unit, less than a workgroup, not a multiple of the four units per workgroup


def hard_soft(ubits):
    """ubit -> confident soft bit (0 -> +127, 1 -> -127)"""
    return np.where(np.asarray(ubits) & 1, -127, 127).astype(np.int8)


@pytest.mark.parametrize("n", SIZES)
def test_bcch_ccch_xch_encode_match_oracle(gpu_api, orc, n):
    rng = np.random.default_rng(100 + n)
    l2 = rng.integers(0, 256, (n, 24), dtype=np.uint8)
    l2[0] = 0
    m = min(n, 60)
    assert np.array_equal(gpu_api.bcch_encode_batch(l2)[:m], orc.bcch_encode(l2[:m]))
    assert np.array_equal(gpu_api.ccch_encode_batch(l2)[:m], orc.ccch_encode(l2[:m]))
    assert np.array_equal(gpu_api.xch_dc12_encode_batch(l2)[:m], np.stack([orc.xch_dc12_encode(x) for x in l2[:m]]))
    # the rest of the batch: every burst decodes back to its payload
    got, crc, _ = gpu_api.bcch_decode_batch(hard_soft(gpu_api.bcch_encode_batch(l2)))
    assert not crc.any() and np.array_equal(got, l2)


@pytest.mark.parametrize("n", SIZES)
def test_facch3_tch3_encode_match_oracle(gpu_api, orc, n):
    rng = np.random.default_rng(200 + n)
    m = min(n, 40)
    l2 = rng.integers(0, 256, (n, 10), dtype=np.uint8)
    bits_s = rng.integers(0, 2, (n, 32), dtype=np.uint8)
    ciph = rng.integers(0, 2, (n, 384), dtype=np.uint8)
    assert np.array_equal(gpu_api.facch3_encode_batch(l2, bits_s)[:m], orc.facch3_encode(l2[:m], bits_s[:m]))
    assert np.array_equal(gpu_api.facch3_encode_batch(l2, bits_s, ciph)[:m], orc.facch3_encode(l2[:m], bits_s[:m], ciph[:m]))
    fr = rng.integers(0, 256, (n, 2, 10), dtype=np.uint8)
    st = rng.integers(0, 2, (n, 4), dtype=np.uint8)
    c3 = rng.integers(0, 2, (n, 208), dtype=np.uint8)
    for mm in (0, 1):
        assert np.array_equal(gpu_api.tch3_encode_batch(fr, st, mm)[:m], orc.tch3_encode(fr[:m, 0], fr[:m, 1], st[:m], mm))
        assert np.array_equal(gpu_api.tch3_encode_batch(fr, st, mm, c3)[:m],
                              orc.tch3_encode(fr[:m, 0], fr[:m, 1], st[:m], mm, c3[:m]))
        # round trip through the GPU decoder, ciphered
        d0, d1, ds, _, _ = gpu_api.tch3_decode_batch(hard_soft(gpu_api.tch3_encode_batch(fr, st, mm, c3)), m=mm, ciph=c3)
        assert np.array_equal(d0, fr[:, 0]) and np.array_equal(d1, fr[:, 1]) and np.array_equal(ds, st)


@pytest.mark.parametrize("n", (1, 6, 202))
def test_nt9_encode_match_oracle(gpu_api, orc, n):
    rng = np.random.default_rng(300 + n)
    m = min(n, 20)
    l2 = rng.integers(0, 256, (n, 38), dtype=np.uint8)
    l2[:, 37] &= 0x0f
    sa = rng.integers(0, 2, (n, 10), dtype=np.uint8)
    stt = rng.integers(0, 2, (n, 4), dtype=np.uint8)
    ciph = rng.integers(0, 2, (n, 658), dtype=np.uint8)
    got = gpu_api.facch9_encode_batch(l2, sa, stt, ciph)
    assert np.array_equal(got[:m], np.stack([orc.facch9_encode(l2[i], sa[i], stt[i], ciph[i]) for i in range(m)]))
    got = gpu_api.facch9_encode_batch(l2, sa, stt)
    assert np.array_equal(got[:m], np.stack([orc.facch9_encode(l2[i], sa[i], stt[i]) for i in range(m)]))
    seq = {1: 1, 6: 3, 202: 101}[n]                       # runs shorter than, equal to, longer than the interleaver depth
    for mode in (0, 1, 2):
        nb = (18, 30, 60)[mode]
        p = rng.integers(0, 256, (n, nb), dtype=np.uint8)
        got = gpu_api.tch9_encode_batch(p, mode, seq, sa, stt, ciph)
        ref = np.concatenate([orc.tch9_encode_seq(p[r:r + seq], mode, sa[r:r + seq], stt[r:r + seq], ciph[r:r + seq])
                              for r in range(0, n, seq)])
        assert np.array_equal(got, ref), mode
    z10, z4 = np.zeros((5, 10), np.uint8), np.zeros((5, 4), np.uint8)
    with pytest.raises(gpu_api.Gmr1HipError):
        gpu_api.tch9_encode_batch(np.zeros((5, 18), np.uint8), 0, 3, z10, z4)               # not whole runs
    with pytest.raises(gpu_api.Gmr1HipError):
        gpu_api.tch9_encode_batch(np.zeros((3, 18), np.uint8), 3, 3, z10[:3], z4[:3])       # mode


def test_tch9_stateful_encode_then_decode(gpu_api, orc):
    """gmr1_tch9_encode burst by burst (the reference's stateful call) = the oracle's sequence = the batch form; fed to
    the stateful decoder, every block comes back two bursts later."""
    rng = np.random.default_rng(7)
    for mode in (0, 2):
        nb = (18, 30, 60)[mode]
        n = 6
        p = rng.integers(0, 256, (n, nb), dtype=np.uint8)
        sa = rng.integers(0, 2, (n, 10), dtype=np.uint8)
        stt = rng.integers(0, 2, (n, 4), dtype=np.uint8)
        ciph = rng.integers(0, 2, (n, 658), dtype=np.uint8)
        enc = gpu_api.Tch9Encoder(mode)
        got = np.stack([enc.encode(p[i], sa[i], stt[i], ciph[i]) for i in range(n)])
        enc.close()
        assert np.array_equal(got, orc.tch9_encode_seq(p, mode, sa, stt, ciph))
        ch = gpu_api.Tch9Channel(mode)
        for i in range(n):
            l2, _, _, _ = ch.decode(hard_soft(got[i]), ciph[i])
            if i >= 2:
                assert np.array_equal(l2, p[i - 2])
        ch.close()


@pytest.mark.parametrize("n", SIZES)
def test_rach_encode_matches_oracle(gpu_api, orc, n):
    rng = np.random.default_rng(400 + n)
    rach = rng.integers(0, 256, (n, 18), dtype=np.uint8)
    rach[:, 17] &= 0x07
    sb = rng.integers(0, 256, n, dtype=np.uint8)
    got = gpu_api.rach_encode_batch(rach, sb)
    m = min(n, 60)
    assert np.array_equal(got[:m], np.stack([orc.rach_encode(rach[i], int(sb[i])) for i in range(m)]))
    dec, rv, _, _ = gpu_api.rach_decode_batch(hard_soft(got), sb)
    assert not rv.any() and np.array_equal(dec, rach)


def test_single_calls_match_oracle(gpu_api, orc):
    rng = np.random.default_rng(9)
    l2 = rng.integers(0, 256, 24, dtype=np.uint8)
    assert np.array_equal(gpu_api.encode_single("bcch", l2), orc.bcch_encode(l2[None])[0])
    assert np.array_equal(gpu_api.encode_single("ccch", l2), orc.ccch_encode(l2[None])[0])
    assert np.array_equal(gpu_api.encode_single("xch_dc12", l2), orc.xch_dc12_encode(l2))
    f = rng.integers(0, 256, 10, dtype=np.uint8)
    bs = rng.integers(0, 2, 32, dtype=np.uint8)
    c = rng.integers(0, 2, 384, dtype=np.uint8)
    assert np.array_equal(gpu_api.encode_single("facch3", f, bs, c), orc.facch3_encode(f[None], bs[None], c[None])[0].reshape(-1))
    assert np.array_equal(gpu_api.encode_single("facch3", f, bs, None), orc.facch3_encode(f[None], bs[None])[0].reshape(-1))
    f0, f1 = rng.integers(0, 256, (2, 10), dtype=np.uint8)
    st = rng.integers(0, 2, 4, dtype=np.uint8)
    c3 = rng.integers(0, 2, 208, dtype=np.uint8)
    for m in (0, 1):
        assert np.array_equal(gpu_api.encode_single("tch3", f0, f1, st, c3, m),
                              orc.tch3_encode(f0[None], f1[None], st[None], m, c3[None])[0])
    l9 = rng.integers(0, 256, 38, dtype=np.uint8)
    sa = rng.integers(0, 2, 10, dtype=np.uint8)
    assert np.array_equal(gpu_api.encode_single("facch9", l9, sa, st, None), orc.facch9_encode(l9, sa, st))
    r = rng.integers(0, 256, 18, dtype=np.uint8)
    assert np.array_equal(gpu_api.encode_single("rach", r, 0xa5), orc.rach_encode(r, 0xa5))


def test_mod_matches_oracle_every_burst_format(gpu_api, orc, pkg):
    rng = np.random.default_rng(21)
    for name in pkg.api.BURST_IDS:
        info = gpu_api.burst_info(name)
        for sid in range(info.n_sync):
            eb = rng.integers(0, 2, (3, info.ebits), dtype=np.uint8)
            got = gpu_api.mod_batch(name, eb, sid)
            ref = np.stack([orc.mod(name, eb[i], sid) for i in range(3)])
            assert got.shape == ref.shape == (3, info.len)
            assert np.array_equal(got, ref), (name, sid)      # the rotation table comes from the host's cosf / sinf: bit for bit
        with pytest.raises(gpu_api.Gmr1HipError):
            gpu_api.mod_batch(name, np.zeros((1, info.ebits), np.uint8), info.n_sync)


def test_pi4cxpsk_mod_reference_call(gpu_api, orc):
    rng = np.random.default_rng(22)
    info = gpu_api.burst_info("bcch")
    eb = rng.integers(0, 2, info.ebits, dtype=np.uint8)
    rc, syms = gpu_api.pi4cxpsk_mod("bcch", eb, 0)
    assert rc == 0 and syms.size == info.len
    assert np.array_equal(syms, orc.mod("bcch", eb, 0))
    rc, _ = gpu_api.pi4cxpsk_mod("bcch", eb, 0, max_len=info.len - 1)
    assert rc == -12                                                    # -ENOMEM, pi4cxpsk.c:752-756


@pytest.mark.parametrize("chain,burst", [("bcch", "bcch"), ("ccch", "dc6"), ("xch_dc12", "dc12"), ("rach", "rach")])
def test_full_size_round_trip_through_samples(gpu_api, pkg, chain, burst):
    """encode -> modulate -> demodulate -> decode, every stage on the GPU, 100 000 bursts (20 000 for the K = 9
    / RACH chains): every payload comes back and every CRC passes."""
    n = 100_000 if chain in ("bcch", "ccch") else 20_000
    rng = np.random.default_rng(31)
    info = gpu_api.burst_info(burst)
    if chain == "rach":
        pay = rng.integers(0, 256, (n, 18), dtype=np.uint8)
        pay[:, 17] &= 0x07
        sb = rng.integers(0, 256, n, dtype=np.uint8)
        eb = gpu_api.rach_encode_batch(pay, sb)
    else:
        pay = rng.integers(0, 256, (n, 24), dtype=np.uint8)
        eb = getattr(gpu_api, chain + "_encode_batch")(pay)
    assert eb.shape == (n, info.ebits)
    syms = gpu_api.mod_batch(burst, eb, 0)
    if chain in ("bcch", "ccch"):
        # the modulator's own output, one sample per symbol, straight into the demodulator
        sps, win, at = 1, 4, 2
        iq = np.zeros((n, info.len + win), np.complex64)
        iq[:, at:at + info.len] = syms
    else:
        # the long DC12 / RACH bursts need the pulse the demodulator expects: raised cosine at 4 samples per symbol
        sps, win, at, span = 4, 16, 8, 5
        shaped = pkg.synth.shape_bursts(syms, sps, 0.0, span=span)
        iq = np.ascontiguousarray(shaped[:, span * sps - at:span * sps - at + info.len * sps + win])
    in_len = info.len * sps + win
    assert iq.shape == (n, in_len)
    off = np.arange(n, dtype=np.uint64) * np.uint64(in_len)
    d = gpu_api.demod_batch(burst, iq, off, in_len, sps=sps, want_ssyms=False)
    assert not d["rv"].any()
    assert np.array_equal(d["ebits"] < 0, eb.astype(bool))
    if chain == "rach":
        dec, rv, _, _ = gpu_api.rach_decode_batch(d["ebits"], sb)
        assert not rv.any() and np.array_equal(dec, pay)
    elif chain == "xch_dc12":
        l2, crc, _ = gpu_api.xch_dc12_decode_batch(d["ebits"])
        assert not crc.any() and np.array_equal(l2, pay)
    else:
        l2, crc, _ = getattr(gpu_api, chain + "_decode_batch")(d["ebits"])
        assert not crc.any() and np.array_equal(l2, pay)


def test_encode_rejects_bad_arguments(gpu_api):
    import ctypes as C
    L = gpu_api.load()
    e = np.zeros(424, np.uint8)
    assert L.gmr1_hip_bcch_encode_batch(C.c_int(1), None, e.ctypes.data_as(C.c_void_p)) == -22
    assert L.gmr1_hip_bcch_encode_batch(C.c_int(-1), e.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p)) == -22
    assert L.gmr1_hip_bcch_encode_batch(C.c_int(0), None, None) == 0
    l2 = np.zeros(10, np.uint8)
    assert L.gmr1_hip_facch3_encode_batch(C.c_int(1), l2.ctypes.data_as(C.c_void_p), None, None,
                                          e.ctypes.data_as(C.c_void_p)) == -22       # status bits are required


def test_standalone_primitives_match_oracle(gpu_api, orc):
    """gmr1_scramble_{sbit,ubit}, gmr1_{de,}interleave_intra, gmr1_{de,}interleave_inter as stand-alone calls (each one
    trip to the GPU) against the oracle's restatement of the same reference functions."""
    import ctypes as C
    O = orc.lib()
    rng = np.random.default_rng(77)

    def ocall(fn, out, *args):
        getattr(O, fn)(out.ctypes.data_as(C.c_void_p), *args)
        return out

    for n in (1, 96, 432, 1000):
        sb = rng.integers(-128, 128, n, dtype=np.int8)
        ub = rng.integers(0, 2, n, dtype=np.uint8)
        assert np.array_equal(gpu_api.scramble_sbit(sb), ocall("orc_scramble_sbit", np.zeros(n, np.int8),
                                                               sb.ctypes.data_as(C.c_void_p), C.c_int(n)))
        assert np.array_equal(gpu_api.scramble_ubit(ub), ocall("orc_scramble_ubit", np.zeros(n, np.uint8),
                                                               ub.ctypes.data_as(C.c_void_p), C.c_int(n)))
        assert np.array_equal(gpu_api.scramble_ubit(gpu_api.scramble_ubit(ub)), ub)
    for N in (12, 53, 81):
        x = rng.integers(0, 256, 8 * N, dtype=np.uint8)
        y = gpu_api.interleave_intra(x, N)
        assert np.array_equal(y, ocall("orc_interleave_intra", np.zeros(8 * N, np.uint8), x.ctypes.data_as(C.c_void_p), C.c_int(N)))
        assert np.array_equal(gpu_api.interleave_intra(y, N, inverse=True), x)
        assert np.array_equal(gpu_api.interleave_intra(x, N, inverse=True),
                              ocall("orc_deinterleave_intra", np.zeros(8 * N, np.uint8), x.ctypes.data_as(C.c_void_p), C.c_int(N)))
    # inter-burst: a run of bursts through the interleaver, then through the de-interleaver: block n comes back at n + 2
    oil, odl = orc.Interleaver(), orc.Interleaver()
    O.orc_interleaver_init(C.byref(oil), C.c_int(3), C.c_int(648))
    O.orc_interleaver_init(C.byref(odl), C.c_int(3), C.c_int(648))
    gi, gd = gpu_api.InterBurstInterleaver(), gpu_api.InterBurstInterleaver()
    blocks = rng.integers(0, 256, (7, 648), dtype=np.uint8)
    for i in range(7):
        ref = np.zeros(648, np.uint8)
        O.orc_interleave_inter(C.byref(oil), ref.ctypes.data_as(C.c_void_p), blocks[i].ctypes.data_as(C.c_void_p))
        got = gi.interleave(blocks[i])
        assert np.array_equal(got, ref), i
        ref2 = np.zeros(648, np.uint8)
        O.orc_deinterleave_inter(C.byref(odl), ref2.ctypes.data_as(C.c_void_p), ref.ctypes.data_as(C.c_void_p))
        got2 = gd.deinterleave(got)
        assert np.array_equal(got2, ref2), i
        if i >= 2:
            assert np.array_equal(got2, blocks[i - 2])
    gi.close()
    gd.close()


def test_encoders_linear_at_full_size(gpu_api):
    """Size-independent property at the benchmark's batch size: every channel coder is GF(2)-affine, so over 100 000
    random payload pairs encode(x ^ y) = encode(x) ^ encode(y) ^ encode(0) bit for bit (BCCH, xCH / DC12, RACH incl. the
    SB mask, TCH9 with its inter-burst interleaver in runs of 50)."""
    n = 100_000
    rng = np.random.default_rng(41)
    x = rng.integers(0, 256, (n, 24), dtype=np.uint8)
    y = rng.integers(0, 256, (n, 24), dtype=np.uint8)
    z0 = np.zeros((1, 24), np.uint8)
    for enc in (gpu_api.bcch_encode_batch, gpu_api.xch_dc12_encode_batch):
        assert np.array_equal(enc(x ^ y), enc(x) ^ enc(y) ^ enc(z0))
    rx, ry = x[:, :18], y[:, :18]
    mx, my = x[:, 18], y[:, 19]
    e0 = gpu_api.rach_encode_batch(np.zeros((1, 18), np.uint8), np.zeros(1, np.uint8))
    assert np.array_equal(gpu_api.rach_encode_batch(rx ^ ry, mx ^ my),
                          gpu_api.rach_encode_batch(rx, mx) ^ gpu_api.rach_encode_batch(ry, my) ^ e0)
    m = 20_000
    px = rng.integers(0, 256, (m, 60), dtype=np.uint8)
    py = rng.integers(0, 256, (m, 60), dtype=np.uint8)
    sa = rng.integers(0, 2, (m, 10), dtype=np.uint8)
    st = rng.integers(0, 2, (m, 4), dtype=np.uint8)
    zs, zt = np.zeros((m, 10), np.uint8), np.zeros((m, 4), np.uint8)
    f = lambda p, a, b: gpu_api.tch9_encode_batch(p, 2, 50, a, b)
    assert np.array_equal(f(px ^ py, sa, st), f(px, sa, st) ^ f(py, zs, zt) ^ f(np.zeros((m, 60), np.uint8), zs, zt))
