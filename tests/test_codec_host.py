"""AMBE decoder, host side (no GPU): the tables the library computes when it loads, the exported calls' behaviour
without a device, and (container only) the codebooks against the reference's src/codec/tables.c read as text."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api(pkg):
    pkg.api.load()
    return pkg.api


@pytest.fixture(scope="module")
def tab(api):
    return api.codec_host_tables()


def _orc_f(name, *args):
    f = getattr(oracle_lib.lib(), name)
    f.restype = C.c_float
    return np.float32(f(*args))


def test_cosine_table_window_and_log2(tab):
    want = np.array([_orc_f("orc_ambe_cos_entry", C.c_int(i)) for i in range(1024)], np.float32)
    assert np.array_equal(tab["cosv"].view(np.uint32), want.view(np.uint32))
    k = np.array([i if i < 40 else (120 - i if i > 80 else 40) for i in range(121)], np.float32)
    assert np.array_equal(tab["win"][:121], (k * np.float32(25)) / np.float32(1000)) and not tab["win"][121:].any()
    for L in range(9, 57):
        assert tab["log2_L"][L] == _orc_f("orc_ambe_log2_int", C.c_int(L))


def test_fundamental_tables_are_the_oracles_powf_on_the_oracles_arguments(tab):
    f0log = [_orc_f("orc_ambe_f0log_sf1", C.c_int(p)) for p in range(128)] + [np.float32(0.0)]
    for p in range(128):
        assert tab["f0_sf1"][p] == _orc_f("orc_ambe_pow2", C.c_float(f0log[p]))
    rng = np.random.default_rng(0)
    picks = [(b, p, r) for b in (0, 1, 63, 127, 128) for p in (0, 1, 64, 123, 127) for r in range(4)]
    picks += [tuple(int(x) for x in rng.integers(0, (129, 128, 4))) for _ in range(3000)]
    picks += [(p, p, r) for p in range(128) for r in range(4)]          # equal pitches: the other branch
    for b, p, r in picks:
        x = _orc_f("orc_ambe_f0log_sf0", C.c_float(f0log[b]), C.c_float(f0log[p]), C.c_int(r))
        assert tab["f0_sf0"][b, p, r] == _orc_f("orc_ambe_pow2", C.c_float(x)), (b, p, r)


def test_tone_amplitudes_and_generator_jumps(tab):
    for a in range(256):
        assert tab["tone_ampl"][a] == oracle_lib.lib().orc_ambe_tone_ampl(C.c_int(a))
    for x0 in (3147, 0, 53124, 65535):
        x = x0
        for i in range(121):
            x = (x * 171 + 11213) % 53125 if i else (x0 * 171 + 11213) % (1 << 32) % 53125
            assert (int(tab["lcg_mul"][i]) * x0 + int(tab["lcg_add"][i])) % 53125 == x
            assert int(tab["lcg_mul"][i]) * 65535 + int(tab["lcg_add"][i]) < 1 << 32     # the kernel's 32-bit product


def test_codebooks_equal_the_oracles(tab):
    lib = oracle_lib.lib()

    def orc(name, n, ctype=C.c_uint32):
        return np.ctypeslib.as_array((ctype * n).in_dll(lib, "orc_ambe_" + name)).copy()
    for name, n in (("gain", 512), ("prba12", 256), ("prba34", 128), ("prba57", 384), ("sf0_perr14", 256),
                    ("sf0_perr58", 128), ("sf0_interp", 4), ("rho", 56)):
        mine = tab[{"sf0_perr14": "perr14", "sf0_perr58": "perr58", "sf0_interp": "interp"}.get(name, name)]
        assert np.array_equal(mine.view(np.uint32), orc(name, n)), name
    assert np.array_equal(tab["hoc"][0].view(np.uint32), orc("hoc0", 512))
    for k in (1, 2, 3):
        assert np.array_equal(tab["hoc"][k][:256].view(np.uint32), orc("hoc%d" % k, 256))
    assert np.array_equal(tab["vuv"], orc("vuv", 64, C.c_uint16))
    assert np.array_equal(tab["hpg"], orc("hpg", 192, C.c_uint8))
    assert (tab["hpg"].reshape(48, 4).sum(1) == np.arange(9, 57)).all()      # the blocks of L harmonics add up to L


@pytest.mark.skipif(not os.path.isfile("/root/reference/src/codec/tables.c"), reason="reference sources not on this machine")
def test_codebooks_against_the_reference_file(tab):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_ambe_tables
    ref = gen_ambe_tables.parse()
    names = {"sf0_perr14": "perr14", "sf0_perr58": "perr58", "sf0_interp": "interp"}
    for name, (dims, kind, words) in ref.items():
        w = np.array(words, np.uint32 if kind == "f32" else (np.uint16 if kind == "u16" else np.uint8))
        if name.startswith("hoc"):
            mine = tab["hoc"][int(name[3])][:len(w)].view(np.uint32)
        else:
            mine = tab[names.get(name, name)]
            mine = mine.view(np.uint32) if kind == "f32" else mine
        assert np.array_equal(mine.reshape(-1), w), name
        # and the decimal literal -> float conversion of the generator against the C compiler's: 0.011230f etc.
    assert np.float32(0.011230) == np.frombuffer(np.uint32(gen_ambe_tables.f32_bits_of_literal("0.011230")).tobytes(), np.float32)[0]


def test_state_size_and_calls_without_a_device(api):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert api.codec_state_bytes() % 16 == 0 and api.codec_state_bytes() > 1000
    f = api.load().gmr1_codec_alloc
    f.restype = C.c_void_p
    assert f() is None                          # no device: NULL, as when the reference's calloc fails
    fr = np.zeros((1, 1, 10), np.uint8)
    with pytest.raises(api.Gmr1HipError, match="-19"):
        api.codec_decode_batch(fr)
    rc = api.load().gmr1_codec_decode_frame(None, None, 160, None, 0)
    assert rc == -22


def _orc_powf(x, y, x_is_base):
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros_like(x)
    oracle_lib.lib().orc_ambe_powf_array(C.c_int(x.size), x.ctypes.data_as(C.c_void_p), C.c_float(y), C.c_int(x_is_base),
                                         out.ctypes.data_as(C.c_void_p))
    return out


def test_restated_powf_returns_libms_bits(api):
    """ambe_libm.h (glibc's powf algorithm as the kernel evaluates it) against the libm the reference calls: identical
    bits for powf(2, y) over the log-magnitude range and far beyond it, and for powf(x, 0.25f) over 60 octaves."""
    rng = np.random.default_rng(8)
    n = 5_000_000
    for lo, hi in ((-20.0, 20.0), (-124.0, 124.0)):
        y = rng.uniform(lo, hi, n).astype(np.float32)
        got = api.codec_libm_check(0, y)
        assert not np.isnan(got).any()
        assert np.array_equal(got.view(np.uint32), _orc_powf(y, 2.0, 0).view(np.uint32))
    x = (np.exp2(rng.uniform(-30, 30, n)) * rng.uniform(1, 2, n)).astype(np.float32)
    got = api.codec_libm_check(1, x)
    assert not np.isnan(got).any()
    assert np.array_equal(got.view(np.uint32), _orc_powf(x, 0.25, 1).view(np.uint32))
    # every float in a stretch around 1 (where the log table's entries meet), consecutively
    x = (np.arange(0x3f000000, 0x3f000000 + 4_000_000, dtype=np.uint32)).view(np.float32)
    assert np.array_equal(api.codec_libm_check(1, x).view(np.uint32), _orc_powf(x, 0.25, 1).view(np.uint32))
    # outside what is restated the check says so (the kernel then evaluates in double precision)
    bad = api.codec_libm_check(1, np.array([0.0, -1.0, np.inf, np.nan, 1e-45], np.float32))
    assert np.isnan(bad).all()
    assert np.isnan(api.codec_libm_check(0, np.array([130.0, -130.0, np.nan], np.float32))).all()


def test_restated_cosf_returns_libms_bits(api):
    """The tone frames' cosf (tone.c:104) runs on a phase that grows without bound: all three ranges of glibc's routine
    (no reduction, one-multiply reduction below 120, 96 bits of 4 / pi above), and every float bit pattern class."""
    rng = np.random.default_rng(9)
    n = 4_000_000
    sets = [rng.uniform(-1, 1, n), rng.uniform(-120, 120, n), rng.uniform(0, 2e6, n), rng.uniform(-1e9, 1e9, n),
            np.arange(0x3f400000, 0x3f400000 + n, dtype=np.uint32).view(np.float32),           # around pi / 4
            rng.integers(0, 0x7f800000, n, dtype=np.uint32).view(np.float32)]                   # any finite magnitude
    for x in sets:
        x = np.ascontiguousarray(x, np.float32)
        want = np.zeros_like(x)
        oracle_lib.lib().orc_ambe_cosf_array(C.c_int(x.size), x.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p))
        got = api.codec_libm_check(2, x)
        differ = int((got.view(np.uint32) != want.view(np.uint32)).sum())
        assert differ == 0, differ
    assert np.isnan(api.codec_libm_check(2, np.array([np.inf, -np.inf, np.nan], np.float32))).all()
