"""The encoder position maps (struct EncPlan, built on the host by csrc/capi_tx.cpp) evaluated bit by bit in
numpy -- the same three steps k_encode runs (tx_kernels.hip) -- against the CPU oracle's encoders.  No GPU needed:
this pins the host logic of the transmit direction; tests/test_gpu_tx.py runs the kernel itself."""
import numpy as np
import pytest

import oracle_lib
from __graft_entry__ import load_package

pkg = load_package()


def eval_plan(P, in0, in1=None, aux0=None, aux1=None, ciph=None, seq_len=1):
    """numpy model of k_encode: (n, n_in0) payload bytes ... -> (n, n_out) ubits"""
    n = in0.shape[0]
    pay = in0 if in1 is None else np.concatenate([in0, in1.reshape(n, -1)], axis=1)
    bits = np.unpackbits(pay, axis=1, bitorder="little").astype(np.uint32)       # payload bit b = byte * 8 + bit from the LSB
    nb = bits.shape[1]
    crc_a = np.bitwise_xor.reduce(bits * P["crc_tab"][:nb].astype(np.uint32), axis=1)
    crc_b = np.bitwise_xor.reduce(bits * P["crc_tab2"][:nb].astype(np.uint32), axis=1)
    ext = np.zeros((n, P["n_ext"] + 64), np.uint32)
    for i, src in enumerate(P["ext_src"]):
        src = int(src)
        if src == 0xffff:
            continue
        if src < 0x4000:
            ext[:, i] = bits[:, src]
        elif src < 0x8000:
            ext[:, i] = (crc_a >> (src & 15)) & 1
        else:
            ext[:, i] = (crc_b >> (src & 15)) & 1
    aux = None
    if P["n_aux0"]:
        aux = aux0.reshape(n, -1) if not P["n_aux1"] else np.concatenate([aux0.reshape(n, -1), aux1.reshape(n, -1)], axis=1)
    out = np.zeros((n, P["n_out"]), np.uint8)
    pos = np.arange(n) % seq_len if P["depth"] > 1 else np.zeros(n, int)
    for e, ds in enumerate(P["out"]):
        ds = int(ds)
        t, poly, kind, scr, ci, d = ds & 1023, (ds >> 10) & 7, (ds >> 13) & 3, (ds >> 15) & 1, (ds >> 16) & 1023, (ds >> 26) & 3
        if kind == 0:
            mask = int(P["poly"][poly])
            v = np.zeros(n, np.uint32)
            for q in range(9):
                if (mask >> q) & 1:
                    v ^= ext[:, t + q]
            if d:
                v = np.where(pos >= d, np.roll(v, d), 0)            # burst n - d of the same run, empty before the run
            bit = v
        elif kind == 2:
            bit = aux[:, t].astype(np.uint32) & 1
        else:
            bit = np.zeros(n, np.uint32)
        bit = bit ^ scr
        if ci and ciph is not None:
            bit = bit ^ (ciph[:, ci - 1] & 1)
        out[:, e] = bit
    return out


def test_plan_sizes():
    want = {"bcch": (24, 424), "ccch": (24, 432), "facch3": (10, 416), "tch3_m0": (20, 212), "tch3_m1": (20, 212),
            "facch9": (38, 662), "tch9_2k4": (18, 662), "tch9_4k8": (30, 662), "tch9_9k6": (60, 662), "rach": (18, 494),
            "xch_dc12": (24, 432)}
    for name, (n_in, n_out) in want.items():
        P = pkg.api.encoder_plan(name)
        assert (P["n_in0"], P["n_out"]) == (n_in, n_out), name
        assert P["n_ext"] <= 512 and P["depth"] == (3 if name.startswith("tch9") else 1)


@pytest.mark.parametrize("chain", ["bcch", "ccch", "xch_dc12"])
def test_plan_l2_24(chain):
    rng = np.random.default_rng(11)
    l2 = rng.integers(0, 256, (40, 24), dtype=np.uint8)
    l2[0] = 0
    l2[1] = 255
    got = eval_plan(pkg.api.encoder_plan(chain), l2)
    if chain == "xch_dc12":
        ref = np.stack([oracle_lib.xch_dc12_encode(x) for x in l2])
    else:
        ref = getattr(oracle_lib, chain + "_encode")(l2)
    assert np.array_equal(got, ref)


def test_plan_facch3():
    rng = np.random.default_rng(12)
    l2 = rng.integers(0, 256, (30, 10), dtype=np.uint8)
    bits_s = rng.integers(0, 2, (30, 32), dtype=np.uint8)
    ciph = rng.integers(0, 2, (30, 384), dtype=np.uint8)
    P = pkg.api.encoder_plan("facch3")
    assert np.array_equal(eval_plan(P, l2, aux0=bits_s).reshape(30, 4, 104), oracle_lib.facch3_encode(l2, bits_s))
    assert np.array_equal(eval_plan(P, l2, aux0=bits_s, ciph=ciph).reshape(30, 4, 104),
                          oracle_lib.facch3_encode(l2, bits_s, ciph))


@pytest.mark.parametrize("m", [0, 1])
def test_plan_tch3(m):
    rng = np.random.default_rng(13 + m)
    fr = rng.integers(0, 256, (30, 2, 10), dtype=np.uint8)
    bits_s = rng.integers(0, 2, (30, 4), dtype=np.uint8)
    ciph = rng.integers(0, 2, (30, 208), dtype=np.uint8)
    P = pkg.api.encoder_plan("tch3_m%d" % m)
    assert np.array_equal(eval_plan(P, fr.reshape(30, 20), aux0=bits_s), oracle_lib.tch3_encode(fr[:, 0], fr[:, 1], bits_s, m))
    assert np.array_equal(eval_plan(P, fr.reshape(30, 20), aux0=bits_s, ciph=ciph),
                          oracle_lib.tch3_encode(fr[:, 0], fr[:, 1], bits_s, m, ciph))


def test_plan_facch9():
    rng = np.random.default_rng(15)
    l2 = rng.integers(0, 256, (12, 38), dtype=np.uint8)
    sa = rng.integers(0, 2, (12, 10), dtype=np.uint8)
    stt = rng.integers(0, 2, (12, 4), dtype=np.uint8)
    ciph = rng.integers(0, 2, (12, 658), dtype=np.uint8)
    P = pkg.api.encoder_plan("facch9")
    ref = np.stack([oracle_lib.facch9_encode(l2[i], sa[i], stt[i], ciph[i]) for i in range(12)])
    assert np.array_equal(eval_plan(P, l2, aux0=sa, aux1=stt, ciph=ciph), ref)
    ref = np.stack([oracle_lib.facch9_encode(l2[i], sa[i], stt[i]) for i in range(12)])
    assert np.array_equal(eval_plan(P, l2, aux0=sa, aux1=stt), ref)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_plan_tch9(mode):
    rng = np.random.default_rng(16 + mode)
    nb = (18, 30, 60)[mode]
    seq = 7
    l2 = rng.integers(0, 256, (2 * seq, nb), dtype=np.uint8)
    sa = rng.integers(0, 2, (2 * seq, 10), dtype=np.uint8)
    stt = rng.integers(0, 2, (2 * seq, 4), dtype=np.uint8)
    ciph = rng.integers(0, 2, (2 * seq, 658), dtype=np.uint8)
    P = pkg.api.encoder_plan("tch9_" + ("2k4", "4k8", "9k6")[mode])
    ref = np.concatenate([oracle_lib.tch9_encode_seq(l2[r * seq:(r + 1) * seq], mode, sa[r * seq:(r + 1) * seq],
                                                     stt[r * seq:(r + 1) * seq], ciph[r * seq:(r + 1) * seq]) for r in range(2)])
    assert np.array_equal(eval_plan(P, l2, aux0=sa, aux1=stt, ciph=ciph, seq_len=seq), ref)


def test_plan_rach():
    rng = np.random.default_rng(19)
    rach = rng.integers(0, 256, (40, 18), dtype=np.uint8)
    sb = rng.integers(0, 256, 40, dtype=np.uint8)
    sb[0] = 0
    P = pkg.api.encoder_plan("rach")
    assert (P["n_in0"], P["n_in1"]) == (18, 1)
    ref = np.stack([oracle_lib.rach_encode(rach[i], int(sb[i])) for i in range(40)])
    assert np.array_equal(eval_plan(P, rach, in1=sb), ref)


def test_plan_call_rejects_bad_arguments():
    import ctypes as C
    L = pkg.api.load()
    size = L.gmr1_hip_encoder_plan(C.c_int(0), None, C.c_int(0))
    assert size > 0
    assert L.gmr1_hip_encoder_plan(C.c_int(-1), None, C.c_int(0)) == -22
    assert L.gmr1_hip_encoder_plan(C.c_int(len(pkg.api.ENC_CHAINS)), None, C.c_int(0)) == -22
    buf = np.zeros(size, np.uint8)
    assert L.gmr1_hip_encoder_plan(C.c_int(0), buf.ctypes.data_as(C.c_void_p), C.c_int(size - 1)) == -22
    assert L.gmr1_hip_encoder_plan(C.c_int(0), buf.ctypes.data_as(C.c_void_p), C.c_int(size)) == size


def test_plans_are_linear_maps():
    """Every chain is GF(2)-linear in (payload, status bits, keystream) up to its scrambler constant: the map of the xor of
    two inputs is the xor of the maps, xor the map of the all-zero input.  (What lets one kernel evaluate all of them.)"""
    rng = np.random.default_rng(5)
    for chain in pkg.api.ENC_CHAINS:
        P = pkg.api.encoder_plan(chain)
        n = 6
        def draw():
            in0 = rng.integers(0, 256, (n, P["n_in0"]), dtype=np.uint8)
            in1 = rng.integers(0, 256, (n, P["n_in1"]), dtype=np.uint8) if P["n_in1"] else None
            a0 = rng.integers(0, 2, (n, P["n_aux0"]), dtype=np.uint8) if P["n_aux0"] else None
            a1 = rng.integers(0, 2, (n, P["n_aux1"]), dtype=np.uint8) if P["n_aux1"] else None
            c = rng.integers(0, 2, (n, P["n_ciph"]), dtype=np.uint8) if P["n_ciph"] else None
            return [in0, in1, a0, a1, c]
        x, y = draw(), draw()
        z = [None if a is None else a ^ b for a, b in zip(x, y)]
        zero = [None if a is None else np.zeros_like(a) for a in x]
        f = lambda v: eval_plan(P, v[0], v[1], v[2], v[3], v[4], seq_len=3)
        assert np.array_equal(f(z), f(x) ^ f(y) ^ f(zero)), chain
