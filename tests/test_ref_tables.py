"""The integer tables of the reference, read from its own source files, against the oracle's and the product's.

Container only (skipped where /root/reference is absent, i.e. on the GPU box): tests/ref_parse.py parses
src/l1/conv.c (every trellis table + the polynomials in its comments), src/l1/punct.c (all 51 masks),
src/sdr/nb.c (all 10 burst formats) and the formulas of scramb.c / interleave.c are restated from the cited lines.
This replaces hand-typed spot rows as the pin of every integer table on the path: the oracle's (oracle/orc_*.c)
and the product's (host tables, encoder plans, exported description objects) are compared entry by entry.
"""
import ctypes as C
import os

import numpy as np
import pytest

import ref_parse

pytestmark = pytest.mark.skipif(not ref_parse.available(), reason="/root/reference is not present on this machine")


class OrcCode(C.Structure):
    _fields_ = [("N", C.c_int), ("K", C.c_int), ("len", C.c_int), ("term", C.c_int),
                ("next_output", (C.c_uint8 * 2) * 256), ("next_state", (C.c_uint8 * 2) * 256),
                ("n_punct", C.c_int), ("punct", C.c_int * 1024)]


TERM = {"CONV_TERM_FLUSH": 0, "CONV_TERM_TRUNCATION": 1, "CONV_TERM_TAIL_BITING": 2}


def _orc_code(orc, fn, which):
    f = getattr(orc.lib(), fn)
    f.restype = C.POINTER(OrcCode)
    return f(C.c_int(which)).contents


# ---------------------------------------------------------------------------------------------------------------
# conv.c
# ---------------------------------------------------------------------------------------------------------------
def test_every_conv_table_follows_from_its_comment_polynomials():
    """Self-consistency of the reference (and of the parser): all 10 codes of conv.c."""
    codes = ref_parse.parse_conv()
    assert sorted(codes) == sorted(["gmr1_conv_k5_12", "gmr1_conv_k5_13", "gmr1_conv_k5_14", "gmr1_conv_k5_15",
                                    "gmr1_conv_k6_14", "gmr1_conv_k9_12", "gmr1_conv_k9_13", "gmr1_conv_k9_14",
                                    "gmr1_conv_tch3"])
    for name, c in codes.items():
        # every printed table is the linear feed-forward code of the generators read off its unit states ...
        out, nxt = ref_parse.trellis_from_polys(c["N"], c["K"], c["polys_table"])
        assert len(c["next_output"]) == 1 << (c["K"] - 1) == len(c["next_state"]), name
        assert out == c["next_output"], name
        assert nxt == c["next_state"], name
        # ... and those generators are the ones the comment above it states -- except for k9_14 (used by no chain of
        # the reference), whose table implements g3 without the D^5 term its comment lists (conv.c:437 vs :440-506)
        if name == "gmr1_conv_k9_14":
            assert c["polys_table"][:3] == c["polys_comment"][:3]
            assert c["polys_table"][3] == c["polys_comment"][3] ^ (1 << 5)
        else:
            assert c["polys_table"] == c["polys_comment"], name


def test_oracle_trellis_generator_reproduces_every_table_of_conv_c(orc):
    """orc_conv_make(polynomials) == the table printed in conv.c, every state, every code (incl. the unused ones)."""
    for name, c in ref_parse.parse_conv().items():
        code = OrcCode()
        polys = (C.c_uint * c["N"])(*c["polys_table"])
        orc.lib().orc_conv_make(C.byref(code), c["N"], c["K"], 10, 0, polys)
        ns = 1 << (c["K"] - 1)
        assert [[code.next_output[s][0], code.next_output[s][1]] for s in range(ns)] == c["next_output"], name
        assert [[code.next_state[s][0], code.next_state[s][1]] for s in range(ns)] == c["next_state"], name


# which base code, length, termination and puncturing each chain is specialised with (reference constructors)
CHAINS = {
    # name: (accessor, index, base code, len, term, (pre, main, post, repeat) or None or "rach")
    "bcch/ccch": ("orc_l1_code", 0, "gmr1_conv_k5_12", 208, "CONV_TERM_FLUSH", None),                 # bcch.c:44-50
    "facch3": ("orc_l1_code", 1, "gmr1_conv_k5_14", 92, "CONV_TERM_FLUSH", None),                     # facch3.c:44-50
    "tch3": ("orc_l1_code", 2, "gmr1_conv_tch3", 48, "CONV_TERM_TAIL_BITING",                         # tch3.c:42-49
             (None, "gmr1_punct_k5_12_P12", None, 0)),
    "facch9": ("orc_nt9_code", 0, "gmr1_conv_k5_12", 316, "CONV_TERM_FLUSH", None),                   # facch9.c:42-48
    "tch9_2k4": ("orc_nt9_code", 1, "gmr1_conv_k5_15", 144, "CONV_TERM_FLUSH",                        # tch9.c:59-64
                 ("gmr1_punct_k5_15_P53", "gmr1_punct_k5_15_P23", "gmr1_punct_k5_15_Ps53", 41)),
    "tch9_4k8": ("orc_nt9_code", 2, "gmr1_conv_k5_13", 240, "CONV_TERM_FLUSH",                        # tch9.c:66-71
                 ("gmr1_punct_k5_13_P15", "gmr1_punct_k5_13_P25", "gmr1_punct_k5_13_Ps15", 41)),
    "tch9_9k6": ("orc_nt9_code", 3, "gmr1_conv_k5_12", 480, "CONV_TERM_FLUSH",                        # tch9.c:73-78
                 ("gmr1_punct_k5_12_P25", "gmr1_punct_k5_12_P23", "gmr1_punct_k5_12_Ps25", 158)),
    "xch_dc12": ("orc_xch_code", 0, "gmr1_conv_k9_13", 208, "CONV_TERM_TAIL_BITING",                  # xch_dc12.c:45-53
                 (None, "gmr1_punct_k9_13_P1213", None, 0)),
    "rach": ("orc_xch_code", 1, "gmr1_conv_k5_14", 159, "CONV_TERM_FLUSH", "rach"),                   # rach.c:44-66
}


def _coded_len(N, K, length, term):
    return (length + (K - 1 if term == "CONV_TERM_FLUSH" else 0)) * N      # osmo_conv_get_output_length(code, 0)


def _ref_chain(name):
    _, _, base, length, term, p = CHAINS[name]
    conv = ref_parse.parse_conv()[base]
    masks = ref_parse.parse_punct()
    N, K = conv["N"], conv["K"]
    cl = _coded_len(N, K, length, term)
    if p is None:
        punct = []
    elif p == "rach":
        punct = sorted([4 * i + 2 for i in range(135)] + [4 * i + 3 for i in range(135)])     # rach.c:57-62
    else:
        pre, main, post, rep = p
        punct = ref_parse.puncturer_generate(N, cl, masks.get(pre), masks[main], masks.get(post), rep)
    return conv, length, term, cl, punct


@pytest.mark.parametrize("name", sorted(CHAINS))
def test_oracle_chain_codes_equal_the_reference_constructors(orc, name):
    """Trellis, length, termination and the whole punctured-position list of every chain's specialised code: the
    oracle's (built from polynomials and its own copy of the masks) against conv.c + punct.c run through the
    reference's generator."""
    acc, idx = CHAINS[name][:2]
    code = _orc_code(orc, acc, idx)
    conv, length, term, cl, punct = _ref_chain(name)
    ns = 1 << (conv["K"] - 1)
    assert (code.N, code.K, code.len, code.term) == (conv["N"], conv["K"], length, TERM[term])
    assert [[code.next_output[s][0], code.next_output[s][1]] for s in range(ns)] == conv["next_output"]
    assert [[code.next_state[s][0], code.next_state[s][1]] for s in range(ns)] == conv["next_state"]
    assert code.n_punct == len(punct)
    assert list(code.punct[:code.n_punct]) == punct
    assert punct == sorted(punct) and (not punct or punct[-1] < cl)
    # the burst carries exactly the unpunctured bits (what each decoder is handed)
    carried = {"bcch/ccch": 424, "facch3": 384, "tch3": 72, "facch9": 640, "tch9_2k4": 648, "tch9_4k8": 648,
               "tch9_9k6": 648, "xch_dc12": 432, "rach": 382}[name]
    assert cl - len(punct) == carried


# ---------------------------------------------------------------------------------------------------------------
# punct.c
# ---------------------------------------------------------------------------------------------------------------
def test_all_masks_parse_and_are_self_consistent():
    masks = ref_parse.parse_punct()
    assert len(masks) == 51          # punct.h:54-106 declares 51 (SURVEY.md counts 53)
    # the names punct.h declares (reference include/osmocom/gmr1/l1/punct.h:54-106) are exactly these
    import re
    with open(os.path.join(ref_parse.REF, "include/osmocom/gmr1/l1/punct.h")) as f:
        declared = re.findall(r"extern const struct gmr1_puncturer (\w+);", f.read())
    assert declared == list(masks)
    for name, m in masks.items():
        assert len(m["mask"]) == m["L"] * m["N"], name
        if name == "gmr1_punct_k5_12_E":
            # quirk (punct.c:313-324): scheme E *repeats* a bit -- its mask holds a 2 and no 0, r = 1 counts the
            # repetition; gmr1_puncturer_generate only looks for zeros, so E punctures nothing
            assert m["mask"] == [1, 2, 1, 1, 1, 1, 1, 1] and m["r"] == 1
            continue
        assert set(m["mask"]) <= {0, 1}, name
        assert m["mask"].count(0) == m["r"], name                      # r = number of punctured bits of the mask
        n_in_name = int(name.split("_")[3][1])                          # k5_12 -> rate 1/2
        assert m["N"] == n_in_name, name


def test_known_answer_fixture_agrees_with_the_parsed_masks():
    """tests/golden/known_answers.json (derived by hand in round 1) against the mechanical derivation."""
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "known_answers.json")) as f:
        g = json.load(f)
    assert _ref_chain("tch9_9k6")[4] == g["tch9_9k6_punctured"]
    assert _ref_chain("xch_dc12")[4][:13] == g["xch_dc12_punctured_first13"]
    conv = ref_parse.parse_conv()
    name_of = {"k5_12": "gmr1_conv_k5_12", "k5_13": "gmr1_conv_k5_13", "k5_14": "gmr1_conv_k5_14",
               "k5_15": "gmr1_conv_k5_15", "k9_13": "gmr1_conv_k9_13", "tch3_k7": "gmr1_conv_tch3"}
    for k, spec in g["conv_spot"].items():
        c = conv[name_of[k]]
        assert spec["polys"] == c["polys_comment"], k
        for s, row in spec["next_output_rows"].items():
            assert c["next_output"][int(s)] == row, (k, s)


# ---------------------------------------------------------------------------------------------------------------
# nb.c / pi4cxpsk.c
# ---------------------------------------------------------------------------------------------------------------
def test_all_10_burst_formats_equal_nb_c(pkg, orc):
    """Sync sequences (position, symbols), data chunks, guard, length, ebits and modulation of every burst format:
    nb.c against the oracle's table and against the product's host table."""
    ref = ref_parse.parse_nb()
    assert sorted(ref) == sorted(pkg.api.BURST_IDS)
    mods = ref_parse.parse_modulations()
    rot = {"gmr1_pi2cbpsk": np.pi / 2, "gmr1_pi4cbpsk": np.pi / 4, "gmr1_pi4cqpsk": np.pi / 4}
    assert {m: mods[m]["rotation_expr"] for m in mods} == {"gmr1_pi2cbpsk": "M_PIf/2", "gmr1_pi4cbpsk": "M_PIf/4",
                                                            "gmr1_pi4cqpsk": "M_PIf/4"}
    for name in pkg.api.BURST_IDS:
        r = ref[name]
        for who, f in (("oracle", orc.burst_format(name)), ("product", pkg.api.burst_format(name))):
            assert f.length == r["len"] and f.ebits == r["ebits"], (who, name)
            assert f.nbits == mods[r["mod"]]["nbits"], (who, name)
            assert abs(f.rotation - rot[r["mod"]]) < 1e-6, (who, name)
            assert [(p, list(s)) for p, s in (c for c in sum(f.sync, []))] == \
                   [(p, s) for p, s in sum(r["sync"], [])], (who, name)
            assert [len(seq) for seq in f.sync] == [len(seq) for seq in r["sync"]], (who, name)
            assert list(f.data) == r["data"], (who, name)
        b = pkg.api.burst_info(name)
        assert (b.guard_pre, b.guard_post) == (r["guard_pre"], r["guard_post"]), name
        ob = orc.burst(name).contents
        assert (ob.guard_pre, ob.guard_post) == (r["guard_pre"], r["guard_post"]), name


# ---------------------------------------------------------------------------------------------------------------
# scramb.c / interleave.c / crc.c: formulas and parameters as the files state them
# ---------------------------------------------------------------------------------------------------------------
def test_scrambler_and_interleavers_follow_the_reference_formulas(pkg, orc):
    import re
    src = open(os.path.join(ref_parse.REF, "src/l1/scramb.c")).read()
    init = int(re.search(r"#define GMR1_SCRAMBLE_REG_INIT\s+(0x[0-9a-fA-F]+)", src).group(1), 16)     # scramb.c:39
    assert re.search(r"b = \(\(reg_val >> 14\) \^ reg_val\) & 1;\s*\*reg = \(reg_val << 1\) \| b;", src)  # :47-48
    r, bits = init, []
    for _ in range(662):                                  # longest burst (NT9)
        b = ((r >> 14) ^ r) & 1
        r = ((r << 1) | b) & 0xFFFF
        bits.append(b)
    ones = np.ones(662, np.int8)
    got = orc.scramble_sbit(ones)
    assert np.array_equal(got < 0, np.array(bits, bool))
    # intra-burst (de)interleaver, interleave.c:46-87: kep = N * ((5 * kc) & 7) + (kc >> 3)
    isrc = open(os.path.join(ref_parse.REF, "src/l1/interleave.c")).read()
    assert re.search(r"\(\s*\(\s*5\s*\*\s*\w+\s*\)\s*&\s*7\s*\)", isrc) or "5 * kc" in isrc or "5*kc" in isrc
    for N in (12, 14, 33, 53, 80, 81):
        x = np.random.default_rng(N).integers(0, 256, 8 * N).astype(np.uint8)
        kc = np.arange(8 * N)
        assert np.array_equal(orc.deinterleave_intra(x, N), x[N * ((5 * kc) & 7) + (kc >> 3)])


def test_crc_parameters_equal_crc_c(orc):
    import re
    src = ref_parse._strip_comments(open(os.path.join(ref_parse.REF, "src/l1/crc.c")).read())
    got = {}
    for m in re.finditer(r"const struct osmo_crc\d+gen_code (\w+)\s*=\s*\{(.*?)\};", src, flags=re.S):
        f = {k: int(v, 0) for k, v in re.findall(r"\.(\w+)\s*=\s*(0x[0-9a-fA-F]+|\d+)", m.group(2))}
        got[m.group(1)] = (f["bits"], f["poly"], f["init"], f["remainder"])
    assert got == {"gmr1_crc8": (8, 0x9b, 0, 0), "gmr1_crc12": (12, 0x80f, 0, 0), "gmr1_crc16": (16, 0x1021, 0, 0)}

    class Crc(C.Structure):
        _fields_ = [("bits", C.c_int), ("poly", C.c_uint32), ("init", C.c_uint32), ("remainder", C.c_uint32)]
    f = orc.lib().orc_crc_compute_bits
    f.restype = C.c_uint32
    rng = np.random.default_rng(5)
    for name, (bits, poly, init, rem) in got.items():
        msg = rng.integers(0, 2, 200).astype(np.uint8)
        reg = init                                       # bitwise long division, MSB first
        for b in msg:
            top = ((reg >> (bits - 1)) & 1) ^ int(b)
            reg = (reg << 1) & ((1 << bits) - 1)
            if top:
                reg ^= poly
        c = Crc(bits, poly, init, rem)
        assert f(C.byref(c), msg.ctypes.data_as(C.c_void_p), C.c_int(200)) == (reg ^ rem), name


# ---------------------------------------------------------------------------------------------------------------
# the description objects the library exports (l1/conv.h, l1/punct.h, l1/crc.h) against the reference files
# ---------------------------------------------------------------------------------------------------------------
class ConvCode(C.Structure):      # struct osmo_conv_code, include/osmocom/gmr1/compat.h
    _fields_ = [("N", C.c_int), ("K", C.c_int), ("len", C.c_int), ("term", C.c_int),
                ("next_output", C.POINTER(C.c_uint8 * 2)), ("next_state", C.POINTER(C.c_uint8 * 2)),
                ("next_term_output", C.c_void_p), ("next_term_state", C.c_void_p), ("puncture", C.POINTER(C.c_int))]


class PunctHead(C.Structure):     # struct gmr1_puncturer up to its flexible mask[]
    _fields_ = [("r", C.c_int), ("L", C.c_int), ("N", C.c_int)]


def _exported_mask(lib, name):
    h = PunctHead.in_dll(lib, name)
    m = (C.c_uint8 * (h.L * h.N)).from_address(C.addressof(h) + C.sizeof(PunctHead))
    return h, list(m)


def test_exported_conv_objects_equal_conv_c(pkg):
    lib = pkg.api.load()
    for name, c in ref_parse.parse_conv().items():
        e = ConvCode.in_dll(lib, name)
        ns = 1 << (c["K"] - 1)
        assert (e.N, e.K, e.len, e.term) == (c["N"], c["K"], 0, TERM[c["term"]]), name
        assert [[e.next_output[s][0], e.next_output[s][1]] for s in range(ns)] == c["next_output"], name
        assert [[e.next_state[s][0], e.next_state[s][1]] for s in range(ns)] == c["next_state"], name
        assert not e.next_term_output and not e.next_term_state and not e.puncture, name


def test_exported_puncturers_and_crcs_equal_the_reference(pkg):
    lib = pkg.api.load()
    for name, m in ref_parse.parse_punct().items():
        h, mask = _exported_mask(lib, name)
        assert (h.r, h.L, h.N) == (m["r"], m["L"], m["N"]), name
        assert mask == m["mask"], name

    class Crc8(C.Structure):
        _fields_ = [("bits", C.c_int), ("poly", C.c_uint8), ("init", C.c_uint8), ("remainder", C.c_uint8)]

    class Crc16(C.Structure):
        _fields_ = [("bits", C.c_int), ("poly", C.c_uint16), ("init", C.c_uint16), ("remainder", C.c_uint16)]
    for name, T, exp in (("gmr1_crc8", Crc8, (8, 0x9b, 0, 0)), ("gmr1_crc12", Crc16, (12, 0x80f, 0, 0)),
                         ("gmr1_crc16", Crc16, (16, 0x1021, 0, 0))):
        c = T.in_dll(lib, name)
        assert (c.bits, c.poly, c.init, c.remainder) == exp, name


@pytest.mark.parametrize("name", [n for n in sorted(CHAINS) if CHAINS[n][5] not in (None, "rach")])
def test_exported_puncturer_generate_reproduces_the_reference_lists(pkg, name):
    """gmr1_puncturer_generate on the exported schemes, called the way the reference's constructors call it
    (tch3.c:46-48, tch9.c:59-78, xch_dc12.c:49-52), against the reference's generator run over the parsed masks."""
    lib = pkg.api.load()
    _, _, base, length, term, (pre, main, post, rep) = CHAINS[name]
    code = ConvCode()
    C.memmove(C.byref(code), C.addressof(ConvCode.in_dll(lib, base)), C.sizeof(ConvCode))
    code.len, code.term = length, TERM[term]
    ptr = lambda n: C.c_void_p(C.addressof(PunctHead.in_dll(lib, n))) if n else None
    lib.gmr1_puncturer_generate.restype = C.c_int
    assert lib.gmr1_puncturer_generate(C.byref(code), ptr(pre), ptr(main), ptr(post), C.c_int(rep)) == 0
    got = []
    while code.puncture[len(got)] >= 0:
        got.append(code.puncture[len(got)])
    assert got == _ref_chain(name)[4]
    C.CDLL(None).free(code.puncture)
    # a scheme for another code rate is refused
    wrong = "gmr1_punct_k5_13_P16" if code.N != 3 else "gmr1_punct_k5_12_P23"
    assert lib.gmr1_puncturer_generate(C.byref(code), None, ptr(wrong), None, 0) == -22
