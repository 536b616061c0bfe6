"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol the
headers declare, carries the same burst tables as the oracle, and refuses to compute
without a GPU (no CPU fallback)."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# declared in include/ for the application's sake but defined BY the application, as in the reference
# (src/Makefile.am:8 compiles src/gsmtap.c into gmr1_rx: it allocates a libosmocore msgb, which this library does not link)
PROGRAM_SUPPLIED = {"gmr1_gsmtap_makemsg"}


def _declared_symbols():
    funcs, data = set(), set()
    for h in glob.glob(os.path.join(ROOT, "include", "**", "*.h"), recursive=True):
        txt = open(h).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        txt = re.sub(r"//[^\n]*", "", txt)
        funcs |= set(re.findall(r"\b(gmr1_[a-z0-9_]+)\s*\(", txt))
        data |= set(re.findall(r"extern\s+(?:const\s+)?struct\s+\w+\s+(gmr1_\w+)\s*;", txt))
    return funcs, data


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.api.load()
    funcs, data = _declared_symbols()
    assert funcs and data
    assert PROGRAM_SUPPLIED <= funcs
    funcs -= PROGRAM_SUPPLIED
    for name in sorted(funcs | data):
        assert hasattr(lib, name), f"{name} is declared in include/ but not exported"
    # and the python mirror lists the same set
    assert set(pkg.api.EXPORTED_FUNCTIONS) == funcs
    assert set(pkg.api.EXPORTED_DATA) == data


def test_version_and_error_strings(pkg):
    lib = pkg.api.load()
    assert b"gfx950" in lib.gmr1_hip_version()
    assert isinstance(lib.gmr1_hip_last_error(), bytes)


def test_burst_tables_equal_oracle_tables(pkg, orc):
    """Two independent transcriptions of ETSI TS 101 376-5-2 7.4 (reference src/sdr/nb.c)."""
    for name in pkg.api.BURST_IDS:
        assert pkg.api.burst_format(name) == orc.burst_format(name), name
    with pytest.raises(pkg.api.Gmr1HipError):
        pkg.api.burst_info(99)


def test_burst_table_invariants(pkg):
    for name in pkg.api.BURST_IDS:
        f = pkg.api.burst_format(name)
        used = np.zeros(f.length, int)
        for p, l in f.data:
            used[p:p + l] += 1
        assert sum(l for _, l in f.data) * f.nbits == f.ebits
        for seq in f.sync:
            u = used.copy()
            for p, syms in seq:
                u[p:p + len(syms)] += 1
                assert max(syms) < (1 << f.nbits) * (2 if f.nbits == 1 else 1)
            assert u.max() == 1, name                     # sync and data never overlap
            assert u[:2].sum() == 0 and u[-3:].sum() == 0  # guard symbols stay empty


def test_exported_struct_layout_matches_reference_abi(pkg):
    """gmr1_bcch_burst etc. are pointer-linked structs laid out like the reference's
    (include/osmocom/gmr1/sdr/pi4cxpsk.h:43-98); walk one through raw memory."""
    lib = pkg.api.load()

    class Mod(C.Structure):
        _fields_ = [("rotation", C.c_float), ("nbits", C.c_int), ("syms", C.c_void_p), ("bits", C.c_void_p)]

    class Sync(C.Structure):
        _fields_ = [("pos", C.c_int), ("len", C.c_int), ("syms", C.c_uint8 * 32), ("_ref", C.c_void_p)]

    class Data(C.Structure):
        _fields_ = [("pos", C.c_int), ("len", C.c_int)]

    class Burst(C.Structure):
        _fields_ = [("mod", C.POINTER(Mod)), ("guard_pre", C.c_int), ("guard_post", C.c_int),
                    ("len", C.c_int), ("ebits", C.c_int), ("sync", C.POINTER(Sync) * 4),
                    ("data", C.POINTER(Data))]

    b = Burst.in_dll(lib, "gmr1_dc6_burst")
    assert (b.guard_pre, b.guard_post, b.len, b.ebits) == (2, 3, 234, 432)
    assert abs(b.mod.contents.rotation - np.pi / 4) < 1e-6 and b.mod.contents.nbits == 2
    assert (b.sync[0][0].pos, b.sync[0][0].len, list(b.sync[0][0].syms[:7])) == (28, 7, [0, 0, 0, 2, 2, 0, 2])
    assert b.sync[0][3].pos == -1 and not b.sync[1]
    assert (b.data[1].pos, b.data[1].len) == (35, 84) and b.data[4].pos == -1
    sd = Burst.in_dll(lib, "gmr1_sdcch_burst")
    assert all(bool(sd.sync[i]) for i in range(4))       # 4 sequences, no NULL terminator


# every gmr1_* function / data object the reference's application references (src/gmr1_rx.c), except
# gmr1_gsmtap_makemsg, which is compiled into the program itself from src/gsmtap.c (src/Makefile.am:8)
GMR1_RX_LINKS = [
    "gmr1_a5", "gmr1_bcch_burst", "gmr1_bcch_decode", "gmr1_ccch_decode", "gmr1_dc6_burst", "gmr1_dkab_demod",
    "gmr1_facch3_decode", "gmr1_facch9_decode", "gmr1_fcch_burst", "gmr1_fcch_fine", "gmr1_fcch_rough",
    "gmr1_fcch_rough_multi", "gmr1_fcch_snr", "gmr1_interleaver_init", "gmr1_nt3_facch_burst",
    "gmr1_nt3_speech_burst", "gmr1_nt9_burst", "gmr1_pi4cxpsk_demod", "gmr1_pi4cxpsk_detect", "gmr1_tch3_decode",
    "gmr1_tch9_decode",
]


def test_library_covers_what_gmr1_rx_links_against(pkg):
    lib = pkg.api.load()
    for name in GMR1_RX_LINKS:
        assert hasattr(lib, name), f"gmr1_rx.c needs {name}"
    ref = "/root/reference/src/gmr1_rx.c"
    if os.path.exists(ref):          # only in the build container: the list above is what the file references
        with open(ref) as f:
            used = set(re.findall(r"\b(gmr1_[a-z0-9_]+)\b", f.read()))
        used -= {"gmr1_gsmtap_makemsg", "gmr1_interleaver", "gmr1_pi4cxpsk_burst"}     # own code / type names
        assert used == set(GMR1_RX_LINKS)


def test_no_cpu_fallback(pkg):
    """Without a usable GPU every compute entry point fails loudly (-ENODEV)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    eb = np.zeros((4, 424), np.int8)
    with pytest.raises(pkg.api.Gmr1HipError, match="-19"):
        pkg.api.bcch_decode_batch(eb)
    with pytest.raises(pkg.api.Gmr1HipError, match="-19"):
        pkg.api.rx_bcch_ccch_batch(np.zeros(2048, np.complex64), [0], [0])
    l2, rv, _ = pkg.api.bcch_decode(eb[0])
    assert rv == -19
    with pytest.raises(pkg.api.Gmr1HipError, match="-19"):
        pkg.api.Shard(bytes(128), 0, 1)
    with pytest.raises(pkg.api.Gmr1HipError, match="-19"):
        pkg.api.xch_dc12_decode_batch(np.zeros((2, 432), np.int8))
    with pytest.raises(pkg.api.Gmr1HipError, match="-19"):
        pkg.api.tch3_rx_batch(np.zeros(1024, np.complex64), [0, 480], 474)
    with pytest.raises(pkg.api.Gmr1HipError, match="-19"):
        pkg.api.rach_decode_batch(np.zeros((2, 494), np.int8), 0)
    with pytest.raises(pkg.api.Gmr1HipError, match="-19"):
        pkg.api.xch_dc12_decode(np.zeros(432, np.int8))
    # transmit direction: encoders and modulator
    with pytest.raises(pkg.api.Gmr1HipError, match="-19"):
        pkg.api.bcch_encode_batch(np.zeros((2, 24), np.uint8))
    with pytest.raises(pkg.api.Gmr1HipError, match="-19"):
        pkg.api.tch9_encode_batch(np.zeros((3, 18), np.uint8), 0, 3, np.zeros((3, 10), np.uint8), np.zeros((3, 4), np.uint8))
    with pytest.raises(pkg.api.Gmr1HipError, match="-19"):
        pkg.api.mod_batch("bcch", np.zeros((1, 424), np.uint8))
    with pytest.raises(pkg.api.Gmr1HipError):
        pkg.api.encode_single("bcch", np.zeros(24, np.uint8))          # void call: bits_e untouched, error recorded
    rc, _ = pkg.api.pi4cxpsk_mod("bcch", np.zeros(424, np.uint8))
    assert rc == -19


def test_product_does_not_reference_oracle():
    """Nothing under the package or include/ may import, include or link the oracle."""
    bad = []
    for d in ("osmo-gmr_amd", "include"):
        for path in glob.glob(os.path.join(ROOT, d, "**", "*"), recursive=True):
            if os.path.isfile(path) and path.endswith((".py", ".h", ".hip", ".cpp", ".c", ".inc")):
                txt = open(path, errors="replace").read()
                if re.search(r"oracle_lib|liborc|import\s+orc_chan|from\s+orc_chan|orc_[a-z0-9_]+\(|#include\s+\"orc_", txt):
                    bad.append(path)
    assert not bad, bad


def test_gsmtap_packet_layout(pkg):
    """gmr1_gsmtap_makemsg (reference src/gsmtap.c:43-71): version 2, 4-word header, type GMR1_UM,
    frame number big-endian, sub_type = channel type, L2 appended.  Host-only, no GPU needed."""
    api = pkg.api
    rec = np.zeros(1, api.RX_RECORD)
    rec["arfcn"], rec["type"], rec["fn"], rec["tn"], rec["len"] = 1007, 2, 0x00A1B2C3, 17, 24
    rec["l2"][0] = np.arange(24, dtype=np.uint8) + 100
    pkt = api.gsmtap_pack(rec[0])
    assert len(pkt) == 40
    assert pkt[:4] == bytes([2, 4, 0x0A, 17]) and pkt[4:8] == bytes(4)
    assert pkt[8:12] == bytes([0x00, 0xA1, 0xB2, 0xC3]) and pkt[12] == 2 and pkt[13:16] == bytes(3)
    assert pkt[16:] == bytes(range(100, 124))
    pkt2 = api.gsmtap_pack(rec[0], with_arfcn=True)
    assert pkt2[4:6] == bytes([1007 >> 8, 1007 & 0xFF]) and pkt2[6:] == pkt[6:]


def test_gsmtap_big_packet_layout(pkg):
    api = pkg.api
    rec = np.zeros(1, api.RX_BIG_RECORD)
    rec["type"], rec["fn"], rec["tn"], rec["len"] = 0x18, 123456, 5, 60
    rec["l2"][0][:60] = np.arange(60, dtype=np.uint8)
    pkt = api.gsmtap_pack_big(rec[0])
    assert len(pkt) == 76 and pkt[:4] == bytes([2, 4, 0x0A, 5]) and pkt[12] == 0x18
    assert pkt[8:12] == (123456).to_bytes(4, "big") and pkt[16:] == bytes(range(60))


def test_headers_are_plain_c_and_link(pkg, tmp_path):
    """tests/c/abi_smoke.c: a C99 program over every public header, linked against the library, run without a GPU."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = os.path.join(ROOT, "tests", "c", "abi_smoke.c")
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.dirname(pkg.api.lib_path())
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), src,
                           "-o", exe, "-L" + libdir, "-lgmr1_hip", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip()


def test_exported_code_descriptions_agree_with_the_oracle(pkg, orc):
    """l1/conv.h + l1/punct.h objects and gmr1_puncturer_generate (host code) without the reference tree: the TCH9 9k6
    code specialised as tch9.c:73-78 does has the oracle's trellis and the 320 punctured positions of the fixture."""
    import json
    lib = pkg.api.load()

    class ConvCode(C.Structure):
        _fields_ = [("N", C.c_int), ("K", C.c_int), ("len", C.c_int), ("term", C.c_int),
                    ("next_output", C.POINTER(C.c_uint8 * 2)), ("next_state", C.POINTER(C.c_uint8 * 2)),
                    ("next_term_output", C.c_void_p), ("next_term_state", C.c_void_p), ("puncture", C.POINTER(C.c_int))]
    code = ConvCode()
    C.memmove(C.byref(code), C.addressof(ConvCode.in_dll(lib, "gmr1_conv_k5_12")), C.sizeof(ConvCode))
    code.len = 480
    addr = lambda n: C.c_void_p(C.addressof(C.c_int.in_dll(lib, n)))
    lib.gmr1_puncturer_generate.restype = C.c_int
    rc = lib.gmr1_puncturer_generate(C.byref(code), addr("gmr1_punct_k5_12_P25"), addr("gmr1_punct_k5_12_P23"),
                                     addr("gmr1_punct_k5_12_Ps25"), C.c_int(158))
    assert rc == 0
    got = []
    while code.puncture[len(got)] >= 0:
        got.append(code.puncture[len(got)])
    with open(os.path.join(ROOT, "tests", "golden", "known_answers.json")) as f:
        assert got == json.load(f)["tch9_9k6_punctured"]
    assert got == list(orc.tch9_punct(2))
    C.CDLL(None).free(code.puncture)
    for s in range(16):
        assert [code.next_state[s][0], code.next_state[s][1]] == [(2 * s) & 15, (2 * s + 1) & 15]
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "known_answers.json")))["conv_spot"]["k5_12"]
    for s, row in g["next_output_rows"].items():
        assert [code.next_output[int(s)][0], code.next_output[int(s)][1]] == row


def test_plain_makefile_recipe(tmp_path):
    """The repository's Makefile (what a C maintainer runs instead of build.py; reference src/Makefile.am:1-24,
    src/sdr/Makefile.am:5-7, src/l1/Makefile.am:5-9): a dry run of `make all check install` names hipcc for every source build.py
    compiles, the pkg-config file comes out right, and `make install` of an (already built) library lays out lib/, include/ and
    lib/pkgconfig/ under DESTDIR.  (The compile itself is __graft_entry__.build()'s: the same commands.)"""
    import shutil
    import subprocess
    if shutil.which("make") is None:
        pytest.skip("no make")
    from __graft_entry__ import load_package
    pkg = load_package()
    pkg.build.build()                      # (a no-op when the library is up to date)
    r = subprocess.run(["make", "-n", "-B", "all", "check"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for src in pkg.build.HIP_SOURCES + pkg.build.CXX_SOURCES:
        assert f"osmo-gmr_amd/csrc/{src}" in r.stdout, src
    assert "--offload-arch=gfx950" in r.stdout and "-shared" in r.stdout and "abi_smoke" in r.stdout
    bdir = str(tmp_path / "b")
    r = subprocess.run(["make", f"BUILD={bdir}", f"{bdir}/gmr1_hip.pc", "PREFIX=/opt/gmr1"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    pc = open(os.path.join(bdir, "gmr1_hip.pc")).read()
    assert "prefix=/opt/gmr1" in pc and "Libs: -L${libdir} -lgmr1_hip" in pc and "Cflags: -I${includedir}" in pc
    # install: the library is there already (build()), so only the copy runs -- never a compile (-o: the objects are not remade)
    dest = str(tmp_path / "dest")
    r = subprocess.run(["make", "-o", pkg.build.LIB.replace(ROOT + "/", ""), f"BUILD={bdir}", "install", "PREFIX=/usr", f"DESTDIR={dest}"],
                       cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.exists(os.path.join(dest, "usr/lib/libgmr1_hip.so"))
    assert os.path.exists(os.path.join(dest, "usr/lib/pkgconfig/gmr1_hip.pc"))
    assert os.path.exists(os.path.join(dest, "usr/include/gmr1_hip.h"))
    assert os.path.exists(os.path.join(dest, "usr/include/osmocom/gmr1/sdr/pi4cxpsk.h"))
