"""ctypes binding of the CPU oracle (oracle/liborc.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liborc.so")

MAX_SYNC, MAX_CHUNKS, MAX_SYNC_SYMS = 4, 8, 32
BURST_IDS = ["bcch", "dc2", "dc6", "dc12", "nt3_speech", "nt3_facch", "nt6", "nt9", "rach", "sdcch"]


class Chunk(C.Structure):
    _fields_ = [("pos", C.c_int), ("len", C.c_int), ("syms", C.c_uint8 * MAX_SYNC_SYMS)]


class Burst(C.Structure):
    _fields_ = [
        ("name", C.c_char_p), ("rotation", C.c_float), ("nbits", C.c_int),
        ("guard_pre", C.c_int), ("guard_post", C.c_int), ("len", C.c_int), ("ebits", C.c_int),
        ("n_sync", C.c_int), ("n_sync_chunks", C.c_int * MAX_SYNC),
        ("sync", (Chunk * MAX_CHUNKS) * MAX_SYNC),
        ("n_data", C.c_int), ("data", Chunk * MAX_CHUNKS),
    ]


class FcchBurst(C.Structure):
    _fields_ = [("freq", C.c_float), ("len", C.c_int)]


def build(force: bool = False) -> str:
    if os.environ.get("ORC_LIBRARY"):                 # tests/test_sanitize.py: the same sources built with ASan + UBSan
        return os.environ["ORC_LIBRARY"]
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liborc.so"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_burst_get.restype = C.POINTER(Burst)
        _lib.orc_burst_get.argtypes = [C.c_int]
        # the checker follows the product's default: decision D1b (libosmocore's accelerated decoder where it dispatches
        # to it); ORC_CONV_MODE=0 (bench.py --conv-decoder generic) restates the generic decoder everywhere (D1)
        _lib.orc_conv_set_mode(C.c_int(0 if os.environ.get("ORC_CONV_MODE") == "0" else 1))
    return _lib


class conv_mode:
    """with oracle_lib.conv_mode(1): ...  -- decision D1b (libosmocore's accelerated Viterbi decoder for K in {5, 7},
    N in {2, 3, 4}; oracle/orc_3p_acc.c) instead of D1 (the generic decoder) inside the block."""
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = lib().orc_conv_get_mode()
        lib().orc_conv_set_mode(C.c_int(self.mode))

    def __exit__(self, *exc):
        lib().orc_conv_set_mode(C.c_int(self.prev))


class peak_stop_shift:
    """with oracle_lib.peak_stop_shift(+1): ... -- decision D3's early/late bisection one halving longer (-1: shorter)."""
    def __init__(self, steps):
        self.steps = steps

    def __enter__(self):
        lib().orc_peak_set_stop_shift(C.c_int(self.steps))

    def __exit__(self, *exc):
        lib().orc_peak_set_stop_shift(C.c_int(0))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def burst(name_or_id):
    i = BURST_IDS.index(name_or_id) if isinstance(name_or_id, str) else int(name_or_id)
    return lib().orc_burst_get(i)


def burst_format(name_or_id):
    """oracle burst table -> synth.BurstFormat (for cross checks)."""
    from __graft_entry__ import load_package
    synth = load_package().synth
    b = burst(name_or_id).contents
    sync = []
    for s in range(b.n_sync):
        chunks = []
        for c in range(b.n_sync_chunks[s]):
            ch = b.sync[s][c]
            chunks.append((ch.pos, [int(ch.syms[k]) for k in range(ch.len)]))
        sync.append(chunks)
    data = [(b.data[c].pos, b.data[c].len) for c in range(b.n_data)]
    return synth.BurstFormat(b.name.decode(), float(b.rotation), b.nbits, b.len, b.ebits, sync, data)


# ---- l1 ---------------------------------------------------------------------
def _enc(fn, l2, nbits, *extra):
    l2 = np.ascontiguousarray(l2, np.uint8)
    out = np.zeros((l2.shape[0], nbits), np.uint8)
    f = getattr(lib(), fn)
    for i in range(l2.shape[0]):
        f(_p(out[i], C.c_uint8), _p(l2[i], C.c_uint8), *extra)
    return out


def bcch_encode(l2):
    return _enc("orc_bcch_encode", l2, 424)


def ccch_encode(l2):
    return _enc("orc_ccch_encode", l2, 432)


def _dec24(fn, ebits):
    ebits = np.ascontiguousarray(ebits, np.int8)
    n = ebits.shape[0]
    l2 = np.zeros((n, 24), np.uint8)
    crc = np.zeros(n, np.int32)
    conv = np.zeros(n, np.int32)
    f = getattr(lib(), fn)
    f.restype = C.c_int
    cv = C.c_int()
    for i in range(n):
        crc[i] = f(_p(l2[i], C.c_uint8), _p(ebits[i], C.c_int8), C.byref(cv))
        conv[i] = cv.value
    return l2, crc, conv


def bcch_decode(ebits):
    return _dec24("orc_bcch_decode", ebits)


def ccch_decode(ebits):
    return _dec24("orc_ccch_decode", ebits)


def facch3_encode(l2, bits_s, ciph=None):
    l2 = np.ascontiguousarray(l2, np.uint8)
    bits_s = np.ascontiguousarray(bits_s, np.uint8)
    if ciph is not None:
        ciph = np.ascontiguousarray(ciph, np.uint8)
    out = np.zeros((l2.shape[0], 4, 104), np.uint8)
    for i in range(l2.shape[0]):
        lib().orc_facch3_encode(_p(out[i], C.c_uint8), _p(l2[i], C.c_uint8), _p(bits_s[i], C.c_uint8),
                                None if ciph is None else _p(ciph[i], C.c_uint8))
    return out


def facch3_decode(ebits):
    ebits = np.ascontiguousarray(ebits, np.int8).reshape(-1, 416)
    n = ebits.shape[0]
    l2 = np.zeros((n, 10), np.uint8)
    s = np.zeros((n, 32), np.uint8)
    crc = np.zeros(n, np.int32)
    conv = np.zeros(n, np.int32)
    f = lib().orc_facch3_decode
    f.restype = C.c_int
    cv = C.c_int()
    for i in range(n):
        crc[i] = f(_p(l2[i], C.c_uint8), _p(s[i], C.c_uint8), _p(ebits[i], C.c_int8), None, C.byref(cv))
        conv[i] = cv.value
    return l2, s, crc, conv


def tch3_encode(f0, f1, bits_s, m=0, ciph=None):
    f0 = np.ascontiguousarray(f0, np.uint8)
    f1 = np.ascontiguousarray(f1, np.uint8)
    bits_s = np.ascontiguousarray(bits_s, np.uint8)
    if ciph is not None:
        ciph = np.ascontiguousarray(ciph, np.uint8)
    out = np.zeros((f0.shape[0], 212), np.uint8)
    for i in range(f0.shape[0]):
        lib().orc_tch3_encode(_p(out[i], C.c_uint8), _p(f0[i], C.c_uint8), _p(f1[i], C.c_uint8),
                              _p(bits_s[i], C.c_uint8), None if ciph is None else _p(ciph[i], C.c_uint8), C.c_int(m))
    return out


def tch3_decode(ebits, m=0, ciph=None):
    ebits = np.ascontiguousarray(ebits, np.int8)
    n = ebits.shape[0]
    if ciph is not None:
        ciph = np.ascontiguousarray(ciph, np.uint8).reshape(n, 208)
    f0 = np.zeros((n, 10), np.uint8)
    f1 = np.zeros((n, 10), np.uint8)
    s = np.zeros((n, 4), np.uint8)
    c0 = np.zeros(n, np.int32)
    c1 = np.zeros(n, np.int32)
    a, b = C.c_int(), C.c_int()
    for i in range(n):
        lib().orc_tch3_decode(_p(f0[i], C.c_uint8), _p(f1[i], C.c_uint8), _p(s[i], C.c_uint8),
                              _p(ebits[i], C.c_int8), None if ciph is None else _p(ciph[i], C.c_uint8), C.c_int(m),
                              C.byref(a), C.byref(b))
        c0[i], c1[i] = a.value, b.value
    return f0, f1, s, c0, c1


def scramble_sbit(x):
    x = np.ascontiguousarray(x, np.int8)
    out = np.zeros_like(x)
    lib().orc_scramble_sbit(_p(out, C.c_int8), _p(x, C.c_int8), C.c_int(x.size))
    return out


def deinterleave_intra(x, N):
    x = np.ascontiguousarray(x, np.uint8)
    out = np.zeros_like(x)
    lib().orc_deinterleave_intra(_p(out, C.c_uint8), _p(x, C.c_uint8), C.c_int(N))
    return out


# ---- sdr ----------------------------------------------------------------------
def demod(bt, iq, sps, freq_shift=0.0):
    """single-burst demod -> dict(rv, ebits, sync_id, toa, freq_err, ssyms).  bt: a name / id of the oracle's table, or a
    Burst structure of the caller's (pointer)."""
    b = burst(bt) if isinstance(bt, (str, int)) else bt
    iq = np.ascontiguousarray(iq, np.complex64)
    eb = np.zeros(b.contents.ebits, np.int8)
    ss = np.zeros(b.contents.len, np.float32)
    sid, toa, fe = C.c_int(-1), C.c_float(), C.c_float()
    f = lib().orc_pi4cxpsk_demod
    f.restype = C.c_int
    rv = f(b, _p(iq, C.c_float), C.c_int(iq.size), C.c_int(sps), C.c_float(freq_shift),
           _p(eb, C.c_int8), C.byref(sid), C.byref(toa), C.byref(fe), _p(ss, C.c_float))
    return dict(rv=rv, ebits=eb, sync_id=sid.value, toa=toa.value, freq_err=fe.value, ssyms=ss)


def detect(bts, e_toa, iq, sps, freq_shift=0.0):
    arr = (C.POINTER(Burst) * len(bts))(*[burst(b) for b in bts])
    iq = np.ascontiguousarray(iq, np.complex64)
    bid, sid, toa = C.c_int(-1), C.c_int(-1), C.c_float()
    f = lib().orc_pi4cxpsk_detect
    f.restype = C.c_int
    rv = f(arr, C.c_int(len(bts)), C.c_float(e_toa), _p(iq, C.c_float), C.c_int(iq.size),
           C.c_int(sps), C.c_float(freq_shift), C.byref(bid), C.byref(sid), C.byref(toa))
    return dict(rv=rv, bt_id=bid.value, sync_id=sid.value, toa=toa.value)


def mod_order(iq, sps, freq_shift=0.0):
    iq = np.ascontiguousarray(iq, np.complex64)
    f = lib().orc_pi4cxpsk_mod_order
    f.restype = C.c_int
    return f(_p(iq, C.c_float), C.c_int(iq.size), C.c_int(sps), C.c_float(freq_shift))


def mod(bt, ebits, sync_id=0):
    b = burst(bt)
    ebits = np.ascontiguousarray(ebits, np.uint8)
    out = np.zeros(b.contents.len, np.complex64)
    lib().orc_pi4cxpsk_mod(b, _p(ebits, C.c_uint8), C.c_int(sync_id), _p(out, C.c_float))
    return out


def _fcch(which="fcch"):
    return FcchBurst.in_dll(lib(), {"fcch": "orc_fcch_burst", "fcch3_lband": "orc_fcch3_lband_burst",
                                    "fcch3_sband": "orc_fcch3_sband_burst"}[which])


def fcch_rough(iq, sps, freq_shift=0.0, which="fcch"):
    iq = np.ascontiguousarray(iq, np.complex64)
    toa = C.c_int()
    f = lib().orc_fcch_rough
    f.restype = C.c_int
    rv = f(C.byref(_fcch(which)), _p(iq, C.c_float), C.c_int(iq.size), C.c_int(sps), C.c_float(freq_shift), C.byref(toa))
    return rv, toa.value


def fcch_rough_multi(iq, sps, freq_shift=0.0, N=16, which="fcch"):
    iq = np.ascontiguousarray(iq, np.complex64)
    toas = np.zeros(N, np.int32)
    f = lib().orc_fcch_rough_multi
    f.restype = C.c_int
    rv = f(C.byref(_fcch(which)), _p(iq, C.c_float), C.c_int(iq.size), C.c_int(sps), C.c_float(freq_shift),
           _p(toas, C.c_int), C.c_int(N))
    return rv, toas[:max(rv, 0)].copy()


def fcch_fine(iq, sps, freq_shift=0.0, which="fcch"):
    iq = np.ascontiguousarray(iq, np.complex64)
    toa, fe = C.c_int(), C.c_float()
    f = lib().orc_fcch_fine
    f.restype = C.c_int
    rv = f(C.byref(_fcch(which)), _p(iq, C.c_float), C.c_int(iq.size), C.c_int(sps), C.c_float(freq_shift),
           C.byref(toa), C.byref(fe))
    return rv, toa.value, fe.value


def fcch_snr(iq, sps, freq_shift=0.0, which="fcch"):
    iq = np.ascontiguousarray(iq, np.complex64)
    snr = C.c_float()
    f = lib().orc_fcch_snr
    f.restype = C.c_int
    rv = f(C.byref(_fcch(which)), _p(iq, C.c_float), C.c_int(iq.size), C.c_int(sps), C.c_float(freq_shift), C.byref(snr))
    return rv, snr.value


def demod_decode_batch(iq, offset, kind, sps=4, freq_shift=None, want_ebits=True, want_ssyms=True):
    """BCCH/CCCH fused chain over a batch; iq is a flat complex64 array."""
    iq = np.ascontiguousarray(iq, np.complex64).reshape(-1)
    offset = np.ascontiguousarray(offset, np.uint64)
    kind = np.ascontiguousarray(kind, np.uint8)
    n = kind.size
    out = dict(l2=np.zeros((n, 24), np.uint8), crc=np.zeros(n, np.int32), conv=np.zeros(n, np.int32),
               toa=np.zeros(n, np.float32), freq_err=np.zeros(n, np.float32), rv=np.zeros(n, np.int32))
    eb = np.zeros((n, 432), np.int8) if want_ebits else None
    ss = np.zeros((n, 234), np.float32) if want_ssyms else None
    fs = np.ascontiguousarray(freq_shift, np.float32) if freq_shift is not None else None
    lib().orc_demod_decode_batch(
        C.c_int(n), _p(iq, C.c_float), _p(offset, C.c_uint64), _p(kind, C.c_uint8), C.c_int(sps),
        _p(fs, C.c_float) if fs is not None else None,
        _p(out["l2"], C.c_uint8), _p(out["crc"], C.c_int32), _p(out["conv"], C.c_int32),
        _p(out["toa"], C.c_float), _p(out["freq_err"], C.c_float),
        _p(eb, C.c_int8) if eb is not None else None,
        _p(ss, C.c_float) if ss is not None else None, _p(out["rv"], C.c_int32))
    out["ebits"], out["ssyms"] = eb, ss
    return out


# ---- receive control loop (gmr1_rx equivalent for one BCCH carrier) ----------------------------
RX_RECORD = np.dtype([("arfcn", "<u2"), ("chain", "u1"), ("type", "u1"), ("fn", "<u4"),
                      ("tn", "u1"), ("crc", "u1"), ("len", "u1"), ("pad", "u1"),
                      ("conv", "<i4"), ("l2", "u1", (24,))])


def rx_run(iq, sps=4, arfcn=0, max_records=4096):
    """orc_rx_run: FCCH acquisition + BCCH/CCCH frame loop of the reference's gmr1_rx on one carrier.
    Returns (rv, records (RX_RECORD array), n_chains)."""
    iq = np.ascontiguousarray(iq, np.complex64)
    out = np.zeros(max_records, RX_RECORD)
    n, nch = C.c_int(), C.c_int()
    f = lib().orc_rx_run
    f.restype = C.c_int
    rv = f(_p(iq, C.c_float), C.c_int(iq.size), C.c_int(sps), C.c_int(arfcn),
           out.ctypes.data_as(C.c_void_p), C.c_int(max_records), C.byref(n), C.byref(nch))
    return rv, out[:min(n.value, max_records)].copy(), nch.value


def dkab_demod(iq, sps=4, freq_shift=0.0, p=0):
    """orc_dkab_demod -> (rv, ebits[8], toa)"""
    iq = np.ascontiguousarray(iq, np.complex64)
    eb = np.zeros(8, np.int8)
    toa = C.c_float(0.0)
    f = lib().orc_dkab_demod
    f.restype = C.c_int
    rv = f(_p(iq, C.c_float), C.c_int(iq.size), C.c_int(sps), C.c_float(freq_shift), C.c_int(p),
           eb.ctypes.data_as(C.c_void_p), C.byref(toa))
    return rv, eb, toa.value


def a5(n, key, fn, nbits):
    """orc_a5 -> (dl bits, ul bits)"""
    key = np.ascontiguousarray(key, np.uint8)
    dl = np.zeros(nbits, np.uint8)
    ul = np.zeros(nbits, np.uint8)
    lib().orc_a5(C.c_int(n), key.ctypes.data_as(C.c_void_p), C.c_uint32(int(fn)), C.c_int(nbits),
                 dl.ctypes.data_as(C.c_void_p), ul.ctypes.data_as(C.c_void_p))
    return dl, ul


def rx_run_tch(iq, tch, sps=4, arfcn=0, kc=None, max_records=1 << 16):
    """orc_rx_run_tch: gmr1_rx with the traffic carrier `tch` (same length / timing as iq) and key kc."""
    iq = np.ascontiguousarray(iq, np.complex64)
    p_tch = None
    if tch is not None:
        tch = np.ascontiguousarray(tch, np.complex64)
        assert tch.size == iq.size
        p_tch = _p(tch, C.c_float)
    p_kc = None
    if kc is not None:
        kc = np.ascontiguousarray(kc, np.uint8)
        p_kc = kc.ctypes.data_as(C.c_void_p)
    out = np.zeros(max_records, RX_RECORD)
    n, nch = C.c_int(), C.c_int()
    f = lib().orc_rx_run_tch
    f.restype = C.c_int
    rv = f(_p(iq, C.c_float), p_tch, C.c_int(iq.size), C.c_int(sps), C.c_int(arfcn), p_kc,
           out.ctypes.data_as(C.c_void_p), C.c_int(max_records), C.byref(n), C.byref(nch))
    return rv, out[:min(n.value, max_records)].copy(), nch.value


class Interleaver(C.Structure):
    """struct orc_interleaver"""
    _fields_ = [("N", C.c_int), ("K", C.c_int), ("n", C.c_int), ("bits_cpp", C.c_uint8 * (3 * 648))]


def facch9_encode(l2, sacch, status, ciph=None):
    l2 = np.ascontiguousarray(l2, np.uint8)
    e = np.zeros(662, np.uint8)
    sacch = np.ascontiguousarray(sacch, np.uint8)
    status = np.ascontiguousarray(status, np.uint8)
    cp = None if ciph is None else np.ascontiguousarray(ciph, np.uint8).ctypes.data_as(C.c_void_p)
    lib().orc_facch9_encode(e.ctypes.data_as(C.c_void_p), l2.ctypes.data_as(C.c_void_p),
                            sacch.ctypes.data_as(C.c_void_p), status.ctypes.data_as(C.c_void_p), cp)
    return e


def facch9_decode(ebits, ciph=None):
    """-> (l2[38], sacch[10] sbits, status[4] sbits, crc, conv)"""
    ebits = np.ascontiguousarray(ebits, np.int8)
    l2 = np.zeros(38, np.uint8)
    sacch = np.zeros(10, np.int8)
    status = np.zeros(4, np.int8)
    conv = C.c_int(0)
    cp = None if ciph is None else np.ascontiguousarray(ciph, np.uint8).ctypes.data_as(C.c_void_p)
    f = lib().orc_facch9_decode
    f.restype = C.c_int
    crc = f(l2.ctypes.data_as(C.c_void_p), sacch.ctypes.data_as(C.c_void_p), status.ctypes.data_as(C.c_void_p),
            ebits.ctypes.data_as(C.c_void_p), cp, C.byref(conv))
    return l2, sacch, status, crc, conv.value


TCH9_BYTES = (18, 30, 60)      # 2k4, 4k8, 9k6 (tch9.c:92-94)


def tch9_encode_seq(l2s, mode, sacch=None, status=None, ciph=None):
    """Encode a sequence of TCH9 blocks of one channel (inter-burst interleaver carried along) -> (n, 662)"""
    l2s = np.ascontiguousarray(l2s, np.uint8)
    n = l2s.shape[0]
    il = Interleaver()
    lib().orc_interleaver_init(C.byref(il), C.c_int(3), C.c_int(648))
    out = np.zeros((n, 662), np.uint8)
    z10, z4 = np.zeros(10, np.uint8), np.zeros(4, np.uint8)
    for i in range(n):
        sa = z10 if sacch is None else np.ascontiguousarray(sacch[i], np.uint8)
        stt = z4 if status is None else np.ascontiguousarray(status[i], np.uint8)
        cp = None if ciph is None else np.ascontiguousarray(ciph[i], np.uint8).ctypes.data_as(C.c_void_p)
        lib().orc_tch9_encode(out[i].ctypes.data_as(C.c_void_p), l2s[i].ctypes.data_as(C.c_void_p), C.c_int(mode),
                              sa.ctypes.data_as(C.c_void_p), stt.ctypes.data_as(C.c_void_p), cp, C.byref(il))
    return out


def tch9_decode_seq(ebits, mode, ciph=None):
    """Decode a sequence of TCH9 bursts of one channel -> (l2 (n, bytes), sacch (n, 10), status (n, 4), conv (n,))"""
    ebits = np.ascontiguousarray(ebits, np.int8)
    n = ebits.shape[0]
    nb = TCH9_BYTES[mode]
    il = Interleaver()
    lib().orc_interleaver_init(C.byref(il), C.c_int(3), C.c_int(648))
    l2 = np.zeros((n, nb), np.uint8)
    sacch = np.zeros((n, 10), np.int8)
    status = np.zeros((n, 4), np.int8)
    conv = np.zeros(n, np.int32)
    for i in range(n):
        cv = C.c_int(0)
        cp = None if ciph is None else np.ascontiguousarray(ciph[i], np.uint8).ctypes.data_as(C.c_void_p)
        lib().orc_tch9_decode(l2[i].ctypes.data_as(C.c_void_p), sacch[i].ctypes.data_as(C.c_void_p),
                              status[i].ctypes.data_as(C.c_void_p), ebits[i].ctypes.data_as(C.c_void_p),
                              C.c_int(mode), cp, C.byref(il), C.byref(cv))
        conv[i] = cv.value
    return l2, sacch, status, conv


def tch9_punct(mode):
    idx = (C.c_int * 1024)()
    f = lib().orc_tch9_punct
    f.restype = C.c_int
    n = f(C.c_int(mode), idx)
    return np.array(idx[:n], np.int64)


def xch_dc12_encode(l2):
    l2 = np.ascontiguousarray(l2, np.uint8)
    e = np.zeros(432, np.uint8)
    lib().orc_xch_dc12_encode(e.ctypes.data_as(C.c_void_p), l2.ctypes.data_as(C.c_void_p))
    return e


def xch_dc12_decode(ebits):
    """-> (l2[24], crc, conv)"""
    ebits = np.ascontiguousarray(ebits, np.int8)
    assert ebits.size == 432
    l2 = np.zeros(24, np.uint8)
    conv = C.c_int(0)
    f = lib().orc_xch_dc12_decode
    f.restype = C.c_int
    crc = f(l2.ctypes.data_as(C.c_void_p), ebits.ctypes.data_as(C.c_void_p), C.byref(conv))
    return l2, crc, conv.value


def rach_encode(rach, sb_mask):
    rach = np.ascontiguousarray(rach, np.uint8)
    assert rach.size == 18
    e = np.zeros(494, np.uint8)
    lib().orc_rach_encode(e.ctypes.data_as(C.c_void_p), rach.ctypes.data_as(C.c_void_p), C.c_uint8(sb_mask))
    return e


def rach_decode(ebits, sb_mask):
    """-> (rach[18], rv, conv, (crc8, crc12))"""
    ebits = np.ascontiguousarray(ebits, np.int8)
    assert ebits.size == 494
    rach = np.zeros(18, np.uint8)
    conv = C.c_int(0)
    crc = (C.c_int * 2)()
    f = lib().orc_rach_decode
    f.restype = C.c_int
    rv = f(rach.ctypes.data_as(C.c_void_p), ebits.ctypes.data_as(C.c_void_p), C.c_uint8(sb_mask),
           C.byref(conv), crc)
    return rach, rv, conv.value, (crc[0], crc[1])


RX_BIG_RECORD = np.dtype([("arfcn", "<u2"), ("chain", "u1"), ("type", "u1"), ("fn", "<u4"),
                          ("tn", "u1"), ("crc", "u1"), ("len", "u1"), ("pad", "u1"),
                          ("conv", "<i4"), ("l2", "u1", (64,))])


def rx_run_full(iq, tch, csd, sps=4, arfcn=0, kc=None, max_records=1 << 16, max_big=1 << 14):
    """orc_rx_run_full: gmr1_rx with traffic carrier, key and CSD carrier -> (rv, records, big records, n_chains)"""
    iq = np.ascontiguousarray(iq, np.complex64)

    def ptr(x):
        if x is None:
            return None
        x = np.ascontiguousarray(x, np.complex64)
        assert x.size == iq.size
        keep.append(x)
        return _p(x, C.c_float)
    keep = []
    p_tch, p_csd = ptr(tch), ptr(csd)
    p_kc = None
    if kc is not None:
        kc = np.ascontiguousarray(kc, np.uint8)
        p_kc = kc.ctypes.data_as(C.c_void_p)
    out = np.zeros(max_records, RX_RECORD)
    big = np.zeros(max_big, RX_BIG_RECORD)
    n, nb, nch = C.c_int(), C.c_int(), C.c_int()
    f = lib().orc_rx_run_full
    f.restype = C.c_int
    rv = f(_p(iq, C.c_float), p_tch, p_csd, C.c_int(iq.size), C.c_int(sps), C.c_int(arfcn), p_kc,
           out.ctypes.data_as(C.c_void_p), C.c_int(max_records), C.byref(n),
           big.ctypes.data_as(C.c_void_p), C.c_int(max_big), C.byref(nb), C.byref(nch))
    return rv, out[:min(n.value, max_records)].copy(), big[:min(nb.value, max_big)].copy(), nch.value


# ---- AMBE speech decoder (oracle/orc_ambe.c) ----

def ambe_state_size():
    f = lib().orc_ambe_state_size
    f.restype = C.c_size_t
    return f()


class AmbeDecoder:
    """One oracle decoder; `cleared=True` selects the other reading of decision D9 (voicing entries above L are 0)."""

    def __init__(self, cleared=False):
        self.buf = C.create_string_buffer(ambe_state_size())
        lib().orc_ambe_init(self.buf)
        lib().orc_ambe_set_cleared(self.buf, C.c_int(1 if cleared else 0))

    def decode(self, frames):
        """frames [n, 10] uint8 -> (pcm [n, 160] int16, rv [n] int32)"""
        frames = np.ascontiguousarray(frames, np.uint8).reshape(-1, 10)
        n = len(frames)
        pcm = np.zeros((n, 160), np.int16)
        rv = np.zeros(n, np.int32)
        lib().orc_ambe_decode_stream(self.buf, _p(frames, C.c_uint8), C.c_int(n), _p(pcm, C.c_int16), _p(rv, C.c_int32))
        return pcm, rv

    def decode_frame(self, frame, N=160):
        frame = np.ascontiguousarray(frame, np.uint8)
        pcm = np.zeros(max(N, 160), np.int16)
        rv = lib().orc_ambe_decode_frame(self.buf, _p(pcm, C.c_int16), C.c_int(N), _p(frame, C.c_uint8), C.c_int(0))
        return pcm[:max(N, 160)], rv

    def state_words(self):
        return np.frombuffer(self.buf.raw, np.uint32).copy()


def ambe_decode(frames, cleared=False):
    return AmbeDecoder(cleared).decode(frames)


def ambe_unpack(frame):
    out = (C.c_uint * 14)()
    lib().orc_ambe_unpack(_p(np.ascontiguousarray(frame, np.uint8), C.c_uint8), out)
    return list(out)
