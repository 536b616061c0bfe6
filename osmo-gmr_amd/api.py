"""ctypes mirror of the C ABI in include/gmr1_hip.h (libgmr1_hip.so).

This is plumbing only: every function forwards to the shared library, which runs
HIP kernels.  There is no Python or CPU implementation behind these calls --
if the library is missing or no GPU is usable they raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

MAX_SYNC, MAX_CHUNKS, MAX_SYNC_SYMS = 4, 8, 32
BURST_IDS = ["bcch", "dc2", "dc6", "dc12", "nt3_speech", "nt3_facch", "nt6", "nt9", "rach", "sdcch"]

# every symbol include/gmr1_hip.h and include/osmocom/gmr1/**.h declare
EXPORTED_FUNCTIONS = [
    "gmr1_hip_init", "gmr1_hip_last_error", "gmr1_hip_version", "gmr1_hip_burst_info",
    "gmr1_hip_set_conv_decoder", "gmr1_hip_get_conv_decoder", "gmr1_hip_clock_probe_dev", "gmr1_hip_rx_run_last_timing",
    "gmr1_hip_demod_batch_dev", "gmr1_hip_demod_batch", "gmr1_hip_demod_taps",
    "gmr1_hip_bcch_decode_batch_dev", "gmr1_hip_ccch_decode_batch_dev",
    "gmr1_hip_bcch_decode_batch", "gmr1_hip_ccch_decode_batch",
    "gmr1_hip_rx_bcch_ccch_batch_dev", "gmr1_hip_rx_bcch_ccch_batch",
    "gmr1_hip_rx_bcch_ccch_batch_planar_dev", "gmr1_hip_iq_to_planar_dev",
    "gmr1_pi4cxpsk_demod", "gmr1_bcch_decode", "gmr1_ccch_decode",
    "gmr1_hip_fcch_rough_batch_dev", "gmr1_hip_fcch_rough_batch",
    "gmr1_hip_fcch_fine_batch_dev", "gmr1_hip_fcch_fine_batch",
    "gmr1_hip_fcch_snr_batch_dev", "gmr1_hip_fcch_snr_batch",
    "gmr1_fcch_rough", "gmr1_fcch_fine", "gmr1_fcch_snr", "gmr1_fcch_rough_multi",
    "gmr1_hip_fcch_rough_multi_batch_dev", "gmr1_hip_fcch_rough_multi_batch",
    "gmr1_hip_facch3_decode_batch_dev", "gmr1_hip_facch3_decode_batch",
    "gmr1_hip_tch3_decode_batch_dev", "gmr1_hip_tch3_decode_batch", "gmr1_hip_tch3_rx_batch_dev", "gmr1_hip_tch3_rx_batch",
    "gmr1_facch3_decode", "gmr1_tch3_decode",
    "gmr1_hip_detect_batch_dev", "gmr1_hip_detect_batch",
    "gmr1_hip_mod_order_batch_dev", "gmr1_hip_mod_order_batch",
    "gmr1_pi4cxpsk_detect", "gmr1_pi4cxpsk_mod_order",
    "gmr1_hip_rx_run_dev", "gmr1_hip_rx_run", "gmr1_hip_gsmtap_pack",
    "gmr1_hip_rx_run_tch_dev", "gmr1_hip_rx_run_tch", "gmr1_hip_rx_run_full_dev", "gmr1_hip_rx_run_full", "gmr1_hip_gsmtap_pack_big",
    "gmr1_hip_channelize_plan", "gmr1_hip_channelize_dev", "gmr1_hip_channelize", "gmr1_hip_channelize_planar_dev",
    "gmr1_hip_facch9_decode_batch_dev", "gmr1_hip_facch9_decode_batch", "gmr1_facch9_decode",
    "gmr1_hip_tch9_decode_batch_dev", "gmr1_hip_tch9_decode_batch",
    "gmr1_tch9_decode", "gmr1_interleaver_init", "gmr1_interleaver_fini",
    "gmr1_hip_dkab_demod_batch_dev", "gmr1_hip_dkab_demod_batch", "gmr1_dkab_demod",
    "gmr1_hip_a5_batch_dev", "gmr1_hip_a5_batch", "gmr1_a5", "gmr1_a5_1",
    "gmr1_hip_xch_dc12_decode_batch_dev", "gmr1_hip_xch_dc12_decode_batch", "gmr1_xch_dc12_decode",
    "gmr1_hip_rach_decode_batch_dev", "gmr1_hip_rach_decode_batch", "gmr1_rach_decode",
    "gmr1_hip_bcch_encode_batch_dev", "gmr1_hip_bcch_encode_batch", "gmr1_bcch_encode",
    "gmr1_hip_ccch_encode_batch_dev", "gmr1_hip_ccch_encode_batch", "gmr1_ccch_encode",
    "gmr1_hip_xch_dc12_encode_batch_dev", "gmr1_hip_xch_dc12_encode_batch", "gmr1_xch_dc12_encode",
    "gmr1_hip_facch3_encode_batch_dev", "gmr1_hip_facch3_encode_batch", "gmr1_facch3_encode",
    "gmr1_hip_tch3_encode_batch_dev", "gmr1_hip_tch3_encode_batch", "gmr1_tch3_encode",
    "gmr1_hip_facch9_encode_batch_dev", "gmr1_hip_facch9_encode_batch", "gmr1_facch9_encode",
    "gmr1_hip_tch9_encode_batch_dev", "gmr1_hip_tch9_encode_batch", "gmr1_tch9_encode",
    "gmr1_hip_rach_encode_batch_dev", "gmr1_hip_rach_encode_batch", "gmr1_rach_encode",
    "gmr1_hip_mod_batch_dev", "gmr1_hip_mod_batch", "gmr1_pi4cxpsk_mod", "gmr1_hip_encoder_plan",
    "gmr1_scramble_sbit", "gmr1_scramble_ubit", "gmr1_interleave_intra", "gmr1_deinterleave_intra",
    "gmr1_interleave_inter", "gmr1_deinterleave_inter",
    "gmr1_puncturer_generate",
    "gmr1_hip_ddc_plan", "gmr1_hip_ddc_dev", "gmr1_hip_ddc",
    "gmr1_hip_shard_unique_id", "gmr1_hip_shard_create", "gmr1_hip_shard_adopt", "gmr1_hip_shard_destroy",
    "gmr1_hip_rx_run_sharded", "gmr1_hip_rx_run_sharded_resident",
    "gmr1_codec_alloc", "gmr1_codec_release", "gmr1_codec_decode_frame", "gmr1_codec_decode_dtx",
    "gmr1_hip_codec_state_bytes", "gmr1_hip_codec_init_dev", "gmr1_hip_codec_decode_batch_dev",
    "gmr1_hip_codec_decode_batch", "gmr1_hip_codec_host_tables", "gmr1_hip_codec_libm_check",
]
EXPORTED_DATA = [
    "gmr1_pi2cbpsk", "gmr1_pi4cbpsk", "gmr1_pi4cqpsk",
    "gmr1_bcch_burst", "gmr1_dc2_burst", "gmr1_dc6_burst", "gmr1_dc12_burst",
    "gmr1_nt3_speech_burst", "gmr1_nt3_facch_burst", "gmr1_nt6_burst", "gmr1_nt9_burst",
    "gmr1_rach_burst", "gmr1_sdcch_burst",
    "gmr1_fcch_burst", "gmr1_fcch3_lband_burst", "gmr1_fcch3_sband_burst",
    # code descriptions for libosmocore's own codec (l1/conv.h, l1/crc.h, l1/punct.h): host data, not used by the kernels
    "gmr1_conv_k5_12", "gmr1_conv_k5_13", "gmr1_conv_k5_14", "gmr1_conv_k5_15", "gmr1_conv_k6_14", "gmr1_conv_k9_12",
    "gmr1_conv_k9_13", "gmr1_conv_k9_14", "gmr1_conv_tch3", "gmr1_crc8", "gmr1_crc12", "gmr1_crc16",
] + ["gmr1_punct_" + _n for _n in """
    k5_12_P23 k5_12_P25 k5_12_Ps25 k5_12_P311 k5_12_P412 k5_12_Ps412 k5_12_P12 k5_12_Ps12 k5_12_A k5_12_B
    k5_12_C k5_12_D k5_12_E k5_12_P38 k5_12_P26 k5_12_P37 k5_13_P16 k5_13_P25 k5_13_P15 k5_13_Ps15 k5_13_P78
    k5_15_P23 k5_15_P53 k5_15_Ps53 k7_12_P23 k7_12_P410 k7_12_P512 k7_12_P116 k7_12_P148 k7_12_P184
    k7_12_P1152 k7_12_P45 k7_12_P245 k9_12_P13 k9_12_P47 k9_12_P34 k9_12_P17 k9_12_P19 k9_12_P26 k9_12_P110
    k9_12_P14 k9_12_P45 k9_12_P234 k6_14_P45 k9_14_P148 k9_14_P65 k9_13_P12 k9_13_P1213 k9_13_P44 k9_13_P33
    k9_13_P65
""".split()]


class Chunk(C.Structure):
    _fields_ = [("pos", C.c_int32), ("len", C.c_int32), ("syms", C.c_uint8 * MAX_SYNC_SYMS)]


class BurstFlat(C.Structure):
    _fields_ = [
        ("name", C.c_char * 16), ("rotation", C.c_float), ("nbits", C.c_int32),
        ("guard_pre", C.c_int32), ("guard_post", C.c_int32), ("len", C.c_int32), ("ebits", C.c_int32),
        ("n_sync", C.c_int32), ("n_sync_chunks", C.c_int32 * MAX_SYNC),
        ("sync", (Chunk * MAX_CHUNKS) * MAX_SYNC),
        ("n_data", C.c_int32), ("data", Chunk * MAX_CHUNKS),
    ]


class CxVec(C.Structure):
    """struct osmo_cxvec (include/osmocom/gmr1/compat.h)."""
    _fields_ = [("len", C.c_int), ("max_len", C.c_int), ("flags", C.c_int), ("data", C.c_void_p)]


class Gmr1HipError(RuntimeError):
    pass


_lib = None


def lib_path() -> str:
    # tools/ (never the tests, never bench.py's default run) may point at the profiling build: build.py --profile
    return os.environ.get("GMR1_HIP_LIBRARY") or _build.LIB


def load(build_if_missing: bool = False):
    """dlopen libgmr1_hip.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        if build_if_missing:
            _build.build()
        else:
            raise Gmr1HipError(
                f"{path} is missing: the HIP extension has not been built "
                "(run __graft_entry__.build()); there is no CPU fallback")
    _lib = C.CDLL(path)
    _lib.gmr1_hip_last_error.restype = C.c_char_p
    _lib.gmr1_hip_version.restype = C.c_char_p
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = load().gmr1_hip_last_error().decode(errors="replace")
        raise Gmr1HipError(f"{what} failed with {rc}: {msg}")


def _np(a, dtype):
    a = np.ascontiguousarray(a, dtype)
    return a, a.ctypes.data_as(C.c_void_p)


def burst_info(name_or_id) -> BurstFlat:
    i = BURST_IDS.index(name_or_id) if isinstance(name_or_id, str) else int(name_or_id)
    out = BurstFlat()
    _check(load().gmr1_hip_burst_info(C.c_int(i), C.byref(out)), "gmr1_hip_burst_info")
    return out


def burst_format(name_or_id):
    """Product burst table -> synth.BurstFormat."""
    from . import synth
    b = burst_info(name_or_id)
    sync = []
    for s in range(b.n_sync):
        sync.append([(b.sync[s][c].pos, [int(b.sync[s][c].syms[k]) for k in range(b.sync[s][c].len)])
                     for c in range(b.n_sync_chunks[s])])
    data = [(b.data[c].pos, b.data[c].len) for c in range(b.n_data)]
    return synth.BurstFormat(b.name.decode(), float(b.rotation), b.nbits, b.len, b.ebits, sync, data)


CONV_GENERIC, CONV_ACC = 0, 1


def set_conv_decoder(decoder: int):
    """gmr1_hip_set_conv_decoder: which libosmocore Viterbi decoder the layer-1 chains reproduce (process-wide)."""
    _check(load().gmr1_hip_set_conv_decoder(C.c_int(int(decoder))), "gmr1_hip_set_conv_decoder")


def get_conv_decoder() -> int:
    return int(load().gmr1_hip_get_conv_decoder())


class conv_decoder:
    """with api.conv_decoder(api.CONV_ACC): ...  -- switch the decoder for a block (tests, bench)."""

    def __init__(self, decoder):
        self.decoder = int(decoder)

    def __enter__(self):
        self.prev = get_conv_decoder()
        set_conv_decoder(self.decoder)
        return self

    def __exit__(self, *exc):
        set_conv_decoder(self.prev)
        return False


def init(device: int = 0):
    _check(load().gmr1_hip_init(C.c_int(device)), "gmr1_hip_init")


def clock_probe_dev(stream, micros=200):
    """(shader clock held right now in MHz, wall counter rate in MHz), measured on the device behind what `stream` holds."""
    core, wall = C.c_double(), C.c_double()
    f = load().gmr1_hip_clock_probe_dev
    f.restype = C.c_int
    _check(f(C.c_void_p(stream), C.c_int(micros), C.byref(core), C.byref(wall)), "gmr1_hip_clock_probe_dev")
    return core.value, wall.value


# ---------------------------------------------------------------------------
# host-pointer batch calls (numpy in, numpy out)
# ---------------------------------------------------------------------------
def demod_batch(burst, iq, offset, in_len, sps=4, freq_shift=None, want_ssyms=True):
    bid = BURST_IDS.index(burst) if isinstance(burst, str) else int(burst)
    info = burst_info(bid)
    iq, p_iq = _np(np.asarray(iq).reshape(-1), np.complex64)
    offset, p_off = _np(offset, np.uint64)
    n = offset.size
    fs_p = None
    if freq_shift is not None:
        fs, fs_p = _np(freq_shift, np.float32)
    eb = np.zeros((n, info.ebits), np.int8)
    sid = np.zeros(n, np.int32)
    toa = np.zeros(n, np.float32)
    fe = np.zeros(n, np.float32)
    ss = np.zeros((n, info.len), np.float32) if want_ssyms else None
    rv = np.zeros(n, np.int32)
    rc = load().gmr1_hip_demod_batch(
        C.c_int(bid), C.c_int(n), C.c_int(sps), C.c_int(in_len), p_iq, C.c_uint64(iq.size), p_off, fs_p,
        eb.ctypes.data_as(C.c_void_p), C.c_int(info.ebits), sid.ctypes.data_as(C.c_void_p),
        toa.ctypes.data_as(C.c_void_p), fe.ctypes.data_as(C.c_void_p),
        ss.ctypes.data_as(C.c_void_p) if ss is not None else None, rv.ctypes.data_as(C.c_void_p))
    _check(rc, "gmr1_hip_demod_batch")
    return dict(rv=rv, ebits=eb, sync_id=sid, toa=toa, freq_err=fe, ssyms=ss)


def demod_taps(burst, iq, sps=4, freq_shift=0.0):
    """One burst through gmr1_hip_demod_taps: the batch entry's outputs plus the four vectors the reference dumps under
    ENABLE_DEBUG_SIGNAL (sdr/defs.h:35-39): corr (pi4cxpsk.c:251), burst (:545), align (:345), final (:582)."""
    bid = BURST_IDS.index(burst) if isinstance(burst, str) else int(burst)
    info = burst_info(bid)
    iq, p_iq = _np(np.asarray(iq).reshape(-1), np.complex64)
    in_len = iq.size
    w = in_len - info.len * sps + 1
    out = dict(corr=np.zeros(max(w, 0), np.float32), burst=np.zeros(in_len, np.complex64),
               align=np.zeros(info.len, np.complex64), final=np.zeros(info.len, np.complex64),
               ebits=np.zeros(info.ebits, np.int8), sync_id=np.zeros(1, np.int32), toa=np.zeros(1, np.float32),
               freq_err=np.zeros(1, np.float32), ssyms=np.zeros(info.len, np.float32), rv=np.zeros(1, np.int32))
    ptr = lambda k: out[k].ctypes.data_as(C.c_void_p)
    rc = load().gmr1_hip_demod_taps(
        C.c_int(bid), C.c_int(sps), C.c_int(in_len), p_iq, C.c_float(freq_shift),
        ptr("corr"), ptr("burst"), ptr("align"), ptr("final"),
        ptr("ebits"), ptr("sync_id"), ptr("toa"), ptr("freq_err"), ptr("ssyms"), ptr("rv"))
    _check(rc, "gmr1_hip_demod_taps")
    for k in ("sync_id", "toa", "freq_err", "rv"):
        out[k] = out[k][0]
    return out


def _l1_batch(fn, ebits, neb):
    ebits, p = _np(ebits, np.int8)
    ebits = ebits.reshape(-1, neb)
    n = ebits.shape[0]
    l2 = np.zeros((n, 24), np.uint8)
    crc = np.zeros(n, np.int32)
    conv = np.zeros(n, np.int32)
    rc = getattr(load(), fn)(C.c_int(n), p, l2.ctypes.data_as(C.c_void_p),
                             crc.ctypes.data_as(C.c_void_p), conv.ctypes.data_as(C.c_void_p))
    _check(rc, fn)
    return l2, crc, conv


def bcch_decode_batch(ebits):
    return _l1_batch("gmr1_hip_bcch_decode_batch", ebits, 424)


def ccch_decode_batch(ebits):
    return _l1_batch("gmr1_hip_ccch_decode_batch", ebits, 432)


def rx_bcch_ccch_batch(iq, offset, kind, sps=4, freq_shift=None, want_ebits=True, want_ssyms=True):
    iq, p_iq = _np(np.asarray(iq).reshape(-1), np.complex64)
    offset, p_off = _np(offset, np.uint64)
    kind, p_kind = _np(kind, np.uint8)
    n = kind.size
    fs_p = None
    if freq_shift is not None:
        fs, fs_p = _np(freq_shift, np.float32)
    out = dict(l2=np.zeros((n, 24), np.uint8), crc=np.zeros(n, np.int32), conv=np.zeros(n, np.int32),
               toa=np.zeros(n, np.float32), freq_err=np.zeros(n, np.float32), rv=np.zeros(n, np.int32))
    eb = np.zeros((n, 432), np.int8) if want_ebits else None
    ss = np.zeros((n, 234), np.float32) if want_ssyms else None
    rc = load().gmr1_hip_rx_bcch_ccch_batch(
        C.c_int(n), C.c_int(sps), p_iq, C.c_uint64(iq.size), p_off, p_kind, fs_p,
        out["l2"].ctypes.data_as(C.c_void_p), out["crc"].ctypes.data_as(C.c_void_p),
        out["conv"].ctypes.data_as(C.c_void_p), out["toa"].ctypes.data_as(C.c_void_p),
        out["freq_err"].ctypes.data_as(C.c_void_p),
        eb.ctypes.data_as(C.c_void_p) if eb is not None else None,
        ss.ctypes.data_as(C.c_void_p) if ss is not None else None,
        out["rv"].ctypes.data_as(C.c_void_p))
    _check(rc, "gmr1_hip_rx_bcch_ccch_batch")
    out["ebits"], out["ssyms"] = eb, ss
    return out


# ---------------------------------------------------------------------------
# reference-style single-burst calls (the legacy C API, through ctypes)
# ---------------------------------------------------------------------------
def pi4cxpsk_demod(burst_name, iq, sps=4, freq_shift=0.0):
    """gmr1_pi4cxpsk_demod(&gmr1_<name>_burst, cxvec, ...) exactly as C callers use it; burst_name may also be a
    CallerBurst (the caller's own description of a format)."""
    L = load()
    if isinstance(burst_name, CallerBurst):
        bt_addr, n_eb = burst_name.address, int(burst_name.burst.ebits)
    else:
        bt = C.c_void_p.in_dll(L, f"gmr1_{burst_name}_burst")   # address of the exported struct
        bt_addr, n_eb = C.addressof(bt), burst_info(burst_name).ebits
    iq = np.ascontiguousarray(iq, np.complex64)
    vec = CxVec(iq.size, iq.size, 0, iq.ctypes.data_as(C.c_void_p))
    eb = np.zeros(n_eb, np.int8)
    sid, toa, fe = C.c_int(-1), C.c_float(), C.c_float()
    f = L.gmr1_pi4cxpsk_demod
    f.restype = C.c_int
    rv = f(C.c_void_p(bt_addr), C.byref(vec), C.c_int(sps), C.c_float(freq_shift),
           eb.ctypes.data_as(C.c_void_p), C.byref(sid), C.byref(toa), C.byref(fe))
    return dict(rv=rv, ebits=eb, sync_id=sid.value, toa=toa.value, freq_err=fe.value)


def _decode1(fn, ebits):
    ebits = np.ascontiguousarray(ebits, np.int8)
    l2 = np.zeros(24, np.uint8)
    cv = C.c_int()
    f = getattr(load(), fn)
    f.restype = C.c_int
    rv = f(l2.ctypes.data_as(C.c_void_p), ebits.ctypes.data_as(C.c_void_p), C.byref(cv))
    return l2, rv, cv.value


def bcch_decode(ebits):
    return _decode1("gmr1_bcch_decode", ebits)


def ccch_decode(ebits):
    return _decode1("gmr1_ccch_decode", ebits)


# ---------------------------------------------------------------------------
# device-pointer calls (torch tensors as plumbing for HBM + streams)
# ---------------------------------------------------------------------------
def rx_bcch_ccch_batch_dev(stream, n, sps, iq, offset, kind, freq_shift, l2, crc, conv, toa, freq_err,
                           ebits, ssyms, rv):
    """All tensor arguments are device pointers (ints) or None."""
    f = load().gmr1_hip_rx_bcch_ccch_batch_dev
    f.restype = C.c_int
    vp = lambda x: C.c_void_p(x) if x else None
    rc = f(vp(stream), C.c_int(n), C.c_int(sps), vp(iq), vp(offset), vp(kind), vp(freq_shift),
           vp(l2), vp(crc), vp(conv), vp(toa), vp(freq_err), vp(ebits), vp(ssyms), vp(rv))
    _check(rc, "gmr1_hip_rx_bcch_ccch_batch_dev")


def rx_bcch_ccch_batch_planar_dev(stream, n, sps, iq_planes, plane_stride, offset, kind, freq_shift, l2, crc, conv, toa,
                                  freq_err, ebits, ssyms, rv):
    """The same call on a polyphase-planar sample array (include/gmr1_hip.h); device pointers (ints) or None."""
    f = load().gmr1_hip_rx_bcch_ccch_batch_planar_dev
    f.restype = C.c_int
    vp = lambda x: C.c_void_p(x) if x else None
    rc = f(vp(stream), C.c_int(n), C.c_int(sps), vp(iq_planes), C.c_uint64(plane_stride), vp(offset), vp(kind), vp(freq_shift),
           vp(l2), vp(crc), vp(conv), vp(toa), vp(freq_err), vp(ebits), vp(ssyms), vp(rv))
    _check(rc, "gmr1_hip_rx_bcch_ccch_batch_planar_dev")


def iq_to_planar_dev(stream, sps, n_samples, iq, iq_planes, plane_stride):
    """Interleaved device sample array -> polyphase-planar (sample s to iq_planes[(s % sps) * plane_stride + s // sps])."""
    f = load().gmr1_hip_iq_to_planar_dev
    f.restype = C.c_int
    vp = lambda x: C.c_void_p(x) if x else None
    rc = f(vp(stream), C.c_int(sps), C.c_uint64(n_samples), vp(iq), vp(iq_planes), C.c_uint64(plane_stride))
    _check(rc, "gmr1_hip_iq_to_planar_dev")


# ---------------------------------------------------------------------------
# FCCH acquisition
# ---------------------------------------------------------------------------
FCCH_TYPES = ["fcch", "fcch3_lband", "fcch3_sband"]
FCCH_LEN = [117, 468, 468]


class FcchBurst(C.Structure):
    """struct gmr1_fcch_burst (include/osmocom/gmr1/sdr/fcch.h)."""
    _fields_ = [("freq", C.c_float), ("len", C.c_int)]


def _fcch_id(t):
    return FCCH_TYPES.index(t) if isinstance(t, str) else int(t)


def fcch_rough_batch(iq, offset, length, sps=4, freq_shift=None, fcch_type="fcch"):
    iq, p_iq = _np(np.asarray(iq).reshape(-1), np.complex64)
    offset, p_off = _np(offset, np.uint64)
    n = offset.size
    fs_p = None
    if freq_shift is not None:
        fs, fs_p = _np(freq_shift, np.float32)
    toa = np.zeros(n, np.int32)
    rv = np.zeros(n, np.int32)
    rc = load().gmr1_hip_fcch_rough_batch(
        C.c_int(_fcch_id(fcch_type)), C.c_int(n), C.c_int(sps), C.c_int(length), p_iq, C.c_uint64(iq.size),
        p_off, fs_p, toa.ctypes.data_as(C.c_void_p), rv.ctypes.data_as(C.c_void_p))
    _check(rc, "gmr1_hip_fcch_rough_batch")
    return toa, rv


def fcch_fine_batch(iq, offset, sps=4, freq_shift=None, fcch_type="fcch"):
    iq, p_iq = _np(np.asarray(iq).reshape(-1), np.complex64)
    offset, p_off = _np(offset, np.uint64)
    n = offset.size
    fs_p = None
    if freq_shift is not None:
        fs, fs_p = _np(freq_shift, np.float32)
    toa = np.zeros(n, np.int32)
    fe = np.zeros(n, np.float32)
    rc = load().gmr1_hip_fcch_fine_batch(
        C.c_int(_fcch_id(fcch_type)), C.c_int(n), C.c_int(sps), p_iq, C.c_uint64(iq.size), p_off, fs_p,
        toa.ctypes.data_as(C.c_void_p), fe.ctypes.data_as(C.c_void_p))
    _check(rc, "gmr1_hip_fcch_fine_batch")
    return toa, fe


def fcch_snr_batch(iq, offset, sps=4, freq_shift=None, fcch_type="fcch"):
    iq, p_iq = _np(np.asarray(iq).reshape(-1), np.complex64)
    offset, p_off = _np(offset, np.uint64)
    n = offset.size
    fs_p = None
    if freq_shift is not None:
        fs, fs_p = _np(freq_shift, np.float32)
    snr = np.zeros(n, np.float32)
    rc = load().gmr1_hip_fcch_snr_batch(
        C.c_int(_fcch_id(fcch_type)), C.c_int(n), C.c_int(sps), p_iq, C.c_uint64(iq.size), p_off, fs_p,
        snr.ctypes.data_as(C.c_void_p))
    _check(rc, "gmr1_hip_fcch_snr_batch")
    return snr


def _fcch_struct(fcch_type):
    name = {"fcch": "gmr1_fcch_burst", "fcch3_lband": "gmr1_fcch3_lband_burst",
            "fcch3_sband": "gmr1_fcch3_sband_burst"}[fcch_type]
    return FcchBurst.in_dll(load(), name)


def fcch_rough(iq, sps=4, freq_shift=0.0, fcch_type="fcch"):
    """gmr1_fcch_rough(&gmr1_fcch_burst, cxvec, sps, freq_shift, &toa): the reference's own call."""
    iq = np.ascontiguousarray(iq, np.complex64)
    vec = CxVec(iq.size, iq.size, 0, iq.ctypes.data_as(C.c_void_p))
    toa = C.c_int()
    f = load().gmr1_fcch_rough
    f.restype = C.c_int
    rv = f(C.byref(_fcch_struct(fcch_type)), C.byref(vec), C.c_int(sps), C.c_float(freq_shift), C.byref(toa))
    return rv, toa.value


def fcch_fine(iq, sps=4, freq_shift=0.0, fcch_type="fcch"):
    iq = np.ascontiguousarray(iq, np.complex64)
    vec = CxVec(iq.size, iq.size, 0, iq.ctypes.data_as(C.c_void_p))
    toa, fe = C.c_int(), C.c_float()
    f = load().gmr1_fcch_fine
    f.restype = C.c_int
    rv = f(C.byref(_fcch_struct(fcch_type)), C.byref(vec), C.c_int(sps), C.c_float(freq_shift),
           C.byref(toa), C.byref(fe))
    return rv, toa.value, fe.value


def fcch_snr(iq, sps=4, freq_shift=0.0, fcch_type="fcch"):
    iq = np.ascontiguousarray(iq, np.complex64)
    vec = CxVec(iq.size, iq.size, 0, iq.ctypes.data_as(C.c_void_p))
    snr = C.c_float()
    f = load().gmr1_fcch_snr
    f.restype = C.c_int
    rv = f(C.byref(_fcch_struct(fcch_type)), C.byref(vec), C.c_int(sps), C.c_float(freq_shift), C.byref(snr))
    return rv, snr.value


def fcch_rough_batch_dev(stream, fcch_type, n, sps, length, iq, offset, freq_shift, toa, rv):
    f = load().gmr1_hip_fcch_rough_batch_dev
    f.restype = C.c_int
    vp = lambda x: C.c_void_p(x) if x else None
    rc = f(vp(stream), C.c_int(_fcch_id(fcch_type)), C.c_int(n), C.c_int(sps), C.c_int(length),
           vp(iq), vp(offset), vp(freq_shift), vp(toa), vp(rv))
    _check(rc, "gmr1_hip_fcch_rough_batch_dev")


# ---------------------------------------------------------------------------
# traffic-channel layer 1
# ---------------------------------------------------------------------------
def facch3_decode_batch(ebits, ciph=None):
    """ebits (n, 4, 104) or (n, 416) int8 -> l2 (n,10), bits_s (n,32), crc, conv."""
    ebits, p = _np(ebits, np.int8)
    ebits = ebits.reshape(-1, 416)
    n = ebits.shape[0]
    cp = None
    if ciph is not None:
        ciph, cp = _np(ciph, np.uint8)
    l2 = np.zeros((n, 10), np.uint8)
    s = np.zeros((n, 32), np.uint8)
    crc = np.zeros(n, np.int32)
    conv = np.zeros(n, np.int32)
    rc = load().gmr1_hip_facch3_decode_batch(C.c_int(n), p, cp, l2.ctypes.data_as(C.c_void_p),
                                             s.ctypes.data_as(C.c_void_p), crc.ctypes.data_as(C.c_void_p),
                                             conv.ctypes.data_as(C.c_void_p))
    _check(rc, "gmr1_hip_facch3_decode_batch")
    return l2, s, crc, conv


def tch3_decode_batch(ebits, m=0, ciph=None):
    """ebits (n, 212) int8 -> frame0 (n,10), frame1 (n,10), bits_s (n,4), conv0, conv1."""
    ebits, p = _np(ebits, np.int8)
    ebits = ebits.reshape(-1, 212)
    n = ebits.shape[0]
    cp = None
    if ciph is not None:
        ciph, cp = _np(ciph, np.uint8)
    fr = np.zeros((n, 2, 10), np.uint8)
    s = np.zeros((n, 4), np.uint8)
    conv = np.zeros((n, 2), np.int32)
    rc = load().gmr1_hip_tch3_decode_batch(C.c_int(n), C.c_int(m), p, cp, fr.ctypes.data_as(C.c_void_p),
                                           s.ctypes.data_as(C.c_void_p), conv.ctypes.data_as(C.c_void_p))
    _check(rc, "gmr1_hip_tch3_decode_batch")
    return fr[:, 0], fr[:, 1], s, conv[:, 0], conv[:, 1]


def tch3_rx_batch(iq, offset, in_len, sps=4, freq_shift=None, m=0, ciph=None, want_ebits=True):
    """NT3 speech bursts from samples to speech frames (rx_tch3's demodulate-then-decode, one call):
    dict(rv, sync_id, toa, ebits (n, 212), frame0, frame1, bits_s, conv0, conv1)."""
    iq, p_iq = _np(np.asarray(iq).reshape(-1), np.complex64)
    offset, p_off = _np(offset, np.uint64)
    n = offset.size
    fs_p = cp = None
    if freq_shift is not None:
        fs, fs_p = _np(freq_shift, np.float32)
    if ciph is not None:
        ciph, cp = _np(ciph, np.uint8)
    eb = np.zeros((n, 212), np.int8) if want_ebits else None
    sid = np.zeros(n, np.int32)
    toa = np.zeros(n, np.float32)
    rv = np.zeros(n, np.int32)
    fr = np.zeros((n, 2, 10), np.uint8)
    st = np.zeros((n, 4), np.uint8)
    conv = np.zeros((n, 2), np.int32)
    vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    rc = load().gmr1_hip_tch3_rx_batch(C.c_int(n), C.c_int(sps), C.c_int(in_len), p_iq, C.c_uint64(iq.size), p_off, fs_p,
                                       C.c_int(m), cp, vp(eb), vp(sid), vp(toa), vp(rv), vp(fr), vp(st), vp(conv))
    _check(rc, "gmr1_hip_tch3_rx_batch")
    return dict(rv=rv, sync_id=sid, toa=toa, ebits=eb, frame0=fr[:, 0], frame1=fr[:, 1], bits_s=st, conv0=conv[:, 0],
                conv1=conv[:, 1])


def facch3_decode(ebits):
    """gmr1_facch3_decode(l2, bits_s, bits_e, NULL, &conv): the reference's own call."""
    ebits = np.ascontiguousarray(ebits, np.int8).reshape(416)
    l2 = np.zeros(10, np.uint8)
    s = np.zeros(32, np.uint8)
    cv = C.c_int()
    f = load().gmr1_facch3_decode
    f.restype = C.c_int
    rv = f(l2.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), ebits.ctypes.data_as(C.c_void_p),
           None, C.byref(cv))
    return l2, s, rv, cv.value


def tch3_decode(ebits, m=0):
    ebits = np.ascontiguousarray(ebits, np.int8).reshape(212)
    f0 = np.zeros(10, np.uint8)
    f1 = np.zeros(10, np.uint8)
    s = np.zeros(4, np.uint8)
    c0, c1 = C.c_int(), C.c_int()
    f = load().gmr1_tch3_decode
    f.restype = None
    f(f0.ctypes.data_as(C.c_void_p), f1.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p),
      ebits.ctypes.data_as(C.c_void_p), None, C.c_int(m), C.byref(c0), C.byref(c1))
    return f0, f1, s, c0.value, c1.value


# ---------------------------------------------------------------------------
# burst type detection / modulation order
# ---------------------------------------------------------------------------
def detect_batch(bursts, iq, offset, in_len, sps=4, freq_shift=None, e_toa=None):
    ids = np.array([BURST_IDS.index(b) if isinstance(b, str) else int(b) for b in bursts], np.int32)
    iq, p_iq = _np(np.asarray(iq).reshape(-1), np.complex64)
    offset, p_off = _np(offset, np.uint64)
    n = offset.size
    fs_p = et_p = None
    if freq_shift is not None:
        fs, fs_p = _np(freq_shift, np.float32)
    if e_toa is not None:
        et, et_p = _np(np.broadcast_to(np.asarray(e_toa, np.float32), (n,)), np.float32)
    bt = np.zeros(n, np.int32)
    sid = np.zeros(n, np.int32)
    toa = np.zeros(n, np.float32)
    rv = np.zeros(n, np.int32)
    rc = load().gmr1_hip_detect_batch(
        C.c_int(ids.size), ids.ctypes.data_as(C.c_void_p), C.c_int(n), C.c_int(sps), C.c_int(in_len), p_iq,
        C.c_uint64(iq.size), p_off, fs_p, et_p, bt.ctypes.data_as(C.c_void_p), sid.ctypes.data_as(C.c_void_p),
        toa.ctypes.data_as(C.c_void_p), rv.ctypes.data_as(C.c_void_p))
    _check(rc, "gmr1_hip_detect_batch")
    return dict(rv=rv, bt_id=bt, sync_id=sid, toa=toa)


def mod_order_batch(iq, offset, in_len, sps=4, freq_shift=None):
    iq, p_iq = _np(np.asarray(iq).reshape(-1), np.complex64)
    offset, p_off = _np(offset, np.uint64)
    n = offset.size
    fs_p = None
    if freq_shift is not None:
        fs, fs_p = _np(freq_shift, np.float32)
    order = np.zeros(n, np.int32)
    rc = load().gmr1_hip_mod_order_batch(C.c_int(n), C.c_int(sps), C.c_int(in_len), p_iq, C.c_uint64(iq.size),
                                         p_off, fs_p, order.ctypes.data_as(C.c_void_p))
    _check(rc, "gmr1_hip_mod_order_batch")
    return order


class _RefMod(C.Structure):       # struct gmr1_pi4cxpsk_modulation (sdr/pi4cxpsk.h)
    _fields_ = [("rotation", C.c_float), ("nbits", C.c_int), ("syms", C.c_void_p), ("bits", C.c_void_p)]


class _RefSync(C.Structure):      # struct gmr1_pi4cxpsk_sync
    _fields_ = [("pos", C.c_int), ("len", C.c_int), ("syms", C.c_uint8 * 32), ("_ref", C.c_void_p)]


class _RefData(C.Structure):      # struct gmr1_pi4cxpsk_data
    _fields_ = [("pos", C.c_int), ("len", C.c_int)]


class _RefBurst(C.Structure):     # struct gmr1_pi4cxpsk_burst
    _fields_ = [("mod", C.POINTER(_RefMod)), ("guard_pre", C.c_int), ("guard_post", C.c_int),
                ("len", C.c_int), ("ebits", C.c_int), ("sync", C.POINTER(_RefSync) * 4),
                ("data", C.POINTER(_RefData))]


class CallerBurst:
    """A caller-defined `struct gmr1_pi4cxpsk_burst` (the reference lets applications describe their own burst formats
    and pass them to gmr1_pi4cxpsk_demod / _detect): a deep copy of an exported one in memory of our own, optionally
    edited.  `.address` is what goes where `&gmr1_xyz_burst` would."""

    def __init__(self, like: str):
        src = _RefBurst.in_dll(load(), f"gmr1_{like}_burst")
        self.burst = _RefBurst()
        self.burst.mod = src.mod                       # the modulation objects are the library's (as in the reference)
        self.burst.guard_pre, self.burst.guard_post = src.guard_pre, src.guard_post
        self.burst.len, self.burst.ebits = src.len, src.ebits
        self._keep = []
        for k in range(4):
            if not src.sync[k]:
                break
            n = 0
            while src.sync[k][n].pos >= 0:
                n += 1
            arr = (_RefSync * (n + 1))()
            for i in range(n):
                arr[i].pos, arr[i].len = src.sync[k][i].pos, src.sync[k][i].len
                C.memmove(arr[i].syms, src.sync[k][i].syms, 32)
            arr[n].pos = -1
            self._keep.append(arr)
            self.burst.sync[k] = C.cast(arr, C.POINTER(_RefSync))
        n = 0
        while src.data[n].pos >= 0:
            n += 1
        d = (_RefData * (n + 1))()
        for i in range(n):
            d[i].pos, d[i].len = src.data[i].pos, src.data[i].len
        d[n].pos = -1
        self._keep.append(d)
        self.burst.data = C.cast(d, C.POINTER(_RefData))

    @property
    def address(self):
        return C.addressof(self.burst)


def pi4cxpsk_detect(burst_names, e_toa, iq, sps=4, freq_shift=0.0):
    """gmr1_pi4cxpsk_detect({&gmr1_a_burst, &gmr1_b_burst, NULL}, e_toa, cxvec, ...) as C callers use it; an entry
    may also be a CallerBurst (a description of the caller's own)."""
    L = load()
    arr = (C.c_void_p * (len(burst_names) + 1))()
    for i, nm in enumerate(burst_names):
        arr[i] = nm.address if isinstance(nm, CallerBurst) else C.addressof(C.c_void_p.in_dll(L, f"gmr1_{nm}_burst"))
    arr[len(burst_names)] = None
    iq = np.ascontiguousarray(iq, np.complex64)
    vec = CxVec(iq.size, iq.size, 0, iq.ctypes.data_as(C.c_void_p))
    bt, sid, toa = C.c_int(-1), C.c_int(-1), C.c_float()
    f = L.gmr1_pi4cxpsk_detect
    f.restype = C.c_int
    rv = f(arr, C.c_float(e_toa), C.byref(vec), C.c_int(sps), C.c_float(freq_shift),
           C.byref(bt), C.byref(sid), C.byref(toa))
    return dict(rv=rv, bt_id=bt.value, sync_id=sid.value, toa=toa.value)


def pi4cxpsk_mod_order(iq, sps=4, freq_shift=0.0):
    iq = np.ascontiguousarray(iq, np.complex64)
    vec = CxVec(iq.size, iq.size, 0, iq.ctypes.data_as(C.c_void_p))
    f = load().gmr1_pi4cxpsk_mod_order
    f.restype = C.c_int
    return f(C.byref(vec), C.c_int(sps), C.c_float(freq_shift))


def fcch_rough_multi_batch(iq, offset, length, sps=4, freq_shift=None, N=16, fcch_type="fcch"):
    iq, p_iq = _np(np.asarray(iq).reshape(-1), np.complex64)
    offset, p_off = _np(offset, np.uint64)
    n = offset.size
    fs_p = None
    if freq_shift is not None:
        fs, fs_p = _np(freq_shift, np.float32)
    toa = np.zeros((n, N), np.int32)
    cnt = np.zeros(n, np.int32)
    rc = load().gmr1_hip_fcch_rough_multi_batch(
        C.c_int(_fcch_id(fcch_type)), C.c_int(n), C.c_int(sps), C.c_int(length), p_iq, C.c_uint64(iq.size),
        p_off, fs_p, toa.ctypes.data_as(C.c_void_p), C.c_int(N), cnt.ctypes.data_as(C.c_void_p))
    _check(rc, "gmr1_hip_fcch_rough_multi_batch")
    return cnt, toa


def fcch_rough_multi(iq, sps=4, freq_shift=0.0, N=16, fcch_type="fcch"):
    iq = np.ascontiguousarray(iq, np.complex64)
    vec = CxVec(iq.size, iq.size, 0, iq.ctypes.data_as(C.c_void_p))
    toa = np.zeros(N, np.int32)
    f = load().gmr1_fcch_rough_multi
    f.restype = C.c_int
    rv = f(C.byref(_fcch_struct(fcch_type)), C.byref(vec), C.c_int(sps), C.c_float(freq_shift),
           toa.ctypes.data_as(C.c_void_p), C.c_int(N))
    return rv, toa[:max(rv, 0)].copy()


# ---------------------------------------------------------------------------
# gmr1_rx receive loop over many BCCH carriers (reference src/gmr1_rx.c:605-895)
# ---------------------------------------------------------------------------
RX_RECORD = np.dtype([("arfcn", "<u2"), ("chain", "u1"), ("type", "u1"), ("fn", "<u4"),
                      ("tn", "u1"), ("crc", "u1"), ("len", "u1"), ("pad", "u1"),
                      ("conv", "<i4"), ("l2", "u1", (24,))])
assert RX_RECORD.itemsize == 40


def _rx_run_call(fname, head_args, n, offset, length, arfcn, max_records, out=None):
    offset, p_off = _np(offset, np.uint64)
    length, p_len = _np(length, np.uint64)
    p_arfcn = None
    if arfcn is not None:
        arfcn, p_arfcn = _np(arfcn, np.uint16)
    reuse = out is not None
    if reuse:
        if out.dtype != RX_RECORD or not out.flags.c_contiguous:
            raise ValueError("out must be a contiguous RX_RECORD array")
        max_records = out.size
    else:
        out = np.empty(max(max_records, 1), RX_RECORD)    # only the records written are handed back
    n_rec = C.c_int(0)
    status = np.zeros(max(n, 1), np.int32)
    chains = np.zeros(max(n, 1), np.int32)
    f = getattr(load(), fname)
    f.restype = C.c_int
    rc = f(*head_args, p_off, p_len, p_arfcn, out.ctypes.data_as(C.c_void_p), C.c_int(max_records),
           C.byref(n_rec), status.ctypes.data_as(C.c_void_p), chains.ctypes.data_as(C.c_void_p))
    _check(rc, fname)
    got = out[:min(n_rec.value, max_records)]
    return (got if reuse else got.copy()), status[:n], chains[:n], n_rec.value


def rx_run(iq, offset, length, sps=4, arfcn=None, max_records=1 << 16):
    """gmr1_hip_rx_run: iq is one host complex64 buffer holding every carrier; carrier i is
    iq[offset[i] : offset[i] + length[i]].  Returns (records RX_RECORD[], status[n], n_chains[n], n_found)."""
    iq, p_iq = _np(np.asarray(iq).reshape(-1), np.complex64)
    n = len(offset)
    head = (C.c_int(n), C.c_int(sps), p_iq, C.c_uint64(iq.size))
    return _rx_run_call("gmr1_hip_rx_run", head, n, offset, length, arfcn, max_records)


def rx_run_dev(stream, iq_ptr, offset, length, sps=4, arfcn=None, max_records=1 << 16, out=None):
    """gmr1_hip_rx_run_dev: as rx_run with the capture already in HBM (iq_ptr = device address).
    out: optional preallocated RX_RECORD array the records are written into (a view of it is returned)."""
    n = len(offset)
    head = (C.c_void_p(stream) if stream else None, C.c_int(n), C.c_int(sps), C.c_void_p(iq_ptr))
    return _rx_run_call("gmr1_hip_rx_run_dev", head, n, offset, length, arfcn, max_records, out)


def rx_run_dev_prepared(stream, iq_ptr, offset, length, out, sps=4, arfcn=None):
    """gmr1_hip_rx_run_dev with every argument marshalled ONCE: returns call() -> (records view, status, n_chains, n_found).
    What a C host pays per call is the call itself; a Python caller that repeats the same call (bench.py) should not time
    its own argument conversion either."""
    n = len(offset)
    offset, p_off = _np(offset, np.uint64)
    length, p_len = _np(length, np.uint64)
    p_arfcn = None
    if arfcn is not None:
        arfcn, p_arfcn = _np(arfcn, np.uint16)
    if out.dtype != RX_RECORD or not out.flags.c_contiguous:
        raise ValueError("out must be a contiguous RX_RECORD array")
    n_rec = C.c_int(0)
    status = np.zeros(max(n, 1), np.int32)
    chains = np.zeros(max(n, 1), np.int32)
    f = load().gmr1_hip_rx_run_dev
    f.restype = C.c_int
    args = (C.c_void_p(stream) if stream else None, C.c_int(n), C.c_int(sps), C.c_void_p(iq_ptr), p_off, p_len, p_arfcn,
            out.ctypes.data_as(C.c_void_p), C.c_int(out.size), C.byref(n_rec), status.ctypes.data_as(C.c_void_p),
            chains.ctypes.data_as(C.c_void_p))
    keep = (offset, length, arfcn)                      # the arrays the pointers point into

    def call():
        rc = f(*args)
        if rc:
            _check(rc, "gmr1_hip_rx_run_dev")
        return out[:min(n_rec.value, out.size)], status[:n], chains[:n], n_rec.value
    call.keep = keep
    return call


def rx_run_last_timing():
    """Phases of this thread's last rx_run* call in ms: acquisition, frame loop (kernels), records hand-back, host work around
    the loop, traffic-channel passes."""
    us = (C.c_double * 5)()
    _check(load().gmr1_hip_rx_run_last_timing(us), "gmr1_hip_rx_run_last_timing")
    return dict(zip(("acquisition_ms", "chain_ms", "handback_ms", "host_ms", "traffic_passes_ms"), [v / 1e3 for v in us]))


def rx_run_dev_raw(stream, iq_ptr, offset, length, out_ptr, max_records, sps=4, arfcn=None):
    """gmr1_hip_rx_run_dev with the record buffer given as an address -- device memory or pinned host memory, which the
    library copies into directly.  Returns (n_found, status[n], n_chains[n])."""
    n = len(offset)
    offset, p_off = _np(offset, np.uint64)
    length, p_len = _np(length, np.uint64)
    p_arfcn = None
    if arfcn is not None:
        arfcn, p_arfcn = _np(arfcn, np.uint16)
    n_rec = C.c_int(0)
    status = np.zeros(max(n, 1), np.int32)
    chains = np.zeros(max(n, 1), np.int32)
    f = load().gmr1_hip_rx_run_dev
    f.restype = C.c_int
    rc = f(C.c_void_p(stream) if stream else None, C.c_int(n), C.c_int(sps), C.c_void_p(iq_ptr), p_off, p_len, p_arfcn,
           C.c_void_p(out_ptr), C.c_int(max_records), C.byref(n_rec), status.ctypes.data_as(C.c_void_p),
           chains.ctypes.data_as(C.c_void_p))
    _check(rc, "gmr1_hip_rx_run_dev")
    return n_rec.value, status[:n], chains[:n]


def gsmtap_pack(record, with_arfcn=False) -> bytes:
    """gmr1_hip_gsmtap_pack: the GSMTAP packet the reference would send for one RX_RECORD (src/gsmtap.c:43-71)."""
    rec = np.ascontiguousarray(np.asarray(record, RX_RECORD).reshape(1))
    buf = (C.c_uint8 * 64)()
    f = load().gmr1_hip_gsmtap_pack
    f.restype = C.c_int
    n = f(rec.ctypes.data_as(C.c_void_p), C.c_int(1 if with_arfcn else 0), buf, C.c_int(64))
    if n < 0:
        _check(n, "gmr1_hip_gsmtap_pack")
    return bytes(buf[:n])


# ---------------------------------------------------------------------------
# TCH3 follow-up pieces: DKAB demodulator, A5 keystream
# ---------------------------------------------------------------------------
def dkab_demod_batch(iq, offset, in_len, p, sps=4, freq_shift=None):
    """gmr1_hip_dkab_demod_batch -> (rv[n], ebits[n, 8], toa[n])"""
    iq, p_iq = _np(np.asarray(iq).reshape(-1), np.complex64)
    offset, p_off = _np(offset, np.uint64)
    n = offset.size
    pp, p_p = _np(np.broadcast_to(np.asarray(p, np.int32), (n,)), np.int32)
    fs_p = None
    if freq_shift is not None:
        fs, fs_p = _np(np.broadcast_to(np.asarray(freq_shift, np.float32), (n,)), np.float32)
    eb = np.zeros((n, 8), np.int8)
    toa = np.zeros(n, np.float32)
    rv = np.zeros(n, np.int32)
    f = load().gmr1_hip_dkab_demod_batch
    f.restype = C.c_int
    rc = f(C.c_int(n), C.c_int(sps), C.c_int(in_len), p_iq, C.c_uint64(iq.size), p_off, fs_p, p_p,
           eb.ctypes.data_as(C.c_void_p), toa.ctypes.data_as(C.c_void_p), rv.ctypes.data_as(C.c_void_p))
    _check(rc, "gmr1_hip_dkab_demod_batch")
    return rv, eb, toa


def dkab_demod(iq, sps=4, freq_shift=0.0, p=0):
    """gmr1_dkab_demod, the reference's own call -> (rv, ebits[8], toa)"""
    iq, p_iq = _np(np.asarray(iq).reshape(-1), np.complex64)
    v = CxVec(iq.size, iq.size, 0, iq.ctypes.data_as(C.c_void_p))
    eb = np.zeros(8, np.int8)
    toa = C.c_float(0.0)
    f = load().gmr1_dkab_demod
    f.restype = C.c_int
    rv = f(C.byref(v), C.c_int(sps), C.c_float(freq_shift), C.c_int(p), eb.ctypes.data_as(C.c_void_p), C.byref(toa))
    if rv < 0:
        _check(rv, "gmr1_dkab_demod")
    return rv, eb, toa.value


def a5_batch(alg, keys, fn, nbits, want_ul=False):
    """gmr1_hip_a5_batch: keys (n, 8) or (8,), fn (n,) -> dl (n, nbits) [, ul]"""
    fn, p_fn = _np(np.atleast_1d(fn), np.uint32)
    n = fn.size
    keys, p_k = _np(np.broadcast_to(np.asarray(keys, np.uint8).reshape(-1, 8), (n, 8)), np.uint8)
    dl = np.zeros((n, nbits), np.uint8)
    ul = np.zeros((n, nbits), np.uint8) if want_ul else None
    f = load().gmr1_hip_a5_batch
    f.restype = C.c_int
    rc = f(C.c_int(n), C.c_int(alg), C.c_int(nbits), p_k, p_fn, dl.ctypes.data_as(C.c_void_p),
           ul.ctypes.data_as(C.c_void_p) if want_ul else None)
    _check(rc, "gmr1_hip_a5_batch")
    return (dl, ul) if want_ul else dl


def a5(n, key, fn, nbits):
    """gmr1_a5, the reference's own call -> (dl, ul)"""
    key, p_k = _np(key, np.uint8)
    dl = np.full(nbits, 0xEE, np.uint8)
    ul = np.full(nbits, 0xEE, np.uint8)
    f = load().gmr1_a5
    f.restype = None
    f(C.c_int(n), p_k, C.c_uint32(int(fn)), C.c_int(nbits), dl.ctypes.data_as(C.c_void_p), ul.ctypes.data_as(C.c_void_p))
    return dl, ul


def rx_run_tch(iq, tch, offset, length, sps=4, arfcn=None, kc=None, max_records=1 << 16):
    """gmr1_hip_rx_run_tch: rx_run with the traffic carriers `tch` (same layout as iq) and keys kc (n, 8)."""
    iq, p_iq = _np(np.asarray(iq).reshape(-1), np.complex64)
    p_tch = None
    if tch is not None:
        tch, p_tch = _np(np.asarray(tch).reshape(-1), np.complex64)
        assert tch.size == iq.size
    n = len(offset)
    p_kc = None
    if kc is not None:
        kc, p_kc = _np(np.broadcast_to(np.asarray(kc, np.uint8).reshape(-1, 8), (n, 8)), np.uint8)
    offset, p_off = _np(offset, np.uint64)
    length, p_len = _np(length, np.uint64)
    p_arfcn = None
    if arfcn is not None:
        arfcn, p_arfcn = _np(arfcn, np.uint16)
    out = np.zeros(max(max_records, 1), RX_RECORD)
    n_rec = C.c_int(0)
    status = np.zeros(max(n, 1), np.int32)
    chains = np.zeros(max(n, 1), np.int32)
    f = load().gmr1_hip_rx_run_tch
    f.restype = C.c_int
    rc = f(C.c_int(n), C.c_int(sps), p_iq, p_tch, C.c_uint64(iq.size), p_off, p_len, p_arfcn, p_kc,
           out.ctypes.data_as(C.c_void_p), C.c_int(max_records), C.byref(n_rec),
           status.ctypes.data_as(C.c_void_p), chains.ctypes.data_as(C.c_void_p))
    _check(rc, "gmr1_hip_rx_run_tch")
    return out[:min(n_rec.value, max_records)].copy(), status[:n], chains[:n], n_rec.value


# ---------------------------------------------------------------------------
# wideband -> per-ARFCN channelizer (reference utils/gmr1_rx_sdr.py:391-602)
# ---------------------------------------------------------------------------
def channelize_plan(samp_rate, sps, n_in):
    """-> (n_chans, samples per 2x oversampled channel stream, output samples per channel)"""
    nch, nm, no = C.c_int32(), C.c_uint64(), C.c_uint64()
    f = load().gmr1_hip_channelize_plan
    f.restype = C.c_int
    _check(f(C.c_double(samp_rate), C.c_int(sps), C.c_uint64(n_in), C.byref(nch), C.byref(nm), C.byref(no)),
           "gmr1_hip_channelize_plan")
    return nch.value, nm.value, no.value


def channelize(wide, samp_rate, channels, sps=4, rotation=0.0):
    """gmr1_hip_channelize: host wideband complex64 -> (len(channels), n_out) complex64"""
    wide, p_w = _np(np.asarray(wide).reshape(-1), np.complex64)
    ch, p_ch = _np(channels, np.int32)
    _, _, n_out = channelize_plan(samp_rate, sps, wide.size)
    out = np.zeros((ch.size, n_out), np.complex64)
    no = C.c_uint64()
    f = load().gmr1_hip_channelize
    f.restype = C.c_int
    rc = f(C.c_double(samp_rate), C.c_int(sps), p_w, C.c_uint64(wide.size), C.c_float(rotation), C.c_int(ch.size), p_ch,
           out.ctypes.data_as(C.c_void_p), C.c_uint64(n_out), C.byref(no))
    _check(rc, "gmr1_hip_channelize")
    return out


def ddc_plan(samp_rate, sps, n_in):
    """gmr1_hip_ddc_plan -> (decim1, decim2, resamp, n_out) of the recorder script's direct mode."""
    d1, d2, rs, n_out = C.c_int32(), C.c_int32(), C.c_double(), C.c_uint64()
    _check(load().gmr1_hip_ddc_plan(C.c_double(samp_rate), C.c_int(sps), C.c_uint64(n_in), C.byref(d1), C.byref(d2),
                                    C.byref(rs), C.byref(n_out)), "gmr1_hip_ddc_plan")
    return d1.value, d2.value, rs.value, n_out.value


def ddc(wide, samp_rate, freqs_hz, sps=4):
    """gmr1_hip_ddc: wide (complex64, host) -> array (len(freqs_hz), n_out) complex64."""
    wide, p_w = _np(np.asarray(wide).reshape(-1), np.complex64)
    freqs, p_f = _np(freqs_hz, np.float64)
    _, _, _, n_out = ddc_plan(samp_rate, sps, wide.size)
    out = np.zeros((freqs.size, max(n_out, 1)), np.complex64)
    got = C.c_uint64()
    _check(load().gmr1_hip_ddc(C.c_double(samp_rate), C.c_int(sps), p_w, C.c_uint64(wide.size), C.c_int(freqs.size), p_f,
                               out.ctypes.data_as(C.c_void_p), C.c_uint64(out.shape[1]), C.byref(got)), "gmr1_hip_ddc")
    return out[:, :got.value]


def ddc_dev(stream, wide_ptr, n_in, samp_rate, freqs_hz, out_ptr, out_stride, sps=4):
    freqs, p_f = _np(freqs_hz, np.float64)
    got = C.c_uint64()
    _check(load().gmr1_hip_ddc_dev(C.c_void_p(stream) if stream else None, C.c_double(samp_rate), C.c_int(sps),
                                   C.c_void_p(wide_ptr), C.c_uint64(n_in), C.c_int(freqs.size), p_f, C.c_void_p(out_ptr),
                                   C.c_uint64(out_stride), C.byref(got)), "gmr1_hip_ddc_dev")
    return got.value


def channelize_dev(stream, wide_ptr, n_in, samp_rate, channels, out_ptr, out_stride, sps=4, rotation=0.0):
    ch, p_ch = _np(channels, np.int32)
    no = C.c_uint64()
    f = load().gmr1_hip_channelize_dev
    f.restype = C.c_int
    rc = f(C.c_void_p(stream) if stream else None, C.c_double(samp_rate), C.c_int(sps), C.c_void_p(wide_ptr),
           C.c_uint64(n_in), C.c_float(rotation), C.c_int(ch.size), p_ch, C.c_void_p(out_ptr), C.c_uint64(out_stride),
           C.byref(no))
    _check(rc, "gmr1_hip_channelize_dev")
    return no.value


def channelize_planar_dev(stream, wide_ptr, n_in, samp_rate, channels, out_ptr, out_stride, plane_stride, sps=4, rotation=0.0):
    """gmr1_hip_channelize_planar_dev: the streams written polyphase-planar (sample m of stream i at flat index
    g = i * out_stride + m -> out[(g % sps) * plane_stride + g // sps])."""
    ch, p_ch = _np(channels, np.int32)
    no = C.c_uint64()
    f = load().gmr1_hip_channelize_planar_dev
    f.restype = C.c_int
    rc = f(C.c_void_p(stream) if stream else None, C.c_double(samp_rate), C.c_int(sps), C.c_void_p(wide_ptr),
           C.c_uint64(n_in), C.c_float(rotation), C.c_int(ch.size), p_ch, C.c_void_p(out_ptr), C.c_uint64(out_stride),
           C.c_uint64(plane_stride), C.byref(no))
    _check(rc, "gmr1_hip_channelize_planar_dev")
    return no.value


# ---------------------------------------------------------------------------
# NT9 bursts: FACCH9, TCH9
# ---------------------------------------------------------------------------
TCH9_BYTES = (18, 30, 60)


def facch9_decode_batch(ebits, ciph=None):
    """(n, 662) soft bits -> (l2 (n, 38), sacch (n, 10), status (n, 4), crc (n,), conv (n,))"""
    eb, p_eb = _np(ebits, np.int8)
    n = eb.shape[0]
    p_c = None
    if ciph is not None:
        c, p_c = _np(ciph, np.uint8)
    l2 = np.zeros((n, 38), np.uint8)
    sa = np.zeros((n, 10), np.int8)
    stt = np.zeros((n, 4), np.int8)
    crc = np.zeros(n, np.int32)
    conv = np.zeros(n, np.int32)
    f = load().gmr1_hip_facch9_decode_batch
    f.restype = C.c_int
    rc = f(C.c_int(n), p_eb, p_c, l2.ctypes.data_as(C.c_void_p), sa.ctypes.data_as(C.c_void_p),
           stt.ctypes.data_as(C.c_void_p), crc.ctypes.data_as(C.c_void_p), conv.ctypes.data_as(C.c_void_p))
    _check(rc, "gmr1_hip_facch9_decode_batch")
    return l2, sa, stt, crc, conv


def facch9_decode(ebits, ciph=None):
    """gmr1_facch9_decode, the reference's own call -> (l2, sacch, status, crc, conv)"""
    eb, p_eb = _np(ebits, np.int8)
    p_c = None
    if ciph is not None:
        c, p_c = _np(ciph, np.uint8)
    l2 = np.zeros(38, np.uint8)
    sa = np.zeros(10, np.int8)
    stt = np.zeros(4, np.int8)
    conv = C.c_int(0)
    f = load().gmr1_facch9_decode
    f.restype = C.c_int
    crc = f(l2.ctypes.data_as(C.c_void_p), sa.ctypes.data_as(C.c_void_p), stt.ctypes.data_as(C.c_void_p), p_eb, p_c,
            C.byref(conv))
    if crc < 0:
        _check(crc, "gmr1_facch9_decode")
    return l2, sa, stt, crc, conv.value


class Interleaver(C.Structure):
    """struct gmr1_interleaver (include/osmocom/gmr1/l1/interleave.h)"""
    _fields_ = [("N", C.c_int), ("K", C.c_int), ("n", C.c_int), ("bits_cpp", C.c_void_p)]


class Tch9Channel:
    """One TCH9 channel decoded burst by burst with the reference's own stateful calls:
    gmr1_interleaver_init(&il, 3, 648) once, then gmr1_tch9_decode(...) per burst (gmr1_rx.c:273, :333)."""

    def __init__(self, mode: int, N: int = 3, K: int = 648):
        self.mode = mode
        self.il = Interleaver()
        f = load().gmr1_interleaver_init
        f.restype = C.c_int
        _check(f(C.byref(self.il), C.c_int(N), C.c_int(K)), "gmr1_interleaver_init")

    def decode(self, ebits, ciph=None):
        """-> (l2, sacch (10,), status (4,), conv)"""
        eb, p_eb = _np(ebits, np.int8)
        if eb.size != 662:
            raise ValueError("tch9: a burst has 662 soft bits")
        p_c = None
        if ciph is not None:
            c, p_c = _np(ciph, np.uint8)
        l2 = np.zeros((18, 30, 60)[self.mode], np.uint8)
        sa = np.zeros(10, np.int8)
        stt = np.zeros(4, np.int8)
        conv = C.c_int(0)
        f = load().gmr1_tch9_decode
        f.restype = None
        f(l2.ctypes.data_as(C.c_void_p), sa.ctypes.data_as(C.c_void_p), stt.ctypes.data_as(C.c_void_p), p_eb,
          C.c_int(self.mode), p_c, C.byref(self.il), C.byref(conv))
        return l2, sa, stt, conv.value

    def close(self):
        if self.il.bits_cpp:
            load().gmr1_interleaver_fini(C.byref(self.il))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def xch_dc12_decode_batch(ebits):
    """(n, 432) soft bits of DC12 bursts -> (l2 (n, 24), crc (n,), conv (n,))"""
    eb, p_eb = _np(ebits, np.int8)
    n = eb.shape[0]
    if eb.ndim != 2 or eb.shape[1] != 432:
        raise ValueError("xch_dc12: ebits must be (n, 432)")
    l2 = np.zeros((n, 24), np.uint8)
    crc = np.zeros(n, np.int32)
    conv = np.zeros(n, np.int32)
    f = load().gmr1_hip_xch_dc12_decode_batch
    f.restype = C.c_int
    rc = f(C.c_int(n), p_eb, l2.ctypes.data_as(C.c_void_p), crc.ctypes.data_as(C.c_void_p),
           conv.ctypes.data_as(C.c_void_p))
    _check(rc, "gmr1_hip_xch_dc12_decode_batch")
    return l2, crc, conv


def xch_dc12_decode(ebits):
    """gmr1_xch_dc12_decode, the reference's own call -> (l2, crc, conv)"""
    eb, p_eb = _np(ebits, np.int8)
    l2 = np.zeros(24, np.uint8)
    conv = C.c_int(0)
    f = load().gmr1_xch_dc12_decode
    f.restype = C.c_int
    crc = f(l2.ctypes.data_as(C.c_void_p), p_eb, C.byref(conv))
    if crc < 0:
        _check(crc, "gmr1_xch_dc12_decode")
    return l2, crc, conv.value


def rach_decode_batch(ebits, sb_mask):
    """(n, 494) soft bits of RACH bursts, sb_mask scalar or (n,) -> (rach (n, 18), rv (n,), conv (n,), crc (n, 2))"""
    eb, p_eb = _np(ebits, np.int8)
    n = eb.shape[0]
    if eb.ndim != 2 or eb.shape[1] != 494:
        raise ValueError("rach: ebits must be (n, 494)")
    m = np.ascontiguousarray(np.broadcast_to(np.asarray(sb_mask, np.uint8), (n,)))
    rach = np.zeros((n, 18), np.uint8)
    rv = np.zeros(n, np.int32)
    conv = np.zeros(n, np.int32)
    crc = np.zeros((n, 2), np.int32)
    f = load().gmr1_hip_rach_decode_batch
    f.restype = C.c_int
    rc = f(C.c_int(n), p_eb, m.ctypes.data_as(C.c_void_p), rach.ctypes.data_as(C.c_void_p),
           rv.ctypes.data_as(C.c_void_p), conv.ctypes.data_as(C.c_void_p), crc.ctypes.data_as(C.c_void_p))
    _check(rc, "gmr1_hip_rach_decode_batch")
    return rach, rv, conv, crc


def rach_decode(ebits, sb_mask):
    """gmr1_rach_decode, the reference's own call -> (rach, rv, conv, (crc8, crc12))"""
    eb, p_eb = _np(ebits, np.int8)
    rach = np.zeros(18, np.uint8)
    conv = C.c_int(0)
    crc = (C.c_int * 2)()
    f = load().gmr1_rach_decode
    f.restype = C.c_int
    rv = f(rach.ctypes.data_as(C.c_void_p), p_eb, C.c_uint8(int(sb_mask)), C.byref(conv), crc)
    if rv < 0:
        _check(rv, "gmr1_rach_decode")
    return rach, rv, conv.value, (crc[0], crc[1])


def tch9_decode_batch(ebits, mode, seq_len, ciph=None):
    """(n_chan * seq_len, 662) soft bits, channel after channel -> (l2 (n, bytes), sacch, status, conv)"""
    eb, p_eb = _np(ebits, np.int8)
    n = eb.shape[0]
    assert n % seq_len == 0
    p_c = None
    if ciph is not None:
        c, p_c = _np(ciph, np.uint8)
    l2 = np.zeros((n, TCH9_BYTES[mode]), np.uint8)
    sa = np.zeros((n, 10), np.int8)
    stt = np.zeros((n, 4), np.int8)
    conv = np.zeros(n, np.int32)
    f = load().gmr1_hip_tch9_decode_batch
    f.restype = C.c_int
    rc = f(C.c_int(n // seq_len), C.c_int(seq_len), C.c_int(mode), p_eb, p_c, l2.ctypes.data_as(C.c_void_p),
           sa.ctypes.data_as(C.c_void_p), stt.ctypes.data_as(C.c_void_p), conv.ctypes.data_as(C.c_void_p))
    _check(rc, "gmr1_hip_tch9_decode_batch")
    return l2, sa, stt, conv


RX_BIG_RECORD = np.dtype([("arfcn", "<u2"), ("chain", "u1"), ("type", "u1"), ("fn", "<u4"),
                          ("tn", "u1"), ("crc", "u1"), ("len", "u1"), ("pad", "u1"),
                          ("conv", "<i4"), ("l2", "u1", (64,))])
assert RX_BIG_RECORD.itemsize == 80


def rx_run_full(iq, tch, csd, offset, length, sps=4, arfcn=None, kc=None, max_records=1 << 16, max_big=1 << 14):
    """gmr1_hip_rx_run_full: rx_run_tch plus the CSD carriers -> (records, big records, status, n_chains)"""
    iq, p_iq = _np(np.asarray(iq).reshape(-1), np.complex64)
    keep = []

    def opt(x):
        if x is None:
            return None
        x, p = _np(np.asarray(x).reshape(-1), np.complex64)
        assert x.size == iq.size
        keep.append(x)
        return p
    p_tch, p_csd = opt(tch), opt(csd)
    n = len(offset)
    p_kc = None
    if kc is not None:
        kc, p_kc = _np(np.broadcast_to(np.asarray(kc, np.uint8).reshape(-1, 8), (n, 8)), np.uint8)
    offset, p_off = _np(offset, np.uint64)
    length, p_len = _np(length, np.uint64)
    p_arfcn = None
    if arfcn is not None:
        arfcn, p_arfcn = _np(arfcn, np.uint16)
    out = np.zeros(max(max_records, 1), RX_RECORD)
    big = np.zeros(max(max_big, 1), RX_BIG_RECORD)
    n_rec, n_big = C.c_int(0), C.c_int(0)
    status = np.zeros(max(n, 1), np.int32)
    chains = np.zeros(max(n, 1), np.int32)
    f = load().gmr1_hip_rx_run_full
    f.restype = C.c_int
    rc = f(C.c_int(n), C.c_int(sps), p_iq, p_tch, p_csd, C.c_uint64(iq.size), p_off, p_len, p_arfcn, p_kc,
           out.ctypes.data_as(C.c_void_p), C.c_int(max_records), C.byref(n_rec),
           big.ctypes.data_as(C.c_void_p), C.c_int(max_big), C.byref(n_big),
           status.ctypes.data_as(C.c_void_p), chains.ctypes.data_as(C.c_void_p))
    _check(rc, "gmr1_hip_rx_run_full")
    return (out[:min(n_rec.value, max_records)].copy(), big[:min(n_big.value, max_big)].copy(), status[:n], chains[:n])


def gsmtap_pack_big(record, with_arfcn=False) -> bytes:
    """gmr1_hip_gsmtap_pack_big: the GSMTAP packet of one RX_BIG_RECORD (FACCH9 / TCH9 payloads)."""
    rec = np.ascontiguousarray(np.asarray(record, RX_BIG_RECORD).reshape(1))
    buf = (C.c_uint8 * 96)()
    f = load().gmr1_hip_gsmtap_pack_big
    f.restype = C.c_int
    n = f(rec.ctypes.data_as(C.c_void_p), C.c_int(1 if with_arfcn else 0), buf, C.c_int(96))
    if n < 0:
        _check(n, "gmr1_hip_gsmtap_pack_big")
    return bytes(buf[:n])


# ---- transmit direction: channel encoders and modulator (csrc/capi_tx.cpp, tx_kernels.hip) ----------------
ENC_CHAINS = ("bcch", "ccch", "facch3", "tch3_m0", "tch3_m1", "facch9", "tch9_2k4", "tch9_4k8", "tch9_9k6", "rach",
              "xch_dc12")


def encoder_plan(chain):
    """The position map of one encoder chain (struct EncPlan of csrc/gmr1_dev.h) as a dict of numpy arrays.
    Host-only: works without a GPU."""
    cid = ENC_CHAINS.index(chain) if isinstance(chain, str) else int(chain)
    size = load().gmr1_hip_encoder_plan(C.c_int(cid), None, C.c_int(0))
    if size < 0:
        _check(size, "gmr1_hip_encoder_plan")
    buf = np.zeros(size, np.uint8)
    rc = load().gmr1_hip_encoder_plan(C.c_int(cid), buf.ctypes.data_as(C.c_void_p), C.c_int(size))
    if rc < 0:
        _check(rc, "gmr1_hip_encoder_plan")
    head = buf[:32].view(np.int32)
    o = 32
    poly = buf[o:o + 32].view(np.uint32); o += 32
    n_tab = 64 * 8
    crc_tab = buf[o:o + 2 * n_tab].view(np.uint16); o += 2 * n_tab
    crc_tab2 = buf[o:o + 2 * n_tab].view(np.uint16); o += 2 * n_tab
    ext_src = buf[o:o + 2 * 512].view(np.uint16); o += 2 * 512
    out = buf[o:o + 4 * 672].view(np.uint32); o += 4 * 672
    assert o == size, (o, size)
    names = ("n_in0", "n_in1", "n_ext", "n_out", "n_aux0", "n_aux1", "n_ciph", "depth")
    d = {k: int(v) for k, v in zip(names, head)}
    d.update(poly=poly.copy(), crc_tab=crc_tab.copy(), crc_tab2=crc_tab2.copy(), ext_src=ext_src[:d["n_ext"]].copy(),
             out=out[:d["n_out"]].copy())
    return d


def _opt(a, dtype=np.uint8):
    if a is None:
        return None, None
    return _np(a, dtype)


def _encode_batch(fname, n_out, n, head, arrays):
    """arrays: list of (array or None); -> (n, n_out) uint8"""
    keep, ptrs = [], []
    for a in arrays:
        k, p = _opt(a)
        keep.append(k)
        ptrs.append(p)
    e = np.zeros((n, n_out), np.uint8)
    _check(getattr(load(), fname)(*head, *ptrs, e.ctypes.data_as(C.c_void_p)), fname)
    return e


def bcch_encode_batch(l2):
    l2 = np.ascontiguousarray(l2, np.uint8).reshape(-1, 24)
    return _encode_batch("gmr1_hip_bcch_encode_batch", 424, l2.shape[0], [C.c_int(l2.shape[0])], [l2])


def ccch_encode_batch(l2):
    l2 = np.ascontiguousarray(l2, np.uint8).reshape(-1, 24)
    return _encode_batch("gmr1_hip_ccch_encode_batch", 432, l2.shape[0], [C.c_int(l2.shape[0])], [l2])


def xch_dc12_encode_batch(l2):
    l2 = np.ascontiguousarray(l2, np.uint8).reshape(-1, 24)
    return _encode_batch("gmr1_hip_xch_dc12_encode_batch", 432, l2.shape[0], [C.c_int(l2.shape[0])], [l2])


def facch3_encode_batch(l2, bits_s, ciph=None):
    """l2 (n, 10), bits_s (n, 32), ciph (n, 384) optional -> (n, 4, 104)"""
    l2 = np.ascontiguousarray(l2, np.uint8).reshape(-1, 10)
    n = l2.shape[0]
    return _encode_batch("gmr1_hip_facch3_encode_batch", 416, n, [C.c_int(n)], [l2, bits_s, ciph]).reshape(n, 4, 104)


def tch3_encode_batch(frames, bits_s, m=0, ciph=None):
    """frames (n, 2, 10), bits_s (n, 4), ciph (n, 208) optional -> (n, 212)"""
    frames = np.ascontiguousarray(frames, np.uint8).reshape(-1, 20)
    n = frames.shape[0]
    return _encode_batch("gmr1_hip_tch3_encode_batch", 212, n, [C.c_int(n), C.c_int(m)], [frames, bits_s, ciph])


def facch9_encode_batch(l2, sacch, status, ciph=None):
    l2 = np.ascontiguousarray(l2, np.uint8).reshape(-1, 38)
    n = l2.shape[0]
    return _encode_batch("gmr1_hip_facch9_encode_batch", 662, n, [C.c_int(n)], [l2, sacch, status, ciph])


def tch9_encode_batch(l2, mode, seq_len, sacch, status, ciph=None):
    """l2 (n, 18 | 30 | 60): whole runs of seq_len consecutive bursts of one channel each -> (n, 662)"""
    l2 = np.ascontiguousarray(l2, np.uint8)
    l2 = l2.reshape(-1, (18, 30, 60)[mode] if 0 <= mode < 3 else l2.shape[-1])
    n = l2.shape[0]
    return _encode_batch("gmr1_hip_tch9_encode_batch", 662, n, [C.c_int(mode), C.c_int(n), C.c_int(seq_len)],
                         [l2, sacch, status, ciph])


def rach_encode_batch(rach, sb_mask):
    rach = np.ascontiguousarray(rach, np.uint8).reshape(-1, 18)
    n = rach.shape[0]
    return _encode_batch("gmr1_hip_rach_encode_batch", 494, n, [C.c_int(n)], [rach, np.asarray(sb_mask, np.uint8).reshape(n)])


def mod_batch(burst, ebits, sync_id=0):
    """ebits (n, burst.ebits) ubits -> (n, burst.len) complex64 symbols at one sample per symbol"""
    bid = BURST_IDS.index(burst) if isinstance(burst, str) else int(burst)
    b = burst_info(bid)
    eb, p_eb = _np(ebits, np.uint8)
    eb = eb.reshape(-1, b.ebits)
    out = np.zeros((eb.shape[0], b.len), np.complex64)
    _check(load().gmr1_hip_mod_batch(C.c_int(bid), C.c_int(sync_id), C.c_int(eb.shape[0]), p_eb,
                                     out.ctypes.data_as(C.c_void_p)), "gmr1_hip_mod_batch")
    return out


def encode_single(chain, *args):
    """The reference's own single calls gmr1_<chain>_encode (void unless xch_dc12): returns the burst bits.
    bcch / ccch / xch_dc12: (l2); facch3: (l2, bits_s, ciph | None); tch3: (frame0, frame1, bits_s, ciph | None, m);
    facch9: (l2, sacch, status, ciph | None); rach: (rach, sb_mask)."""
    n_out = {"bcch": 424, "ccch": 432, "xch_dc12": 432, "facch3": 416, "tch3": 212, "facch9": 662, "rach": 494}[chain]
    e = np.full(n_out, 255, np.uint8)
    f = getattr(load(), "gmr1_%s_encode" % chain)
    f.restype = C.c_int if chain == "xch_dc12" else None
    cargs, keep = [], []
    for a in args:
        if a is None:
            cargs.append(None)
        elif isinstance(a, (int, np.integer)):
            cargs.append(C.c_uint8(int(a)) if chain == "rach" else C.c_int(int(a)))
        else:
            k, p = _np(a, np.uint8)
            keep.append(k)
            cargs.append(p)
    rc = f(e.ctypes.data_as(C.c_void_p), *cargs)
    if chain == "xch_dc12":
        _check(rc, "gmr1_xch_dc12_encode")
    if (e == 255).any():
        raise Gmr1HipError("gmr1_%s_encode: %s" % (chain, load().gmr1_hip_last_error().decode(errors="replace")))
    return e


class Tch9Encoder:
    """One TCH9 channel encoded burst by burst with the reference's stateful calls (tch9.h:47-49)."""

    def __init__(self, mode: int):
        self.mode = mode
        self.il = Interleaver()
        f = load().gmr1_interleaver_init
        f.restype = C.c_int
        _check(f(C.byref(self.il), C.c_int(3), C.c_int(648)), "gmr1_interleaver_init")

    def encode(self, l2, sacch, status, ciph=None):
        l2, p_l2 = _np(l2, np.uint8)
        sa, p_sa = _np(sacch, np.uint8)
        stt, p_st = _np(status, np.uint8)
        c, p_c = _opt(ciph)
        e = np.full(662, 255, np.uint8)
        f = load().gmr1_tch9_encode
        f.restype = None
        f(e.ctypes.data_as(C.c_void_p), p_l2, C.c_int(self.mode), p_sa, p_st, p_c, C.byref(self.il))
        if (e == 255).any():
            raise Gmr1HipError("gmr1_tch9_encode: %s" % load().gmr1_hip_last_error().decode(errors="replace"))
        return e

    def close(self):
        if self.il.bits_cpp:
            load().gmr1_interleaver_fini(C.byref(self.il))
            self.il.bits_cpp = None


def pi4cxpsk_mod(burst_name: str, ebits, sync_id=0, max_len=None):
    """The reference's gmr1_pi4cxpsk_mod on one of the exported burst objects -> (rc, symbols)"""
    lib_ = load()
    bt = C.c_void_p.in_dll(lib_, "gmr1_%s_burst" % burst_name)   # address of the exported struct
    b = burst_info(burst_name)
    n = b.len if max_len is None else max_len
    data = np.zeros(max(n, 1), np.complex64)
    v = CxVec()
    v.len = 0
    v.max_len = n
    v.flags = 0
    v.data = data.ctypes.data_as(C.c_void_p)
    eb, p_eb = _np(ebits, np.uint8)
    f = lib_.gmr1_pi4cxpsk_mod
    f.restype = C.c_int
    rc = f(C.c_void_p(C.addressof(bt)), p_eb, C.c_int(sync_id), C.byref(v))
    return rc, data[:v.len].copy()


# ---- stand-alone layer-1 primitives (reference scramb.h / interleave.h), each one blocking GPU call ----------
def _prim(fname, x, dtype, *lead):
    x, p = _np(x, dtype)
    out = np.full(x.shape, 77, dtype)
    f = getattr(load(), fname)
    f.restype = None
    f(*lead, out.ctypes.data_as(C.c_void_p), p, *([C.c_int(x.size)] if "scramble" in fname else []))
    return out


def scramble_sbit(x):
    return _prim("gmr1_scramble_sbit", x, np.int8)


def scramble_ubit(x):
    return _prim("gmr1_scramble_ubit", x, np.uint8)


def interleave_intra(x, N, inverse=False):
    x, p = _np(x, np.uint8)
    assert x.size == 8 * N
    out = np.full(8 * N, 77, np.uint8)
    f = getattr(load(), "gmr1_deinterleave_intra" if inverse else "gmr1_interleave_intra")
    f.restype = None
    f(out.ctypes.data_as(C.c_void_p), p, C.c_int(N))
    return out


class InterBurstInterleaver:
    """gmr1_interleaver_init(il, 3, 648) + gmr1_interleave_inter / gmr1_deinterleave_inter on that object."""

    def __init__(self):
        self.il = Interleaver()
        f = load().gmr1_interleaver_init
        f.restype = C.c_int
        _check(f(C.byref(self.il), C.c_int(3), C.c_int(648)), "gmr1_interleaver_init")

    def _call(self, fname, x):
        x, p = _np(x, np.uint8)
        assert x.size == 648
        out = np.full(648, 77, np.uint8)
        f = getattr(load(), fname)
        f.restype = None
        f(C.byref(self.il), out.ctypes.data_as(C.c_void_p), p)
        return out

    def interleave(self, bits_ep):
        return self._call("gmr1_interleave_inter", bits_ep)

    def deinterleave(self, bits_epp):
        return self._call("gmr1_deinterleave_inter", bits_epp)

    def close(self):
        if self.il.bits_cpp:
            load().gmr1_interleaver_fini(C.byref(self.il))
            self.il.bits_cpp = None


# ---- gmr1_hip_shard.h: the receive loop over the ranks of a node, RCCL exchanges inside the library ------------------
class Shard:
    """One rank's end of the node-wide communicator (gmr1_hip_shard_create).  `id_bytes`: the 128-byte id made by
    Shard.unique_id() on one rank and handed to the others by the host program (here: torch.distributed)."""

    def __init__(self, id_bytes: bytes, rank: int, world: int):
        self._h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(id_bytes)
        _check(load().gmr1_hip_shard_create(C.byref(self._h), buf, C.c_int(rank), C.c_int(world)), "gmr1_hip_shard_create")
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        _check(load().gmr1_hip_shard_unique_id(buf), "gmr1_hip_shard_unique_id")
        return bytes(buf)

    def rx_run(self, stream, iq_ptr, offset, length, sps=4, arfcn=None, root=0, max_records=1 << 17, resident=False):
        """gmr1_hip_rx_run_sharded (resident: gmr1_hip_rx_run_sharded_resident -- iq_ptr is this rank's own memory holding
        the carriers it owns, nothing is scattered).  Returns (records, status, n_chains, timing_ms) on root,
        (None, None, None, timing_ms) elsewhere."""
        offset, p_off = _np(offset, np.uint64)
        length, p_len = _np(length, np.uint64)
        n = len(offset)
        p_arfcn = None
        if arfcn is not None:
            arfcn, p_arfcn = _np(arfcn, np.uint16)
        is_root = self.rank == root
        out = np.empty(max(max_records, 1), RX_RECORD) if is_root else None
        n_rec = C.c_int(0)
        status = np.zeros(max(n, 1), np.int32)
        chains = np.zeros(max(n, 1), np.int32)
        timing = np.zeros(3, np.float32)
        f = load().gmr1_hip_rx_run_sharded_resident if resident else load().gmr1_hip_rx_run_sharded
        f.restype = C.c_int
        rc = f(self._h, C.c_void_p(stream) if stream else None, C.c_int(root), C.c_int(n), C.c_int(sps),
               C.c_void_p(iq_ptr) if iq_ptr else None, p_off, p_len, p_arfcn,
               out.ctypes.data_as(C.c_void_p) if is_root else None, C.c_int(max_records), C.byref(n_rec),
               status.ctypes.data_as(C.c_void_p), chains.ctypes.data_as(C.c_void_p), timing.ctypes.data_as(C.c_void_p))
        _check(rc, "gmr1_hip_rx_run_sharded_resident" if resident else "gmr1_hip_rx_run_sharded")
        if not is_root:
            return None, None, None, timing
        return out[:min(n_rec.value, max_records)].copy(), status[:n], chains[:n], timing

    def close(self):
        if self._h:
            load().gmr1_hip_shard_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------
# AMBE speech decoder (include/osmocom/gmr1/codec/codec.h, gmr1_hip_codec_*)
# ---------------------------------------------------------------------------
CODEC_CLEARED, CODEC_FRESH = 1, 2


def codec_state_bytes() -> int:
    f = load().gmr1_hip_codec_state_bytes
    f.restype = C.c_size_t
    return f()


def codec_decode_batch(frames, state=None, flags=0):
    """frames (n_ch, n_frames, 10) uint8 -> (pcm (n_ch, n_frames, 160) int16, rv (n_ch, n_frames) int32, state).
    state: None = fresh decoders, or the uint8 array a previous call returned (continues those channels)."""
    frames, p_f = _np(frames, np.uint8)
    assert frames.ndim == 3 and frames.shape[2] == 10
    n_ch, n_fr = frames.shape[:2]
    pcm = np.zeros((n_ch, n_fr, 160), np.int16)
    rv = np.zeros((n_ch, n_fr), np.int32)
    if state is None:
        state = np.zeros((n_ch, codec_state_bytes()), np.uint8)
        flags |= CODEC_FRESH
    else:
        state = np.ascontiguousarray(state, np.uint8).copy()
        assert state.shape == (n_ch, codec_state_bytes())
    _check(load().gmr1_hip_codec_decode_batch(C.c_int(n_ch), C.c_int(n_fr), p_f, pcm.ctypes.data_as(C.c_void_p),
                                              rv.ctypes.data_as(C.c_void_p), state.ctypes.data_as(C.c_void_p),
                                              C.c_int(flags)), "gmr1_hip_codec_decode_batch")
    return pcm, rv, state


def codec_init_dev(stream, n_ch, state_ptr, flags=0):
    _check(load().gmr1_hip_codec_init_dev(C.c_void_p(stream) if stream else None, C.c_int(n_ch), C.c_void_p(state_ptr),
                                          C.c_int(flags)), "gmr1_hip_codec_init_dev")


def codec_decode_batch_dev(stream, n_ch, n_frames, frames_ptr, pcm_ptr, rv_ptr, state_ptr):
    _check(load().gmr1_hip_codec_decode_batch_dev(C.c_void_p(stream) if stream else None, C.c_int(n_ch), C.c_int(n_frames),
                                                  C.c_void_p(frames_ptr), C.c_void_p(pcm_ptr),
                                                  C.c_void_p(rv_ptr) if rv_ptr else None, C.c_void_p(state_ptr)),
           "gmr1_hip_codec_decode_batch_dev")


class Codec:
    """The reference's one-channel object: gmr1_codec_alloc / decode_frame / decode_dtx / release."""

    def __init__(self):
        f = load().gmr1_codec_alloc
        f.restype = C.c_void_p
        self.h = f()
        if not self.h:
            raise Gmr1HipError("gmr1_codec_alloc returned NULL: " + load().gmr1_hip_last_error().decode(errors="replace"))

    def decode_frame(self, frame, N=160, bad=0):
        frame, p_f = _np(frame, np.uint8)
        audio = np.zeros(max(N, 160), np.int16)
        rc = load().gmr1_codec_decode_frame(C.c_void_p(self.h), audio.ctypes.data_as(C.c_void_p), C.c_int(N), p_f, C.c_int(bad))
        return audio, rc

    def decode_dtx(self, N=160):
        audio = np.ones(N, np.int16)
        rc = load().gmr1_codec_decode_dtx(C.c_void_p(self.h), audio.ctypes.data_as(C.c_void_p), C.c_int(N))
        return audio, rc

    def release(self):
        if self.h:
            load().gmr1_codec_release(C.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


# layout of the host table image (csrc/ambe_dev.h: struct AmbeTab), for the tests that check it without a GPU
CODEC_TAB_DTYPE = np.dtype([
    ("cosv", "<f4", 1024), ("win", "<f4", 128), ("f0_sf1", "<f4", 128), ("log2_L", "<f4", 64), ("tone_ampl", "<i4", 256),
    ("lcg_mul", "<u4", 128), ("lcg_add", "<u4", 128), ("gain", "<f4", 512), ("prba12", "<f4", 256), ("prba34", "<f4", 128),
    ("prba57", "<f4", 384), ("hoc", "<f4", (4, 512)), ("interp", "<f4", 4), ("perr14", "<f4", 256), ("perr58", "<f4", 128),
    ("rho", "<f4", 56), ("vuv", "<u2", 64), ("hpg", "u1", 192), ("f0_sf0", "<f4", (129, 128, 4)),
])


def codec_host_tables():
    img, n = C.c_void_p(), C.c_size_t()
    _check(load().gmr1_hip_codec_host_tables(C.byref(img), C.byref(n)), "gmr1_hip_codec_host_tables")
    assert n.value == CODEC_TAB_DTYPE.itemsize, (n.value, CODEC_TAB_DTYPE.itemsize)
    raw = C.string_at(img.value, n.value)
    return np.frombuffer(raw, CODEC_TAB_DTYPE)[0]


def codec_libm_check(which, x):
    x, p_x = _np(x, np.float32)
    out = np.zeros(x.size, np.float32)
    _check(load().gmr1_hip_codec_libm_check(C.c_int(which), C.c_int(x.size), p_x, out.ctypes.data_as(C.c_void_p)),
           "gmr1_hip_codec_libm_check")
    return out
