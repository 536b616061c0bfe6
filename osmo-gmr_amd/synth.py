"""Seeded synthetic GMR-1 signal generator (numpy, host side).

Produces the inputs BASELINE.md's configs describe: random L2 payloads run
through this module's OWN channel encoders (written from the ETSI chains the
reference implements in src/l1/{bcch,ccch,facch3,tch3}.c), mapped onto the
burst formats of src/sdr/nb.c, pi/4-rotated, pulse shaped to sps samples per
symbol and impaired (timing, CFO, phase, gain, AWGN).  Also the FCCH
dual-chirp streams of config 2.

Nothing here touches ``oracle/``: tests cross-check these encoders against the
oracle's, and bench.py uses this module to fill HBM with its workload.  It is
test-data generation, not a product path: the product's own encoders and
modulator are the GPU ones (csrc/tx_kernels.hip, ``api.*_encode_batch``,
``api.mod_batch``); these numpy ones are a third, independent implementation
that both the oracle and the GPU path are checked against.

Pulse: the reference demodulator samples symbols directly (no matched filter,
pi4cxpsk.c:286-297) because the channelizer in utils/gmr1_rx_sdr.py:523-529
already applied the receive RRC(0.35); the signal it sees is therefore a
raised-cosine (alpha 0.35) pulse train, which is what is generated here, with
noise added at the matched-filter output (variance N0 per complex sample for
unit symbol energy).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

SYM_RATE = 23400


# --------------------------------------------------------------------------
# burst formats
# --------------------------------------------------------------------------
@dataclass
class BurstFormat:
    name: str
    rotation: float
    nbits: int
    length: int
    ebits: int
    sync: list  # list (per sync sequence) of list of (pos, [symbols])
    data: list  # list of (pos, len)

    def data_positions(self) -> np.ndarray:
        return np.concatenate([np.arange(p, p + l) for p, l in self.data])


# --------------------------------------------------------------------------
# bit-level primitives (vectorised over a leading batch axis)
# --------------------------------------------------------------------------
def unpack_lsb(data: np.ndarray, nbits: int) -> np.ndarray:
    """bytes (B, n) -> bits (B, nbits), bit k = byte k>>3, bit k&7."""
    data = np.asarray(data, dtype=np.uint8)
    bits = np.unpackbits(data, axis=-1, bitorder="little")
    return bits[..., :nbits]


def unpack_msb(data: np.ndarray, nbits: int) -> np.ndarray:
    data = np.asarray(data, dtype=np.uint8)
    bits = np.unpackbits(data, axis=-1, bitorder="big")
    return bits[..., :nbits]


def crc_bits(bits: np.ndarray, nbits_crc: int, poly: int) -> np.ndarray:
    """MSB-first CRC register, init 0, no final xor; returns (B, nbits_crc)."""
    bits = np.asarray(bits, dtype=np.uint32)
    top = 1 << (nbits_crc - 1)
    mask = (1 << nbits_crc) - 1
    crc = np.zeros(bits.shape[:-1], dtype=np.uint32)
    for i in range(bits.shape[-1]):
        crc ^= bits[..., i] << (nbits_crc - 1)
        hi = (crc & top) != 0
        crc = (crc << 1) & mask
        crc = np.where(hi, crc ^ (poly & mask), crc)
    out = np.stack([(crc >> (nbits_crc - 1 - j)) & 1 for j in range(nbits_crc)], axis=-1)
    return out.astype(np.uint8)


def conv_encode(bits: np.ndarray, polys, K: int, flush: bool = True,
                tail_biting: bool = False) -> np.ndarray:
    """Feed-forward convolutional encoder; poly bit i is the D^i tap.

    Output (B, steps*N) in the order g0,g1,.. per input bit."""
    bits = np.asarray(bits, dtype=np.uint8)
    B, L = bits.shape
    if tail_biting:
        pre = bits[:, L - (K - 1):]
        seq = np.concatenate([pre, bits], axis=1)
        steps = L
    elif flush:
        seq = np.concatenate([np.zeros((B, K - 1), np.uint8), bits,
                              np.zeros((B, K - 1), np.uint8)], axis=1)
        steps = L + K - 1
    else:
        seq = np.concatenate([np.zeros((B, K - 1), np.uint8), bits], axis=1)
        steps = L
    out = np.zeros((B, steps, len(polys)), np.uint8)
    for j, g in enumerate(polys):
        acc = np.zeros((B, steps), np.uint8)
        for d in range(K):
            if (g >> d) & 1:
                acc ^= seq[:, K - 1 - d:K - 1 - d + steps]
        out[:, :, j] = acc
    return out.reshape(B, steps * len(polys))


def interleave_intra(bits: np.ndarray, N: int) -> np.ndarray:
    kc = np.arange(8 * N)
    kep = N * ((5 * kc) & 7) + (kc >> 3)
    out = np.empty_like(bits)
    out[..., kep] = bits[..., kc]
    return out


def scramble_mask(n: int) -> np.ndarray:
    r = 0x4D4B
    out = np.zeros(n, np.uint8)
    for i in range(n):
        b = ((r >> 14) ^ r) & 1
        r = ((r << 1) | b) & 0xFFFF
        out[i] = b
    return out


K5_12 = (0x19, 0x17)
K5_14 = (0x19, 0x17, 0x15, 0x1F)
K7_TCH3 = (0x6D, 0x4F)


# --------------------------------------------------------------------------
# channel encoders
# --------------------------------------------------------------------------
def bcch_encode(l2: np.ndarray) -> np.ndarray:
    """(B,24) bytes -> (B,424) encoded hard bits."""
    u = unpack_lsb(l2, 192)
    u = np.concatenate([u, crc_bits(u, 16, 0x1021)], axis=1)
    c = conv_encode(u, K5_12, 5)
    ep = interleave_intra(c, 53)
    return ep ^ scramble_mask(424)


def ccch_encode(l2: np.ndarray) -> np.ndarray:
    """(B,24) bytes -> (B,432) encoded hard bits (4 + 424 + 4)."""
    u = unpack_lsb(l2, 192)
    u = np.concatenate([u, crc_bits(u, 16, 0x1021)], axis=1)
    c = conv_encode(u, K5_12, 5)
    B = c.shape[0]
    ep = np.zeros((B, 432), np.uint8)
    ep[:, 4:428] = interleave_intra(c, 53)
    return ep ^ scramble_mask(432)


def facch3_encode(l2: np.ndarray, bits_s: np.ndarray, ciph: np.ndarray | None = None) -> np.ndarray:
    """(B,10) bytes (76 bits used), (B,32) status bits [, (B,384) cipher stream] -> (B,4,104)."""
    u = unpack_lsb(l2, 76)
    u = np.concatenate([u, crc_bits(u, 16, 0x1021)], axis=1)
    c = conv_encode(u, K5_14, 5)          # (B, 384)
    B = c.shape[0]
    i = np.arange(384)
    cp = np.empty_like(c)
    cp[:, (i & 3) * 96 + (i >> 2)] = c
    out = np.zeros((B, 4, 104), np.uint8)
    scr = scramble_mask(96)
    for b in range(4):
        xmy = interleave_intra(cp[:, 96 * b:96 * b + 96], 12) ^ scr
        if ciph is not None:
            xmy = xmy ^ ciph[:, 96 * b:96 * b + 96]
        out[:, b, :22] = xmy[:, :22]
        out[:, b, 22:30] = bits_s[:, 8 * b:8 * b + 8]
        out[:, b, 30:] = xmy[:, 22:]
    return out


K9_13 = (0x1ED, 0x19B, 0x127)      # reference src/l1/conv.c:345-351
P1213 = np.array([1, 1, 0, 1, 0, 1, 0, 1, 1] * 4 + [1, 1, 1], np.uint8)     # punct.c:1105-1125, 0 = punctured


def xch_dc12_encode(l2: np.ndarray) -> np.ndarray:
    """(B,24) bytes -> (B,432) encoded hard bits of a DC12 burst (reference src/l1/xch_dc12.c:64-79):
    CRC16, K=9 rate 1/3 tail-biting, P(12;13) puncturing, intra-burst interleaver N=54, scrambler."""
    u = unpack_lsb(l2, 192)
    u = np.concatenate([u, crc_bits(u, 16, 0x1021)], axis=1)
    c = conv_encode(u, K9_13, 9, tail_biting=True)
    c = c[:, np.tile(P1213, 16) != 0]
    ep = interleave_intra(c, 54)
    return ep ^ scramble_mask(432)


def rach_encode(rach: np.ndarray, sb_mask: int) -> np.ndarray:
    """(B,18) bytes -> (B,494) encoded hard bits of a RACH burst (reference src/l1/rach.c:78-125): 16 class-1
    bits + CRC8 (xor SB mask), 123 class-2 bits + CRC12, K=5 rate 1/4 with bits 2, 3 of the first 135 steps
    punctured, class-1 part sent twice."""
    rach = np.asarray(rach, np.uint8)
    bits = unpack_lsb(rach, 144)
    u1 = bits[:, :16]
    u2 = bits[:, 16:139]
    c1 = crc_bits(u1, 8, 0x9B) ^ np.array([(sb_mask >> (7 - i)) & 1 for i in range(8)], np.uint8)
    u = np.concatenate([u2, crc_bits(u2, 12, 0x80F), u1, c1], axis=1)
    c = conv_encode(u, K5_14, 5).reshape(-1, 163, 4)
    c = np.concatenate([c[:, :135, :2].reshape(-1, 270), c[:, 135:, :].reshape(-1, 112)], axis=1)
    e1p = interleave_intra(c[:, 270:], 14)
    e2p = c[:, :270].copy()
    e2p[:, :264] = interleave_intra(c[:, :264], 33)
    x = np.concatenate([e1p, e2p, e1p], axis=1) ^ scramble_mask(494)
    return np.concatenate([x[:, 112:248], x[:, :112], x[:, 382:494], x[:, 248:382]], axis=1)


def tch3_perm() -> np.ndarray:
    kc = np.arange(104)
    ii, ij = kc % 24, kc // 24
    return np.where(ii < 8, ij + 5 * ii, ij + 4 * ii + 8)


def tch3_encode(frame0: np.ndarray, frame1: np.ndarray, bits_s: np.ndarray, m: int = 0,
                ciph: np.ndarray | None = None) -> np.ndarray:
    """two (B,10) speech frames + (B,4) status [, (B,208) cipher stream] -> (B,212)."""
    B = frame0.shape[0]
    epp = np.zeros((B, 208), np.uint8)
    perm = tch3_perm()
    for i, fr in enumerate((frame0, frame1)):
        d = unpack_msb(fr, 80)
        c96 = conv_encode(d[:, :48], K7_TCH3, 7, flush=False, tail_biting=True)
        keep = (np.arange(96) % 4) != 3           # P(1;2): every 4th bit punctured
        c = np.concatenate([c96[:, keep], d[:, 48:]], axis=1)   # 72 + 32
        ep = np.empty((B, 104), np.uint8)
        ep[:, perm] = c
        if m:
            epp[:, 104 * i:104 * i + 104] = ep
        else:
            epp[:, i::2] = ep
    xmy = epp ^ scramble_mask(208)
    if ciph is not None:
        xmy = xmy ^ ciph
    out = np.zeros((B, 212), np.uint8)
    out[:, :52] = xmy[:, :52]
    out[:, 52:56] = bits_s
    out[:, 56:] = xmy[:, 52:]
    return out


# --------------------------------------------------------------------------
# modulation + channel
# --------------------------------------------------------------------------
_CQPSK_BITS2SYM = np.array([0, 1, 3, 2])   # bits (b0 b1 MSB first) -> symbol index


def map_symbols(fmt: BurstFormat, ebits: np.ndarray, sync_id=0) -> np.ndarray:
    """(B, ebits) hard bits -> (B, length) complex64 symbols at 1 sps, pi/4 rotated."""
    B = ebits.shape[0]
    sym = np.zeros((B, fmt.length), np.complex64)
    sync_id = np.broadcast_to(np.asarray(sync_id), (B,))
    for sid in np.unique(sync_id):
        rows = np.nonzero(sync_id == sid)[0]
        for pos, syms in fmt.sync[int(sid)]:
            idx = np.asarray(syms)
            if fmt.nbits == 2:
                v = np.exp(1j * (np.pi / 2) * idx)
            else:
                v = np.where(idx == 0, 1.0, -1.0)
            sym[np.ix_(rows, np.arange(pos, pos + len(syms)))] = v.astype(np.complex64)
    dpos = fmt.data_positions()
    if fmt.nbits == 2:
        idx = _CQPSK_BITS2SYM[(ebits[:, 0::2].astype(np.int64) << 1) | ebits[:, 1::2]]
        v = np.exp(1j * (np.pi / 2) * idx)
    else:
        v = 1.0 - 2.0 * ebits.astype(np.float64)
    sym[:, dpos] = v.astype(np.complex64)
    sym *= np.exp(1j * fmt.rotation * np.arange(fmt.length)).astype(np.complex64)
    return sym


def rc_pulse(t: np.ndarray, alpha: float = 0.35) -> np.ndarray:
    """Raised-cosine pulse, t in symbols."""
    t = np.asarray(t, dtype=np.float64)
    den = 1.0 - (2.0 * alpha * t) ** 2
    sing = np.abs(den) < 1e-9
    den = np.where(sing, 1.0, den)
    g = np.sinc(t) * np.cos(np.pi * alpha * t) / den
    return np.where(sing, (np.pi / 4) * np.sinc(1.0 / (2 * alpha)), g)


@dataclass
class BurstBatch:
    iq: np.ndarray          # (B, stride) complex64 windows
    stride: int
    in_len: int             # samples of each window that belong to the burst window
    toa: np.ndarray         # true TOA (samples, fractional) relative to window start
    cfo: np.ndarray         # rad / sample
    extra: dict = field(default_factory=dict)


def synth_windows(fmt: BurstFormat, symbols: np.ndarray, sps: int, win: int, rng: np.random.Generator,
                  *, toa_jitter: int = 0, frac: bool = False, cfo_hz_std: float = 0.0,
                  esn0_db=None, gain_db_std: float = 0.0, span: int = 5,
                  stride: int | None = None) -> BurstBatch:
    """Shape (B, length) symbols into (B, stride) sample windows of length*sps + win.

    The burst nominally starts win/2 samples into the window (gmr1_rx.c:149-170).
    """
    B, L = symbols.shape
    in_len = L * sps + win
    if stride is None:
        stride = in_len
    e_toa = win // 2
    jit = rng.integers(-toa_jitter, toa_jitter + 1, size=B) if toa_jitter else np.zeros(B, np.int64)
    fr = rng.uniform(-0.5, 0.5, size=B) if frac else np.zeros(B)
    toa = e_toa + jit + fr

    # burst-relative samples 4i+p, i in [-span, L+span): x[n] = sum_m s[i-m] g(m + (p - fr)/sps)
    pad = span
    spad = np.zeros((B, L + 4 * pad), np.complex64)
    spad[:, 2 * pad:2 * pad + L] = symbols
    n_sym_out = L + 2 * pad
    body = np.zeros((sps, B, n_sym_out), np.complex64)      # phase-major: contiguous accumulations
    tmp = np.empty((B, n_sym_out), np.complex64)
    for p in range(sps):
        for m in range(-span, span + 1):
            coef = rc_pulse(m + (p - fr) / sps).astype(np.float32)      # (B,)
            # output symbol slot i (burst index i - pad) uses s[i - pad - m]
            np.multiply(spad[:, pad - m:pad - m + n_sym_out], coef[:, None], out=tmp)
            body[p] += tmp
    # sample k of the burst body <-> burst-relative n = k - pad*sps
    body = np.ascontiguousarray(body.transpose(1, 2, 0)).reshape(B, n_sym_out * sps)

    out = np.zeros((B, stride), np.complex64)
    start = (e_toa + jit) - pad * sps                 # window index of body[0]
    for s0 in np.unique(start):
        rows = np.nonzero(start == s0)[0]
        lo = max(0, s0)
        hi = min(in_len, s0 + body.shape[1])
        out[rows, lo:hi] = body[rows, lo - s0:hi - s0]
    del body, tmp

    n = np.arange(in_len, dtype=np.float32)
    cfo = rng.normal(0.0, cfo_hz_std, size=B) * (2 * np.pi / (SYM_RATE * sps)) if cfo_hz_std else np.zeros(B)
    ph0 = rng.uniform(0, 2 * np.pi, size=B)
    ph = ph0.astype(np.float32)[:, None] + cfo.astype(np.float32)[:, None] * n[None, :]
    rot = np.empty((B, in_len), np.complex64)
    rot.real = np.cos(ph)
    rot.imag = np.sin(ph)
    out[:, :in_len] *= rot
    del rot, ph

    if esn0_db is not None:
        esn0 = np.broadcast_to(np.asarray(esn0_db, dtype=np.float64), (B,))
        sigma = np.sqrt(10.0 ** (-esn0 / 10.0) / 2.0).astype(np.float32)
        noise = rng.standard_normal((B, in_len * 2), dtype=np.float32)
        noise *= sigma[:, None]
        out[:, :in_len] += noise.view(np.complex64)
        del noise

    if gain_db_std:
        g = 10.0 ** (rng.normal(0.0, gain_db_std, size=B) / 20.0)
        out *= g[:, None].astype(np.float32)

    return BurstBatch(iq=out, stride=stride, in_len=in_len, toa=toa, cfo=cfo)


def rrc_pulse(t: np.ndarray, alpha: float = 0.35) -> np.ndarray:
    """Root-raised-cosine pulse (unit energy per symbol at 1 sample / symbol), t in symbols."""
    t = np.asarray(t, dtype=np.float64)
    out = np.empty_like(t)
    z = np.abs(t) < 1e-9
    s = np.abs(np.abs(4.0 * alpha * t) - 1.0) < 1e-9
    g = ~(z | s)
    out[z] = 1.0 - alpha + 4.0 * alpha / np.pi
    out[s] = (alpha / np.sqrt(2.0)) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * alpha)) + (1 - 2 / np.pi) * np.cos(np.pi / (4 * alpha)))
    tg = t[g]
    out[g] = (np.sin(np.pi * tg * (1 - alpha)) + 4 * alpha * tg * np.cos(np.pi * tg * (1 + alpha))) / (np.pi * tg * (1 - (4 * alpha * tg) ** 2))
    return out


def shape_bursts(symbols: np.ndarray, sps: int, frac, span: int = 5, pulse: str = "rc") -> np.ndarray:
    """Raised-cosine (pulse="rc": what the demodulator expects to see) or root-raised-cosine
    (pulse="rrc": what a transmitter sends, before the receive filter of a channelizer) shaping of
    (B, L) symbol rows to sps samples per symbol.

    Returns (B, (L + 2*span) * sps): sample k belongs to burst-relative time (k - span*sps) samples,
    delayed by `frac` (scalar or (B,)) samples."""
    B, L = symbols.shape
    fr = np.broadcast_to(np.asarray(frac, dtype=np.float64), (B,))
    pad = span
    spad = np.zeros((B, L + 4 * pad), np.complex64)
    spad[:, 2 * pad:2 * pad + L] = symbols
    n_sym_out = L + 2 * pad
    body = np.zeros((sps, B, n_sym_out), np.complex64)
    tmp = np.empty((B, n_sym_out), np.complex64)
    for p in range(sps):
        for m in range(-span, span + 1):
            coef = (rrc_pulse if pulse == "rrc" else rc_pulse)(m + (p - fr) / sps).astype(np.float32)
            np.multiply(spad[:, pad - m:pad - m + n_sym_out], coef[:, None], out=tmp)
            body[p] += tmp
    return np.ascontiguousarray(body.transpose(1, 2, 0)).reshape(B, n_sym_out * sps)


def imm_ass_payload(rng: np.random.Generator, tn: int, p: int) -> np.ndarray:
    """A CCCH message gmr1_rx takes for an IMMEDIATE ASSIGNMENT (ccch_is_imm_ass / ccch_imm_ass_parse,
    reference src/gmr1_rx.c:235-246): receive timeslot tn (5 bits), DKAB position p (6 bits)."""
    l2 = rng.integers(0, 256, size=24, dtype=np.uint8)
    l2[1], l2[2] = 0x06, 0x3F
    l2[8] = ((p & 0x3F) << 2) | ((tn >> 3) & 0x03)
    l2[9] = ((tn & 0x07) << 5) | (l2[9] & 0x1F)
    return l2


def si1_payload(rng: np.random.Generator, fn: np.ndarray, delay: int, stn: int) -> np.ndarray:
    """BCCH System Information type 1 with a 'Seg 2A bis' carrying the TDMA position, as
    bcch_tdma_align() parses it (reference src/gmr1_rx.c:194-233).  fn is the frame number of
    the frame the burst is sent in; its low three bits must equal (2 + delay) & 7."""
    fn = np.asarray(fn, dtype=np.int64)
    B = fn.size
    assert np.all((fn & 7) == ((2 + delay) & 7))
    l2 = rng.integers(0, 256, size=(B, 24), dtype=np.uint8)
    superframe = (fn >> 6) & 0x1FFF
    multiframe = (fn >> 4) & 3
    mffn_hi = (fn >> 3) & 1
    l2[:, 0] = 0x08 | (l2[:, 0] & 0x07)
    l2[:, 9] = 0x80 | (l2[:, 9] & 0x03)
    l2[:, 10] = ((delay & 0x0F) << 3) | ((stn >> 2) & 0x07)
    l2[:, 11] = ((stn & 3) << 6) | ((superframe >> 7) & 0x3F)
    l2[:, 12] = ((superframe & 0x7F) << 1) | ((multiframe >> 1) & 1)
    l2[:, 13] = ((multiframe & 1) << 7) | (mffn_hi << 6) | (l2[:, 13] & 0x3F)
    return l2


def synth_bcch_carrier(fmt_bcch: BurstFormat, fmt_dc6: BurstFormat, n_samples: int, sps: int,
                       rng: np.random.Generator, *, stn: int = 3, delay: int = 2, fn0: int | None = None,
                       t0: int | None = None, frac: float = 0.0, esn0_db: float = 15.0, cfo_hz: float = 0.0,
                       p_idle: float = 0.15, fcch_db: float = 0.0, imm_ass=(), pulse: str = "rc", span: int = 5,
                       other_first: int = 0, si1_lie=None, absent_bcch=()):
    """One ARFCN of BASELINE.md config 4: FCCH + BCCH (SI1 w/ Seg 2A bis) + CCCH on the
    24-slot / 40 ms TDMA grid (reference src/gmr1_rx.c:852-895 schedule).

    Frame k starts at sample t0 + k*24*39*sps and has frame number fn0 + k; with
    sirfn = (fn - delay) & 63:  sirfn % 8 == 0 -> FCCH, == 2 -> BCCH, else CCCH (DC6) unless idle;
    all on timeslot stn.  imm_ass = [(k, tn, p), ...]: the first CCCH burst sent at frame index >= k
    carries an IMMEDIATE ASSIGNMENT to timeslot tn with DKAB position p (gmr1_rx.c:235-246).
    What a receiver's feedback loop has to survive (counted over the carrier's BCCH bursts in order): the first
    other_first of them carry another SI (no TDMA position: the first SI1 arrives late); si1_lie = {i: (delay, stn)}:
    burst i's SI1 claims that TDMA position instead of the true one; absent_bcch: bursts not transmitted at all.
    Returns (stream complex64, list of dicts describing what was sent)."""
    frame_len = 24 * 39 * sps
    if t0 is None:
        t0 = int(rng.integers(0, frame_len))
    if fn0 is None:
        fn0 = int(rng.integers(0, 1 << 18))
    sigma = np.sqrt(10.0 ** (-esn0_db / 10.0) / 2.0)
    x = rng.standard_normal((n_samples, 2), dtype=np.float32).view(np.complex64).reshape(-1)
    x *= np.float32(sigma)
    n_frames = (n_samples - t0) // frame_len + 1
    fns = fn0 + np.arange(n_frames)
    sirfn = (fns - delay) & 63
    sent = []
    # FCCH
    chirp = fcch_dual_chirp(0.32, 117, sps, frac) * np.float32(10.0 ** (fcch_db / 20.0) / np.sqrt(2.0) * np.sqrt(2.0))
    for k in np.nonzero(sirfn % 8 == 0)[0]:
        pos = t0 + k * frame_len + stn * 39 * sps
        if pos >= 0 and pos + chirp.size <= n_samples:
            x[pos:pos + chirp.size] += chirp
            sent.append(dict(type="fcch", fn=int(fns[k]), pos=pos))
    # BCCH
    kb = np.nonzero(sirfn % 8 == 2)[0]
    if kb.size:
        l2 = si1_payload(rng, fns[kb], delay, stn)
        other = rng.random(kb.size) < 0.25            # some BCCH bursts carry another SI: no TDMA info
        other[:other_first] = True
        for i, (d_l, s_l) in (si1_lie or {}).items():
            if i < kb.size:
                l2[i] = si1_payload(rng, fns[kb[i]:kb[i] + 1] - delay + d_l, d_l, s_l)[0]
                other[i] = False
        l2[other, 0] = 0x10 | (l2[other, 0] & 0x07)
        body = shape_bursts(map_symbols(fmt_bcch, bcch_encode(l2)), sps, frac, span, pulse)
        gone = set(int(i) for i in absent_bcch)
        for i, k in enumerate(kb):
            pos = t0 + k * frame_len + stn * 39 * sps - span * sps
            if i in gone:
                continue
            if pos >= 0 and pos + body.shape[1] <= n_samples:
                x[pos:pos + body.shape[1]] += body[i]
                sent.append(dict(type="bcch", fn=int(fns[k]), pos=pos + span * sps, l2=l2[i].copy()))
    # CCCH
    kc = np.nonzero((sirfn % 8 != 0) & (sirfn % 8 != 2))[0]
    kc = kc[rng.random(kc.size) >= p_idle]
    if kc.size:
        l2 = rng.integers(0, 256, size=(kc.size, 24), dtype=np.uint8)
        l2[:, 1] &= 0xF7                              # never an IMM.ASS (gmr1_rx.c:235-239) by accident
        ia_of = {}
        for (k_req, tn_a, p_a) in imm_ass:
            later = np.nonzero(kc >= k_req)[0]
            if later.size:
                i_a = int(later[0])
                l2[i_a] = imm_ass_payload(rng, tn_a, p_a)
                ia_of[i_a] = (tn_a, p_a)
        body = shape_bursts(map_symbols(fmt_dc6, ccch_encode(l2)), sps, frac, span, pulse)
        for i, k in enumerate(kc):
            pos = t0 + k * frame_len + stn * 39 * sps - span * sps
            if pos >= 0 and pos + body.shape[1] <= n_samples:
                x[pos:pos + body.shape[1]] += body[i]
                sent.append(dict(type="ccch", fn=int(fns[k]), pos=pos + span * sps, l2=l2[i].copy(), k=int(k),
                                 imm_ass=ia_of.get(i)))
    if cfo_hz:
        n = np.arange(n_samples, dtype=np.float64)
        ph = (2 * np.pi * cfo_hz / (SYM_RATE * sps)) * n
        x *= np.exp(1j * ph).astype(np.complex64)
    return x, sent


# --------------------------------------------------------------------------
# FCCH streams (config 2)
# --------------------------------------------------------------------------
def fcch_dual_chirp(freq: float, length: int, sps: int, frac: float = 0.0) -> np.ndarray:
    """sqrt(2) cos(phi(t)), phi = freq*2pi/len * (t - len/2)^2, t in symbols (fcch.c:167-193)."""
    t = (np.arange(length * sps) - frac) / sps - length / 2.0
    return (np.sqrt(2.0) * np.cos(freq * 2 * np.pi / length * t * t)).astype(np.float32)


def synth_fcch_stream(n_samples: int, sps: int, rng: np.random.Generator, *, snr_db: float = 6.0,
                      cfo_hz: float = 0.0, first: int | None = None, period_sym: int = 7488,
                      freq: float = 0.32, length: int = 117):
    """AWGN (variance 1 per complex sample) + FCCH dual chirp every period_sym symbols.

    Returns (stream complex64, list of true start samples)."""
    noise = rng.standard_normal((n_samples, 2), dtype=np.float32)
    x = ((noise[:, 0] + 1j * noise[:, 1]) / np.sqrt(2.0)).astype(np.complex64)
    amp = np.sqrt(10.0 ** (snr_db / 10.0) / 1.0)   # dual chirp has mean-square 1
    chirp = fcch_dual_chirp(freq, length, sps) * np.float32(amp)
    if first is None:
        first = int(rng.integers(0, period_sym * sps))
    starts = []
    pos = first
    while pos + chirp.size <= n_samples:
        x[pos:pos + chirp.size] += chirp
        starts.append(pos)
        pos += period_sym * sps
    if cfo_hz:
        n = np.arange(n_samples)
        x *= np.exp(1j * 2 * np.pi * cfo_hz / (SYM_RATE * sps) * n).astype(np.complex64)
    return x, starts


# --------------------------------------------------------------------------
# TCH3 follow-up (BASELINE.md config 4 with a traffic channel): A5/1, DKAB, the TCH carrier
# --------------------------------------------------------------------------
def a5_1(key, fn, nbits: int) -> np.ndarray:
    """GMR-1 A5/1 downlink keystream (reference src/l1/a5.c:222-282), vectorised over frames:
    key = 8 bytes, fn = (B,) frame numbers -> (B, nbits) bits."""
    key = np.asarray(key, np.uint8)
    fn = np.atleast_1d(np.asarray(fn, np.int64))
    B = fn.size
    lkey = np.tile(key[np.arange(8) ^ 1].astype(np.int64), (B, 1))
    lkey[:, 6] ^= (fn & 0x0000F) << 4
    lkey[:, 3] ^= (fn & 0x00030) << 2
    lkey[:, 1] ^= (fn & 0x007C0) >> 3
    lkey[:, 0] ^= (fn & 0x0F800) >> 11
    lkey[:, 0] ^= (fn & 0x70000) >> 11
    lkey &= 0xFF
    lens = (19, 22, 23, 17)
    taps = (0x072000, 0x311000, 0x660000, 0x013100)
    r = [np.zeros(B, np.int64) for _ in range(4)]

    def par(x):
        x = x ^ (x >> 16)
        x = x ^ (x >> 8)
        x = x ^ (x >> 4)
        x = x ^ (x >> 2)
        x = x ^ (x >> 1)
        return x & 1

    def clk(x, i):
        return ((x << 1) & ((1 << lens[i]) - 1)) | par(x & taps[i])

    for i in range(64):
        b = (lkey[:, i >> 3] >> (7 - (i & 7))) & 1
        for j in range(4):
            r[j] = clk(r[j], j) ^ b
    for j in range(4):
        r[j] |= 1

    def step():
        cb = [(r[3] >> 15) & 1, (r[3] >> 6) & 1, (r[3] >> 1) & 1]
        m = ((cb[0] + cb[1] + cb[2]) >= 2).astype(np.int64)
        for j in range(3):
            r[j] = np.where(cb[j] == m, clk(r[j], j), r[j])
        r[3] = clk(r[3], 3)

    def maj(x, a, b, c):
        return ((((x >> a) & 1) + ((x >> b) & 1) + ((x >> c) & 1)) >= 2).astype(np.int64)

    for _ in range(250):
        step()
    out = np.zeros((B, nbits), np.uint8)
    for i in range(nbits):
        step()
        m0 = maj(r[0], 1, 6, 15) ^ ((r[0] >> 11) & 1)
        m1 = maj(r[1], 3, 8, 14) ^ ((r[1] >> 1) & 1)
        m2 = maj(r[2], 4, 15, 19) ^ (r[2] & 1)
        out[:, i] = m0 ^ m1 ^ m2
    return out


def dkab_symbols(bits: np.ndarray, p: int) -> np.ndarray:
    """(B, 8) bits -> (B, 117) symbols of a DKAB: two keep-alive bursts of 5 pi/4-rotated BPSK symbols at
    symbol 2 + p and 2 + p + 59, differentially encoded (bit = 1: phase flips), nothing in between --
    what gmr1_dkab_demod looks for (reference src/sdr/dkab.c:57-181)."""
    bits = np.asarray(bits, np.uint8)
    B = bits.shape[0]
    sym = np.zeros((B, 117), np.complex64)
    for h in range(2):
        s = np.ones(B)
        base = 2 + p + 59 * h
        for k in range(5):
            if base + k < 117:
                sym[:, base + k] = s
            if k < 4:
                s = s * (1.0 - 2.0 * bits[:, 4 * h + k])
    sym *= np.exp(1j * (np.pi / 4) * np.arange(117)).astype(np.complex64)
    return sym


def synth_tch3_carrier(fmt_speech: BurstFormat, fmt_facch: BurstFormat, n_samples: int, sps: int,
                       rng: np.random.Generator, *, t0: int, fn0: int, k_start: int, tn: int, p: int,
                       kc=None, cipher_from: int | None = None, esn0_db: float = 25.0, cfo_hz: float = 0.0,
                       frac: float = 0.0, mix=(0.35, 0.35, 0.3), k_stop: int | None = None, ass_cmd=None):
    """The traffic carrier of a TCH3 assignment, time-aligned with the BCCH carrier (same t0 / fn0):
    from frame index k_start on, timeslot tn carries per frame a DKAB, an NT3 speech burst, or -- in
    groups of four frames with fn & 3 = 0..3 -- the four bursts of a FACCH3 message (sync sequence
    alternating per message).  mix = probabilities (dkab, speech, facch group) per decision.
    Bursts from frame index `cipher_from` on are A5/1-ciphered with kc (speech: fn of the burst,
    FACCH3: fn of each of its four bursts; reference src/gmr1_rx.c:390-399, 500-503).
    ass_cmd = (k, tn9): every FACCH3 message sent at frame index >= k is an ASSIGNMENT COMMAND 1 to
    timeslot tn9 (facch3_is_ass_cmd_1 / facch3_ass_cmd_1_parse, gmr1_rx.c:248-258).
    Returns (stream, sent list of dicts)."""
    frame_len = 24 * 39 * sps
    sigma = np.sqrt(10.0 ** (-esn0_db / 10.0) / 2.0)
    x = rng.standard_normal((n_samples, 2), dtype=np.float32).view(np.complex64).reshape(-1)
    x *= np.float32(sigma)
    span = 5
    n_frames = (n_samples - t0) // frame_len - 1
    if k_stop is not None:
        n_frames = min(n_frames, k_stop)
    sent = []

    def put(k, symbols):
        body = shape_bursts(symbols[None, :], sps, frac, span)[0]
        pos = t0 + k * frame_len + tn * 39 * sps - span * sps
        if pos >= 0 and pos + body.size <= n_samples:
            x[pos:pos + body.size] += body
            return True
        return False

    def ciphered(k):
        return kc is not None and cipher_from is not None and k >= cipher_from

    k = k_start
    facch_sid = 0
    while k < n_frames:
        fn = fn0 + k
        u = rng.random()
        if u < mix[0]:
            bits = rng.integers(0, 2, size=(1, 8), dtype=np.uint8)
            if put(k, dkab_symbols(bits, p)[0]):
                sent.append(dict(type="dkab", fn=fn, k=k, bits=bits[0]))
            k += 1
        elif u < mix[0] + mix[1] or (fn & 3) != 0 or k + 4 > n_frames:
            f0 = rng.integers(0, 256, size=(1, 10), dtype=np.uint8)
            f1 = rng.integers(0, 256, size=(1, 10), dtype=np.uint8)
            sb = rng.integers(0, 2, size=(1, 4), dtype=np.uint8)
            ciph = a5_1(kc, [fn], 208) if ciphered(k) else None
            e = tch3_encode(f0, f1, sb, 0, ciph)
            if put(k, map_symbols(fmt_speech, e)[0]):
                sent.append(dict(type="speech", fn=fn, k=k, frame0=f0[0], frame1=f1[0], ciph=ciph is not None))
            k += 1
        else:
            l2 = rng.integers(0, 256, size=(1, 10), dtype=np.uint8)
            l2[0, 9] &= 0x0F                                  # 76 bits
            if l2[0, 3] == 0x06 and l2[0, 4] == 0x2E:
                l2[0, 4] = 0                                  # never an ASSIGNMENT COMMAND 1 by accident
            if ass_cmd is not None and k >= ass_cmd[0]:
                l2[0, 3], l2[0, 4] = 0x06, 0x2E
                l2[0, 5] = (l2[0, 5] & 0xFC) | ((ass_cmd[1] >> 3) & 0x03)
                l2[0, 6] = ((ass_cmd[1] & 0x07) << 5) | (l2[0, 6] & 0x1F)
            sb = rng.integers(0, 2, size=(1, 32), dtype=np.uint8)
            ciph = None
            if ciphered(k):
                ciph = np.concatenate([a5_1(kc, [fn + b], 96) for b in range(4)], axis=1)
            e = facch3_encode(l2, sb, ciph)                   # (1, 4, 104)
            ok = True
            for b in range(4):
                ok &= put(k + b, map_symbols(fmt_facch, e[:, b, :], sync_id=facch_sid)[0])
            if ok:
                sent.append(dict(type="facch3", fn=fn, k=k, l2=l2[0], sync_id=facch_sid, ciph=ciph is not None))
            facch_sid ^= 1
            k += 4
    if cfo_hz:
        n = np.arange(n_samples, dtype=np.float64)
        x *= np.exp(1j * (2 * np.pi * cfo_hz / (SYM_RATE * sps)) * n).astype(np.complex64)
    return x, sent


# ---------------------------------------------------------------------------
# AMBE speech frames (the vocoder's input): bit layout of reference src/codec/frame.c:56-75
# ---------------------------------------------------------------------------
AMBE_LAYOUT = {   # field -> [(first bit, bits)], most significant part first
    "pitch": [(0, 7)], "gain": [(7, 6), (50, 2)], "vuv": [(13, 6)], "prba12": [(19, 6), (52, 1)],
    "prba34": [(25, 3), (53, 3)], "prba57": [(28, 3), (56, 4)], "hoc0": [(31, 3), (60, 4)],
    "hoc1": [(34, 3), (64, 3)], "hoc2": [(37, 2), (67, 4)], "hoc3": [(39, 2), (71, 3)],
    "perr14": [(41, 3), (74, 3)], "perr58": [(44, 2), (77, 3)], "mag_rule": [(46, 2)], "pitch_rule": [(48, 2)],
}


def ambe_pack(fields, shape):
    """fields: name -> integer array of `shape` -> frames shape + (10,) uint8."""
    bits = np.zeros(tuple(shape) + (80,), np.uint8)
    for name, parts in AMBE_LAYOUT.items():
        v = np.asarray(fields.get(name, 0), np.int64) + np.zeros(shape, np.int64)
        left = sum(n for _, n in parts)
        for first, n in parts:
            left -= n
            for k in range(n):
                bits[..., first + k] = (v >> (left + n - 1 - k)) & 1
    return np.packbits(bits, axis=-1)


def ambe_speech_frames(n_ch, n_frames, seed=0):
    """(n_ch, n_frames, 10) speech frames: pitch and gain wander slowly and the pitch repeats now and then (voiced
    speech keeps its pitch for a few frames), every other quantiser index is uniform.  The first frame of a channel
    does not ask for pitch interpolation (there is nothing to interpolate from)."""
    rng = np.random.default_rng(seed)
    shape = (n_ch, n_frames)
    step = rng.integers(-6, 7, shape)
    step[rng.random(shape) < 0.3] = 0
    start = rng.integers(10, 110, (n_ch, 1))
    pitch = np.clip(start + np.cumsum(step, axis=1), 0, 123)       # 124..127 mark silence / tone frames
    gain = np.clip(rng.integers(60, 200, (n_ch, 1)) + np.cumsum(rng.integers(-25, 26, shape), axis=1), 0, 255)
    f = {k: rng.integers(0, 1 << sum(n for _, n in parts), shape) for k, parts in AMBE_LAYOUT.items()}
    f["pitch"], f["gain"] = pitch, gain
    f["pitch_rule"][:, 0] = 0
    return ambe_pack(f, shape)
