"""Streams of 10-byte AMBE frames for the vocoder tests -- TEST INFRASTRUCTURE (numpy only, no oracle, no product).

The bit layout is the one the reference unpacks (src/codec/frame.c:56-75), written here the other way round."""
from __future__ import annotations

import numpy as np

# field -> [(first bit, bits)] MSB part first
LAYOUT = {
    "pitch": [(0, 7)], "gain": [(7, 6), (50, 2)], "vuv": [(13, 6)], "prba12": [(19, 6), (52, 1)],
    "prba34": [(25, 3), (53, 3)], "prba57": [(28, 3), (56, 4)], "hoc0": [(31, 3), (60, 4)],
    "hoc1": [(34, 3), (64, 3)], "hoc2": [(37, 2), (67, 4)], "hoc3": [(39, 2), (71, 3)],
    "perr14": [(41, 3), (74, 3)], "perr58": [(44, 2), (77, 3)], "mag_rule": [(46, 2)], "pitch_rule": [(48, 2)],
}
WIDTH = {k: sum(n for _, n in v) for k, v in LAYOUT.items()}
ORDER = ["pitch", "pitch_rule", "gain", "vuv", "prba12", "prba34", "prba57", "hoc0", "hoc1", "hoc2", "hoc3",
         "mag_rule", "perr14", "perr58"]          # the order orc_ambe_unpack reports


def pack(**fields):
    """One speech frame from its quantiser indices (missing fields = 0)."""
    bits = np.zeros(80, np.uint8)
    for name, parts in LAYOUT.items():
        v = int(fields.get(name, 0))
        assert 0 <= v < (1 << WIDTH[name]), (name, v)
        left = WIDTH[name]
        for first, n in parts:
            left -= n
            piece = (v >> left) & ((1 << n) - 1)
            for k in range(n):
                bits[first + k] = (piece >> (n - 1 - k)) & 1
    return np.packbits(bits)


def is_speech(frame):
    return (int(frame[0]) & 0xfc) not in (0xfc, 0xf8)


def silence_frame():
    f = np.zeros(10, np.uint8)
    f[0] = 0xf8
    return f


def tone_frame(code, log_ampl=200, sel=3):
    """Tone frame: type in byte 0, amplitude in byte 1, the 8-bit tone code by majority over bytes 0..7
    (src/codec/tone.c:121-133): six copies of the code out-vote the two fixed bytes."""
    f = np.zeros(10, np.uint8)
    f[0] = 0xfc | (sel & 3)
    f[1] = log_ampl
    f[2:8] = code
    return f


def first_speech_without_interpolation(frames):
    """The reference indexes past its arrays when a stream's first speech frame asks for pitch interpolation
    (decision D10): streams meant for comparison with it start with rule 0."""
    frames = frames.copy()
    for f in frames:
        if is_speech(f):
            f[6] &= 0x3f
            break
    return frames


def primed(frames, seed=0):
    """The stream with one speech frame of the lowest pitches in front: 56 harmonics in both subframes, so every
    per-harmonic voicing entry of the decoder has been written once and nothing read later predates the stream
    (decision D9: the reference's program otherwise shows what its stack held at start-up in the first frames)."""
    rng = np.random.default_rng(seed)
    f = {k: int(rng.integers(0, 1 << w)) for k, w in WIDTH.items()}
    f.update(pitch=122, pitch_rule=0)
    return np.concatenate([pack(**f)[None, :], frames])


def random_stream(n, seed, speech_only=False):
    rng = np.random.default_rng(seed)
    fr = rng.integers(0, 256, (n, 10), dtype=np.uint8)
    if speech_only:
        reserved = (fr[:, 0] & 0xf8) == 0xf8
        fr[reserved, 0] &= 0x7f
    return first_speech_without_interpolation(fr)


def speech_like(n, seed):
    """Pitch and gain wander slowly, pitch repeats now and then (the equal-pitch branch of the interpolation),
    spectral indices are random."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 10), np.uint8)
    pitch, gain = int(rng.integers(10, 110)), int(rng.integers(60, 200))
    for i in range(n):
        if rng.random() > 0.3:
            pitch = int(np.clip(pitch + rng.integers(-6, 7), 0, 123))
        gain = int(np.clip(gain + rng.integers(-25, 26), 0, 255))
        f = {k: int(rng.integers(0, 1 << w)) for k, w in WIDTH.items()}
        f.update(pitch=pitch, gain=gain)
        if i == 0:
            f["pitch_rule"] = 0
        out[i] = pack(**f)
    return out


def mixed_stream(n, seed, invalid_tones=False):
    """Speech with runs of silence frames and of every family of tone frame in between."""
    rng = np.random.default_rng(seed)
    out = speech_like(n, seed)
    codes = list(range(0x80, 0xa4)) + [0xff, 0x00, 0x05, 0x20, 0x7e]
    if invalid_tones:
        codes += [0x7f, 0xa4, 0xf0]
    i = int(rng.integers(5, 20))
    while i < n - 8:
        kind = rng.integers(0, 3)
        run = int(rng.integers(1, 6))
        for k in range(run):
            if kind == 0:
                out[i + k] = silence_frame()
            else:
                out[i + k] = tone_frame(codes[int(rng.integers(0, len(codes)))], int(rng.integers(120, 256)),
                                        int(rng.integers(0, 4)))
        i += run + int(rng.integers(5, 40))
    return out
