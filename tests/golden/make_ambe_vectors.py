#!/usr/bin/env python3
"""Writes tests/golden/ambe_vectors.npz: frames and the PCM the REFERENCE ITSELF gives for them (container only).

The reference's vocoder (src/codec, src/gmr1_ambe_decode.c) is compiled from its sources where they lie into
oracle/_ref/ (tests/ref_codec.py, oracle/Makefile `ref`); nothing of it is copied.  Per stream:
    <name>_frames  [n, 10] uint8     input
    <name>_pcm     [n, 160] int16    output of the reference's program gmr1_ambe_decode on that file
    <name>_clean   [n, 160] int16    output of gmr1_codec_decode_frame entered on a zeroed stack (decision D9's other reading)
    <name>_rv      [n] int32         return values of the library call
The program stops at a frame the library rejects, so streams with invalid tone codes carry only _clean / _rv.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ambe_streams as S     # noqa: E402
import ref_codec             # noqa: E402

if not ref_codec.available():
    sys.exit("needs /root/reference (container only)")

streams = {
    # primed: the program's output is then a function of the stream alone (ambe_streams.primed)
    "random": S.primed(S.random_stream(400, 11, speech_only=True)),
    "speech": S.primed(S.speech_like(400, 12)),
    "mixed": S.primed(S.mixed_stream(400, 13)),
    "unprimed": S.speech_like(200, 15),
    "mixed_invalid": S.mixed_stream(200, 14, invalid_tones=True),
}
out = {}
for name, fr in streams.items():
    clean, rv = ref_codec.decode_clean_stack(fr)
    out[name + "_frames"] = fr
    out[name + "_clean"] = clean
    out[name + "_rv"] = rv
    if (rv == 0).all():
        pcm = ref_codec.decode_with_program(fr)
        assert len(pcm) == len(fr)
        out[name + "_pcm"] = pcm
    print(name, len(fr), "frames; rejected:", int((rv != 0).sum()),
          "program != clean-stack frames:", int((out.get(name + "_pcm", clean) != clean).any(1).sum()))
np.savez_compressed(os.path.join(HERE, "ambe_vectors.npz"), **out)
print("wrote", os.path.join(HERE, "ambe_vectors.npz"), os.path.getsize(os.path.join(HERE, "ambe_vectors.npz")), "bytes")
