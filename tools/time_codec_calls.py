#!/usr/bin/env python3
"""Microseconds per gmr1_codec_decode_frame call (the reference's one-frame API on this library) next to the CPU
oracle's and, when oracle/_ref was built, the reference's own library.  GPU box, repo root: python3 tools/time_codec_calls.py"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package   # noqa: E402
import oracle_lib                           # noqa: E402
import ref_codec                            # noqa: E402

pkg = load_package()
api = pkg.api
api.load()
api.init(0)
fr = pkg.synth.ambe_speech_frames(1, 1000, seed=4)[0]
lib = api.load()
lib.gmr1_codec_alloc.restype = C.c_void_p
c = C.c_void_p(lib.gmr1_codec_alloc())
audio = np.zeros(160, np.int16)
pa = audio.ctypes.data_as(C.c_void_p)
got = np.zeros((len(fr), 160), np.int16)
for i in range(20):
    lib.gmr1_codec_decode_frame(c, pa, 160, fr[i].ctypes.data_as(C.c_void_p), 0)
lib.gmr1_codec_release(c)
c = C.c_void_p(lib.gmr1_codec_alloc())
ptrs = [f.ctypes.data_as(C.c_void_p) for f in fr]
t = time.perf_counter()
for i in range(len(fr)):
    lib.gmr1_codec_decode_frame(c, pa, 160, ptrs[i], 0)
    got[i] = audio
t_gpu = (time.perf_counter() - t) / len(fr)
lib.gmr1_codec_release(c)

d = oracle_lib.AmbeDecoder()
ol = oracle_lib.lib()
t = time.perf_counter()
want = np.zeros_like(got)
for i in range(len(fr)):
    ol.orc_ambe_decode_frame(d.buf, pa, 160, ptrs[i], 0)
    want[i] = audio
t_orc = (time.perf_counter() - t) / len(fr)
out = {"frames": len(fr), "us_per_call_gpu": round(t_gpu * 1e6, 1), "us_per_call_oracle": round(t_orc * 1e6, 1),
       "samples_differing_from_oracle": int((got != want).sum()), "max_abs_diff": int(np.abs(got.astype(int) - want).max())}
if os.path.exists(ref_codec.LIB):
    rl = ref_codec.lib()
    rc = C.c_void_p(rl.gmr1_codec_alloc())
    t = time.perf_counter()
    for i in range(len(fr)):
        rl.gmr1_codec_decode_frame(rc, pa, 160, ptrs[i], 0)
    out["us_per_call_reference"] = round((time.perf_counter() - t) / len(fr) * 1e6, 1)
    rl.gmr1_codec_release(rc)
print(json.dumps(out))
