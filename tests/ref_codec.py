"""The reference's own AMBE vocoder, compiled from its sources where they lie (container only) -- TEST INFRASTRUCTURE.

`make -C oracle ref` builds oracle/_ref/libgmr1_codec_ref.so (src/codec + oracle/ref_codec_shim.c) and this module
also builds oracle/_ref/gmr1_ambe_decode, the reference's file-to-file program, from src/gmr1_ambe_decode.c.  Both are
git-ignored build products; on the GPU box (no /root/reference) the prebuilt files are used if they travelled, and
the committed outputs under tests/golden/ otherwise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
LIB = os.path.join(REF_DIR, "libgmr1_codec_ref.so")
TOOL = os.path.join(REF_DIR, "gmr1_ambe_decode")
TOOL_HIP = os.path.join(REF_DIR, "gmr1_ambe_decode_hip")      # the same main(), linked against libgmr1_hip.so instead
HIP_LIB = os.path.join(ROOT, "osmo-gmr_amd", "libgmr1_hip.so")
CODEC_FILES = ["ambe", "codec", "frame", "math", "synth", "tables", "tone"]


def sources_present():
    return os.path.isfile(os.path.join(REF, "src", "codec", "ambe.c"))


def build():
    """Compiles the reference (when its sources are here).  Returns True if the library and the program exist."""
    if sources_present():
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
        srcs = [os.path.join(REF, "src", "gmr1_ambe_decode.c")] + [
            os.path.join(REF, "src", "codec", f + ".c") for f in CODEC_FILES]
        if not os.path.exists(TOOL) or any(os.path.getmtime(s) > os.path.getmtime(TOOL) for s in srcs):
            subprocess.check_call(["gcc", "-O2", "-g", "-I" + os.path.join(REF, "include"), "-o", TOOL] + srcs + ["-lm"])
        build_program_on_product()
    return os.path.exists(LIB) and os.path.exists(TOOL)


def build_program_on_product():
    """The reference's src/gmr1_ambe_decode.c, unchanged, compiled against THIS repo's include/ and linked against
    libgmr1_hip.so (no other object): the drop-in check for the codec calls.  Returns the path or None."""
    main_c = os.path.join(REF, "src", "gmr1_ambe_decode.c")
    if not (os.path.isfile(main_c) and os.path.exists(HIP_LIB)):
        return TOOL_HIP if os.path.exists(TOOL_HIP) else None
    if not os.path.exists(TOOL_HIP) or max(os.path.getmtime(main_c), os.path.getmtime(HIP_LIB)) > os.path.getmtime(TOOL_HIP):
        os.makedirs(REF_DIR, exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror=implicit-function-declaration",
                               "-I" + os.path.join(ROOT, "include"), "-o", TOOL_HIP, main_c,
                               "-Wl,--no-undefined", "-L" + os.path.dirname(HIP_LIB), "-l:" + os.path.basename(HIP_LIB),
                               "-Wl,-rpath,$ORIGIN/../../osmo-gmr_amd"])
    return TOOL_HIP


def available():
    try:
        return build()
    except (subprocess.CalledProcessError, OSError):
        return False


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB)
        _lib.gmr1_codec_alloc.restype = C.c_void_p
    return _lib


def decode_clean_stack(frames):
    """The library entered on a zeroed stack for every frame (ref_codec_shim.c) -> (pcm [n, 160], rv [n])."""
    frames = np.ascontiguousarray(frames, np.uint8).reshape(-1, 10)
    n = len(frames)
    pcm = np.zeros((n, 160), np.int16)
    rv = np.zeros(n, np.int32)
    c = C.c_void_p(lib().gmr1_codec_alloc())
    lib().ref_codec_decode_stream(c, frames.ctypes.data_as(C.c_void_p), C.c_int(n), pcm.ctypes.data_as(C.c_void_p),
                                  rv.ctypes.data_as(C.c_void_p))
    lib().gmr1_codec_release(c)
    return pcm, rv


def decode_with_program(frames, tool=None, wav=False):
    """The reference's program on a file of frames -> pcm [m, 160] (m < n if it stopped at a frame it rejects)."""
    frames = np.ascontiguousarray(frames, np.uint8).reshape(-1, 10)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.dat"), os.path.join(d, "out.wav" if wav else "out.raw")
        frames.tofile(fin)
        subprocess.run([tool or TOOL, fin, fout], check=True, stderr=subprocess.DEVNULL, timeout=300)
        raw = np.fromfile(fout, np.uint8)
        if wav:
            raw = raw[44:]
        return raw.view(np.int16).reshape(-1, 160)
