"""The reference's application src/gmr1_rx.c (+ src/gsmtap.c), UNCHANGED, built against this repo's include/ and
libgmr1_hip.so and made runnable -- TEST INFRASTRUCTURE (container only: needs /root/reference).

The program's only other dependencies are eleven libosmocore / libosmo-dsp functions that load a file, hold a message
buffer and send a GSMTAP datagram; tests/c/tp_shim_gmr1_rx.c stands in for them (the "datagrams" go to a file) and
tests/tp_headers.py declares them.  The product of `build()` is oracle/_ref/gmr1_rx_hip: git-ignored, it travels to
the GPU box with the snapshot like the library itself, where tests/test_gpu_ref_program.py runs it on a synthetic
capture.  Every gmr1_* call it makes resolves in libgmr1_hip.so (--no-undefined)."""
from __future__ import annotations

import os
import struct
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
EXE = os.path.join(REF_DIR, "gmr1_rx_hip")
HIP_LIB = os.path.join(ROOT, "osmo-gmr_amd", "libgmr1_hip.so")
SHIM = os.path.join(ROOT, "tests", "c", "tp_shim_gmr1_rx.c")
SOURCES = [os.path.join(REF, "src", "gmr1_rx.c"), os.path.join(REF, "src", "gsmtap.c")]


def sources_present():
    return all(os.path.isfile(s) for s in SOURCES)


def build():
    """Returns the executable's path, or None when it neither exists nor can be built here."""
    if not (sources_present() and os.path.exists(HIP_LIB)):
        return EXE if os.path.exists(EXE) else None
    deps = SOURCES + [SHIM, HIP_LIB, os.path.join(ROOT, "tests", "tp_headers.py")]
    if os.path.exists(EXE) and all(os.path.getmtime(d) <= os.path.getmtime(EXE) for d in deps):
        return EXE
    import tp_headers
    os.makedirs(REF_DIR, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        inc = tp_headers.write(os.path.join(tmp, "tp"))
        subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-Wall", "-Werror=implicit-function-declaration",
                               "-DGMR1_HIP_USE_SYSTEM_OSMOCOM", "-I" + os.path.join(ROOT, "include"), "-I" + inc,
                               "-o", EXE] + SOURCES + [SHIM, "-Wl,--no-undefined", "-L" + os.path.dirname(HIP_LIB),
                               "-l:" + os.path.basename(HIP_LIB), "-Wl,-rpath,$ORIGIN/../../osmo-gmr_amd",
                               "-Wl,-rpath,/opt/rocm/lib", "-lm"])
    return EXE


def run(iq, sps=4, timeout=600, env=None):
    """Write `iq` (complex64) as a cfile, run `gmr1_rx_hip sps file`, return (exit code, [(sub_type, fn, tn, l2 bytes)],
    stderr) -- the GSMTAP messages in the order the program sent them (header layout: src/gsmtap.c:55-65)."""
    import numpy as np
    with tempfile.TemporaryDirectory() as tmp:
        cf = os.path.join(tmp, "bcch.cfile")
        out = os.path.join(tmp, "gsmtap.bin")
        np.ascontiguousarray(iq, np.complex64).tofile(cf)
        e = dict(os.environ, GMR1_TEST_GSMTAP_OUT=out)
        e.update(env or {})
        r = subprocess.run([EXE, str(sps), cf], capture_output=True, text=True, timeout=timeout, env=e)
        msgs = []
        if os.path.exists(out):
            raw = open(out, "rb").read()
            o = 0
            while o + 4 <= len(raw):
                (n,) = struct.unpack_from("<I", raw, o)
                m = raw[o + 4:o + 4 + n]
                o += 4 + n
                # struct gsmtap_hdr: version, hdr_len, type, timeslot, arfcn:16, signal_dbm, snr_db, frame_number:32 (network
                # order), sub_type, antenna_nr, sub_slot, res
                version, hdr_len, typ, tn = m[0], m[1], m[2], m[3]
                fn = struct.unpack_from(">I", m, 8)[0]
                msgs.append((m[12], fn, tn, bytes(m[4 * hdr_len:]), (version, typ)))
        return r.returncode, msgs, r.stderr
