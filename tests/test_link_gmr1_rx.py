"""The reference's application compiles and links UNCHANGED against this library's headers and shared object.

Container only (needs /root/reference).  src/gmr1_rx.c and src/gsmtap.c are compiled where they lie with
`-I include -DGMR1_HIP_USE_SYSTEM_OSMOCOM`; the libosmocore / libosmo-dsp headers they include are absent from this
image, so declaration-only headers (just what those two files use) are written into tmp_path for this census, and
the handful of third-party functions get abort() bodies so that the link can be completed with --no-undefined.
Nothing here is an oracle: the point is that every gmr1_* symbol and type the application needs comes from
include/ + libgmr1_hip.so, with the reference's own signatures."""
import os
import subprocess
import textwrap

import pytest

import tp_headers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "src", "gmr1_rx.c")),
                                reason="/root/reference is not present on this machine")

THIRD_PARTY = {   # what gmr1_rx.c + gsmtap.c take from libosmocore / libosmo-dsp
    "cfile_load", "cfile_release", "osmo_cxvec_init_from_data", "osmo_hexparse", "osmo_hexdump_nospc",
    "gsmtap_source_init", "gsmtap_source_add_sink", "gsmtap_sendmsg", "msgb_alloc", "msgb_free", "msgb_put",
}

STUBS = """
    #include <stdlib.h>
    #define S(name) void name(void) { abort(); }
    S(cfile_load) S(cfile_release) S(osmo_cxvec_init_from_data) S(osmo_hexparse) S(osmo_hexdump_nospc)
    S(gsmtap_source_init) S(gsmtap_source_add_sink) S(gsmtap_sendmsg) S(msgb_alloc) S(msgb_free) S(msgb_put)
"""


def _nm(path, *flags):
    out = subprocess.run(["nm", *flags, path], capture_output=True, text=True, check=True).stdout
    return {l.split()[-1] for l in out.splitlines() if l.strip()}


def test_gmr1_rx_compiles_and_links_unchanged(pkg, tmp_path):
    tp_headers.write(tmp_path / "tp")
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + str(tmp_path / "tp"), "-DGMR1_HIP_USE_SYSTEM_OSMOCOM"]
    objs = []
    for src in ("gmr1_rx.c", "gsmtap.c"):
        o = str(tmp_path / (src[:-2] + ".o"))
        r = subprocess.run(["gcc", "-std=gnu99", "-O2", "-Wall", "-c", os.path.join(REF, "src", src), "-o", o] + inc,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "error" not in r.stderr and "incompatible" not in r.stderr and "implicit" not in r.stderr, r.stderr
        objs.append(o)

    # symbol census: what the two objects leave undefined
    defined = set().union(*[_nm(o, "--defined-only") for o in objs])
    undef = set().union(*[_nm(o, "-u") for o in objs]) - defined
    lib = pkg.build.LIB
    exported = _nm(lib, "-D", "--defined-only")
    gmr1 = {s for s in undef if s.startswith("gmr1_")}
    assert gmr1, "no gmr1_* references found"
    missing = gmr1 - exported
    assert not missing, f"gmr1_rx.c needs {sorted(missing)} which libgmr1_hip.so does not export"
    rest = undef - gmr1 - THIRD_PARTY
    # everything else must be the C library / libm
    libc = set()
    for so in ("libc.so.6", "libm.so.6"):
        for d in ("/lib/x86_64-linux-gnu", "/usr/lib/x86_64-linux-gnu", "/lib64"):
            if os.path.exists(os.path.join(d, so)):
                libc |= {s.split("@")[0] for s in _nm(os.path.join(d, so), "-D", "--defined-only")}
                break
    assert rest <= libc, f"unexpected undefined symbols: {sorted(rest - libc)}"
    assert (undef & THIRD_PARTY) == THIRD_PARTY - {"msgb_put"} or (undef & THIRD_PARTY) == THIRD_PARTY

    # and the link completes: the application + abort() bodies for the third-party calls + libgmr1_hip.so
    stubs = tmp_path / "tp_stubs.c"
    stubs.write_text(textwrap.dedent(STUBS))
    exe = str(tmp_path / "gmr1_rx")
    r = subprocess.run(["gcc", "-o", exe] + objs + [str(stubs), "-Wl,--no-undefined",
                        "-L" + os.path.dirname(lib), "-l:" + os.path.basename(lib),
                        "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-lm"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # it loads and runs up to its argument check (reference gmr1_rx.c:912-915: usage, exit code != 0)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert "Usage:" in r.stderr and "bcch.cfile" in r.stderr


def test_gmr1_ambe_decode_compiles_and_links_unchanged(pkg):
    """The vocoder's program needs nothing third-party: src/gmr1_ambe_decode.c + include/ + libgmr1_hip.so is a
    complete link (--no-undefined), and the only gmr1_* symbols it takes are the three codec calls."""
    import ref_codec
    exe = ref_codec.build_program_on_product()
    assert exe and os.path.exists(exe)
    undef = _nm(exe, "-u")
    gmr1 = {x.split("@")[0] for x in undef if x.startswith("gmr1_")}
    assert gmr1 == {"gmr1_codec_alloc", "gmr1_codec_release", "gmr1_codec_decode_frame"}
    assert gmr1 <= _nm(pkg.build.LIB, "-D", "--defined-only")
