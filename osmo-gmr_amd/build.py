"""Build csrc/ into libgmr1_hip.so for gfx950 with hipcc (in-tree, no JIT cache).

hipcc cross-compiles without a GPU, so this runs in the CPU-only container as
the "does it build" check and the resulting .so travels to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgmr1_hip.so")
ARCH = "gfx950"

HIP_SOURCES = ["rx_kernels.hip", "fcch_kernels.hip", "l1_kernels.hip", "tch_kernels.hip", "chan_kernels.hip", "nt9_kernels.hip", "xch_kernels.hip", "tx_kernels.hip", "ambe_kernels.hip", "util_kernels.hip"]
CXX_SOURCES = ["capi.cpp", "capi_fcch.cpp", "capi_l1.cpp", "capi_detect.cpp", "capi_rx.cpp", "capi_tch.cpp", "capi_chan.cpp", "capi_nt9.cpp", "capi_xch.cpp", "capi_tx.cpp", "host_tables.cpp", "l1_tables.cpp", "l1_punct.cpp", "capi_shard.cpp", "capi_ambe.cpp", "ambe_tables.cpp"]

COMMON = [
    "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
    "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def sources():
    out = []
    for s in HIP_SOURCES + CXX_SOURCES:
        p = os.path.join(CSRC, s)
        if os.path.exists(p):
            out.append(p)
    return out


def _lib_stamp(lib: str) -> str:
    return lib + ".flags"


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    # built by another compiler or with other flags: rebuild even if no source changed
    import hashlib
    want = hashlib.sha256(_toolchain_id().encode()).hexdigest()
    stamp = _lib_stamp(LIB)
    if not os.path.exists(stamp) or open(stamp).read().strip() != want:
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    for d, _, files in os.walk(os.path.join(ROOT, "include")):
        deps += [os.path.join(d, f) for f in files]
    return any(os.path.getmtime(p) > t for p in deps)


PROFILE_LIB = os.path.join(HERE, "libgmr1_hip_prof.so")


def build_variant(name: str, defines, profile: bool = True, verbose: bool = False) -> str:
    """An experimental library libgmr1_hip_<name>.so: the same sources with extra -D switches (and, by default, the profiling
    build's), for A/B runs on one GPU box through GMR1_HIP_LIBRARY (tools/exp).  Never loaded by the product."""
    lib = os.path.join(HERE, "libgmr1_hip_%s.so" % name)
    return _build(lib, (["-DGMR1_HIP_PROFILE"] if profile else []) + list(defines), ".%s.o" % name, verbose, False)


def build(force: bool = False, verbose: bool = False, profile: bool = False) -> str:
    """profile=True: the same sources with -DGMR1_HIP_PROFILE into libgmr1_hip_prof.so -- the only build in which the
    GMR1_HIP_DBG_STOP / _AMBE_DBG / _RX_IMPL / ... switches exist (tools/ load it through GMR1_HIP_LIBRARY)."""
    if profile:
        return _build(PROFILE_LIB, ["-DGMR1_HIP_PROFILE"], ".prof.o", verbose, force)
    if not force and not needs_build():
        return LIB
    return _build(LIB, [], ".o", verbose, force)


def _toolchain_id() -> str:
    """What an object file depends on besides its sources: the compiler and the command line (kept next to each object)."""
    try:
        ver = subprocess.run([hipcc(), "--version"], capture_output=True, text=True).stdout
    except OSError:
        ver = ""
    return ver + "\n" + " ".join(COMMON) + "\n" + ARCH


def _build(lib: str, extra, suffix: str, verbose: bool, force: bool = False) -> str:
    LIB = lib
    from concurrent.futures import ThreadPoolExecutor
    import hashlib
    tool = _toolchain_id()

    def compile_one(src):
        obj = os.path.splitext(src)[0] + suffix
        cmd = [hipcc()] + COMMON + extra + ["--offload-arch=" + ARCH, "-c", src, "-o", obj]
        if src.endswith(".hip"):
            cmd.insert(1, "-xhip")
        # an object is reused only if it is newer than its source and every header AND was made by this compiler with this
        # command line (the .flags file next to it); force rebuilds everything
        stamp = obj + ".flags"
        want = hashlib.sha256((tool + "\n" + " ".join(cmd)).encode()).hexdigest()
        have = open(stamp).read().strip() if os.path.exists(stamp) else ""
        if (not force and have == want and os.path.exists(obj)
                and os.path.getmtime(obj) > max(os.path.getmtime(d) for d in [src] + headers)):
            return obj
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        with open(stamp, "w") as fh:
            fh.write(want + "\n")
        return obj

    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    for d, _, files in os.walk(os.path.join(ROOT, "include")):
        headers += [os.path.join(d, f) for f in files]
    with ThreadPoolExecutor(max(1, min(6, os.cpu_count() or 1))) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(_lib_stamp(LIB), "w") as fh:
        fh.write(hashlib.sha256(tool.encode()).hexdigest() + "\n")
    return LIB


if __name__ == "__main__":
    import sys
    if "--variant" in sys.argv:
        # python build.py --variant w5 -DGMR1_EXP_RX4_WAVES=5 [--no-profile]
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], [x for x in sys.argv[1:] if x.startswith("-D")], profile="--no-profile" not in sys.argv, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True, profile="--profile" in sys.argv))
