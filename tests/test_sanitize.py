"""The host side under AddressSanitizer + UndefinedBehaviourSanitizer (the reference has `--enable-sanitize`,
configure.ac:41-51; GPU sanitizers are not available on the pool, so this is the CPU build only):

  * the oracle (oracle/*.c) built with -fsanitize=address,undefined and driven through the same Python entry points the
    parity tests use -- every channel coder / decoder in both Viterbi modes, demodulator, detector, FCCH acquisition, the
    receive loop with its traffic follow-ups, NT9 / xCH / RACH, the AMBE decoder;
  * the library's host-only translation units (burst tables + flattening, code / puncturing descriptions,
    gmr1_puncturer_generate, FCCH tables) and the receive loop's control logic (rx_loop.h) as C++ programs.

A report of either sanitizer aborts the child process, which fails the test with the report."""
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]

pytestmark = pytest.mark.skipif(shutil.which("gcc") is None or shutil.which("g++") is None, reason="no gcc / g++")


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip("libasan.so not installed")
    return p


def test_oracle_under_asan_ubsan(tmp_path):
    asan = _libasan()
    srcs = [os.path.join(ROOT, "oracle", f) for f in sorted(os.listdir(os.path.join(ROOT, "oracle"))) if f.startswith("orc_") and f.endswith(".c")]
    lib = str(tmp_path / "liborc_san.so")
    r = subprocess.run(["gcc", "-std=gnu99", "-Wall", "-fPIC", "-ffp-contract=off", "-shared", "-o", lib] + SAN + srcs + ["-lm"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the workout: the oracle's own CPU test files (golden vectors, round trips, D1 / D1b, xCH / RACH, receive loop, AMBE)
    # with tests/oracle_lib.py pointed at the sanitized build
    env = dict(os.environ, LD_PRELOAD=asan, ORC_LIBRARY=lib,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    files = ["test_oracle.py", "test_oracle_xch.py", "test_oracle_d1b.py", "test_oracle_rx.py", "test_oracle_3p.py",
             "test_oracle_ambe.py", "test_pin_kit.py"]
    files = [os.path.join(ROOT, "tests", f) for f in files if os.path.exists(os.path.join(ROOT, "tests", f))]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + files, capture_output=True, text=True,
                       env=env, cwd=ROOT, timeout=1500)
    tail = (r.stdout + r.stderr)[-4000:]
    assert "runtime error" not in tail and "AddressSanitizer" not in tail, tail
    assert r.returncode == 0, tail


def test_host_translation_units_under_asan_ubsan(tmp_path):
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "osmo-gmr_amd", "csrc"), "-I/opt/rocm/include",
           "-D__HIP_PLATFORM_AMD__"]
    csrc = os.path.join(ROOT, "osmo-gmr_amd", "csrc")
    exe = str(tmp_path / "host_san")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-o", exe] + SAN + inc +
                       [os.path.join(ROOT, "tests", "c", "host_san.cpp"), os.path.join(csrc, "host_tables.cpp"),
                        os.path.join(csrc, "l1_tables.cpp"), os.path.join(csrc, "l1_punct.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "host_san: 0 failures" in r.stdout, (r.stdout + r.stderr)[-3000:]
    # the receive loop's control logic walked through whole captures (rx_loop.h, host build)
    exe2 = str(tmp_path / "rx_loop_san")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-o", exe2] + SAN +
                       ["-I" + csrc, "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "rx_loop_host.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    for args in ((4, 5616000, 8000, 0, 0, 0, 1), (4, 400000, 8000, 37, 5, 11, -2), (8, 900000, 100, 7, 3, 23, 3), (5, 20000, 8000, 0, 0, 0, 0)):
        r = subprocess.run([exe2] + [str(a) for a in args], capture_output=True, text=True, env=env)
        assert r.returncode == 0, (args, (r.stdout + r.stderr)[-2000:])
