#!/usr/bin/env python3
"""Microseconds per (gmr1_pi4cxpsk_demod, gmr1_bcch_decode / gmr1_ccch_decode) pair through the reference's own
one-burst API, from a C program shaped like gmr1_rx.c's loop (tools/legacy_loop.c), next to the CPU oracle making the
same calls; what comes back is compared with the batch entry point.  GPU box, repo root."""
import json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import workloads, oracle_lib
pkg = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1300            # about one carrier-minute of BCCH + CCCH bursts
wl = workloads.bcch_ccch_mix(pkg, n=n, seed=11)
d = tempfile.mkdtemp()
path = os.path.join(d, "bursts.bin")
with open(path, "wb") as f:
    f.write(np.int32(n).tobytes())
    for i in range(n):
        ln = 1016 if wl["kind"][i] == 0 else 976
        o = int(wl["offset"][i])
        f.write(np.int32(wl["kind"][i]).tobytes()); f.write(np.int32(ln).tobytes()); f.write(wl["iq"][o:o + ln].tobytes())
exe = os.path.join(d, "legacy_loop")
lib = pkg.build.LIB
subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "legacy_loop.c"),
                       "-o", exe, "-L" + os.path.dirname(lib), "-l:" + os.path.basename(lib), "-Wl,-rpath," + os.path.dirname(lib),
                       "-Wl,-rpath,/opt/rocm/lib"])
size0 = os.path.getsize(path)
out = json.loads(subprocess.run([exe, path, "4"], capture_output=True, text=True, check=True).stdout)
raw = np.fromfile(path, np.uint8)[size0:]
crc = raw[:4 * n].view(np.int32); l2 = raw[4 * n:].reshape(n, 24)
# the same through the batch entry point, and the CPU oracle call by call
api = pkg.api; api.load(); api.init(0)
b = api.rx_bcch_ccch_batch(wl["iq"], wl["offset"], wl["kind"], sps=4, want_ssyms=False)
ok = (crc == 0) | (b["crc"] == 0)
out["identical_to_batch_entry_point"] = bool(np.array_equal(crc, b["crc"]) and np.array_equal(l2[ok], b["l2"][ok]))
oracle_lib.lib()
t0 = time.perf_counter()
ref = oracle_lib.demod_decode_batch(wl["iq"], wl["offset"], wl["kind"], sps=4, want_ebits=False, want_ssyms=False)
out["oracle_us_per_pair"] = (time.perf_counter() - t0) / n * 1e6
out["identical_to_oracle"] = bool(np.array_equal(crc, ref["crc"]) and np.array_equal(l2[(crc == 0) | (ref["crc"] == 0)], ref["l2"][(crc == 0) | (ref["crc"] == 0)]))
out["carrier_minute_ms"] = out["us_per_pair"] * 1300 / 1e3
print(json.dumps(out))
