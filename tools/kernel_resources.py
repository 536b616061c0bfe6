#!/usr/bin/env python3
"""Register and LDS budget of every kernel, from the code objects' metadata (no GPU needed):
python tools/kernel_resources.py > profiles/<tag>_kernel_resources.txt
waves/SIMD = min(8, 512 // vgprs rounded up to the allocation granule of 8); dynamic LDS (the burst kernels') is not in the
metadata - DESIGN.md gives it per kernel."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "osmo-gmr_amd", "csrc")


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.split("\n")


def main():
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for f in sorted(glob.glob(os.path.join(SRC, "*.hip"))):
            out = os.path.join(td, os.path.basename(f) + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", "-xhip", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function",
                            "-I" + os.path.join(ROOT, "include"), "-I" + SRC, "--offload-arch=gfx950", "--cuda-device-only", "-S",
                            f, "-o", out], check=True, stderr=subprocess.DEVNULL)
            s = open(out).read()
            md = s[s.index("amdhsa.kernels:"):]
            for blk in md.split("  - .agpr_count:")[1:]:
                g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
                rows.append((os.path.basename(f), g("name"), int(g("vgpr_count")), int(g("sgpr_count")),
                             int(g("group_segment_fixed_size")), int(g("vgpr_spill_count")), int(g("sgpr_spill_count"))))
    names = demangle([r[1] for r in rows])
    print(f"{'file':18s} {'vgpr':>4s} {'w/SIMD':>6s} {'sgpr':>4s} {'LDS(static)':>11s} {'vspill':>6s} {'sspill':>6s}  kernel")
    for r, n in zip(rows, names):
        v = (r[2] + 7) // 8 * 8
        n = re.sub(r"\(.*", "", n.replace("void ", "").replace("gmr1::", ""))
        print(f"{r[0]:18s} {r[2]:4d} {min(8, 512 // max(v, 1)):6d} {r[3]:4d} {r[4]:11d} {r[5]:6d} {r[6]:6d}  {n}")


if __name__ == "__main__":
    sys.exit(main())
