#!/usr/bin/env python3
"""profiles/valu_<kernel>.json from a tools/pmc_rx4.sh / tools/pmc_nt3.sh summary: what a kernel's vector ALUs were doing,
MACHINE-WIDE, keyed on the hash of the sources the kernel is compiled from (bench.py reports it as `roofline_valu` only
while that hash matches).

    python3 tools/valu_summary.py <summary.txt> <tag> [--kernel k_rx4] [--kernel-ms 0.2438] [--clock-mhz 2400]

The counters are sums over the launch (mean over the launches of the run):
    kernel quad-cycles   Q = kernel duration x shader clock / 4        (SQ_* cycle counters tick once per four shader cycles)
    avg_resident_waves     = sum SQ_WAVE_CYCLES      / (SIMDs x Q)     waves resident per SIMD, averaged over the launch
    valu_busy              = sum SQ_ACTIVE_INST_VALU / (SIMDs x Q)     share of the launch a SIMD's vector ALU is executing
Neither can exceed its bound (the occupancy the kernel is compiled for; 1).  The duration is the summary's own
`avg_duration_ns` line (the kernel trace of the counter run) unless --kernel-ms is given.  Round 5's form multiplied the
per-wave ratio by a CONSTANT number of resident waves and overstated the busy share wherever fewer were resident on
average (ramp, drain): 0.87 for the headline kernel where the machine-wide figure is 0.74."""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

ap = argparse.ArgumentParser()
ap.add_argument("src")
ap.add_argument("tag")
ap.add_argument("--kernel", default="k_rx4<", help="substring of the kernel's name in the summary")
ap.add_argument("--out", default=None, help="file name under profiles/ (default: valu_<kernel>.json)")
ap.add_argument("--kernel-ms", type=float, default=None)
ap.add_argument("--clock-mhz", type=float, default=2400.0)
ap.add_argument("--simds", type=int, default=1024, help="256 CUs x 4")
ap.add_argument("--waves-per-simd", type=int, default=6, help="what the kernel is compiled for (the bound of avg_resident_waves)")
args = ap.parse_args()

total, per_wave, dur_ns, waves, name = {}, {}, None, None, None
take = False
for line in open(args.src):
    if not line.startswith(" ") and "grid=" in line:
        take = args.kernel in line and name in (None, line.split("  grid=")[0])
        if take:
            name = line.split("  grid=")[0]
            waves = int(re.search(r"waves=(\d+)", line).group(1))
        continue
    if line.startswith("=="):          # a cut-off section of tools/phases.sh: not the whole kernel
        take = False
    if not take:
        continue
    m = re.match(r"\s+(SQ_\w+)\s+([\d.]+)\s+per wave\s+([\d.]+)", line)
    if m:
        total.setdefault(m.group(1), float(m.group(2)))
        per_wave.setdefault(m.group(1), float(m.group(3)))
    m = re.match(r"\s+avg_duration_ns\s+([\d.]+)", line)
    if m and dur_ns is None:
        dur_ns = float(m.group(1))
if args.kernel_ms is not None:
    dur_ns = args.kernel_ms * 1e6
if not total or dur_ns is None:
    sys.exit("no counters / no kernel duration for %r in %s (give --kernel-ms)" % (args.kernel, args.src))
quad = dur_ns * 1e-9 * args.clock_mhz * 1e6 / 4.0
resident = total["SQ_WAVE_CYCLES"] / (args.simds * quad)
busy = total["SQ_ACTIVE_INST_VALU"] / (args.simds * quad)
# (the counter charges every vector instruction one quad-cycle; a kernel of instructions that issue faster than that -- k_tch3's
# packed 16-bit butterflies -- can come out a little above 1: it is then simply saturated, and the file says so)
assert 0.0 < busy <= 1.25 and resident <= args.waves_per_simd * 1.02, (busy, resident)
out = {"_comment": "SQ counters of one launch (mean over the run's launches), machine-wide: valu_busy = sum SQ_ACTIVE_INST_VALU / "
                   "(SIMDs x kernel quad-cycles), avg_resident_waves = sum SQ_WAVE_CYCLES / the same; kernel quad-cycles = duration x "
                   "clock / 4 (tools/valu_summary.py)",
       "tag": args.tag, "source": os.path.relpath(args.src, ROOT), "kernel": name, "waves": waves,
       "kernel_ns": dur_ns, "clock_mhz": args.clock_mhz, "simds": args.simds, "compiled_waves_per_simd": args.waves_per_simd,
       "avg_resident_waves": resident, "valu_busy": busy,
       **({"note": "above 1: SQ_ACTIVE_INST_VALU charges one quad-cycle per instruction, this kernel issues some faster; read as saturated"}
          if busy > 1.0 else {}),
       "valu_insts_per_wave": per_wave.get("SQ_INSTS_VALU"), "salu_insts_per_wave": per_wave.get("SQ_INSTS_SALU"),
       "lds_insts_per_wave": per_wave.get("SQ_INSTS_LDS"), "active_inst_valu_per_wave": per_wave.get("SQ_ACTIVE_INST_VALU"),
       "wave_cycles_per_wave": per_wave.get("SQ_WAVE_CYCLES"),
       "kernel_sources_sha256": bench.kernel_sources_hash()}
fn = args.out or ("valu_%s.json" % re.sub(r"\W+", "_", args.kernel).strip("_"))
json.dump(out, open(os.path.join(ROOT, "profiles", fn), "w"), indent=1)
print(json.dumps(out, indent=1))
