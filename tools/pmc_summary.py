#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: per kernel, mean counter value per dispatch (and per wave)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
match = sys.argv[2] if len(sys.argv) > 2 else ""
files = glob.glob(root + "/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(list))
grid = {}
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if match and match not in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        grid[k] = (int(r["Grid_Size"]), int(r["Workgroup_Size"]))
# kernel durations of the same run (rocprofv3 --kernel-trace writes them beside the counters)
dur = defaultdict(list)
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if match and match not in k:
            continue
        try:
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        except (KeyError, ValueError):
            pass
for k, cs in acc.items():
    g, wg = grid[k]
    waves = g // 64
    print(f"{k[:80]}  grid={g} wg={wg} waves={waves}")
    if dur.get(k):
        d = sorted(dur[k])
        print(f"  {'avg_duration_ns':28s} {sum(d) / len(d):16.1f}  median {d[len(d) // 2]:10.1f}  (n={len(d)}, under the counter collection)")
    for c, v in sorted(cs.items()):
        m = sum(v) / len(v)
        print(f"  {c:28s} {m:16.1f}  per wave {m / max(waves, 1):10.1f}  (n={len(v)})")
