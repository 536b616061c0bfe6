"""Print one timed receive-loop step's acquisition chain (kernels and copies, begin / end relative to the chain's first op)
from rocprofv3's kernel and memory-copy traces."""
import csv, glob, sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
rows.sort()
# the last k_rx_merge closes a step; walk back to the sweep that opened its acquisition
merges = [i for i, r in enumerate(rows) if "k_rx_merge" in r[2]]
last = merges[len(merges) // 2]                            # (a step of the workload's own size: the bench ends on smaller calls)
first = max(i for i, r in enumerate(rows[:last]) if "k_fcch_sweep" in r[2])
first = max(i for i, r in enumerate(rows[:first]) if "k_fcch_sweep" in r[2])     # two sweeps per chain: the first
while first > 0 and rows[first - 1][2].startswith("copy") and rows[first][0] - rows[first - 1][1] < 50_000:
    first -= 1
t0 = rows[first][0]
prev_end = t0
for s, e, n in rows[first:last + 1]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  {n}")
    prev_end = e
