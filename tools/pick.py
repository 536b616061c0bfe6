#!/usr/bin/env python3
"""Print ms_per_step / roofline.frac / kernel_ms of bench.py JSON lines: tools/pick.py file.json ..."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        r = d.get("roofline") or {}
        print(f, "value", round(d["value"], 2), d["unit"], "ms_per_step", round(d["ms_per_step"], 4), "frac", round(r.get("frac", 0), 4),
              "kernel_ms", r.get("kernel_ms"))
    except Exception as e:
        print(f, "unreadable:", e)
