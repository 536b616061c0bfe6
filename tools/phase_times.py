#!/usr/bin/env python3
"""Kernel time of the headline kernel cut off after each phase (GMR1_HIP_DBG_STOP), one bench.py child per cut:
2 = pass 1 (load, statistics, correlation), 3 = + peak / timing, 5 = + sync terms, 6 = + pass 2 (soft bits),
7 = + branch metrics, 0 = everything, 100 = everything but pass 2's re-read of the window (wrong results:
what the second read costs).  Run on the GPU box from the repo root."""
import json, os, subprocess, sys
prev = 0.0
for st in (2, 3, 5, 6, 7, 0, 100):
    # the cut-offs exist only in the profiling build: python osmo-gmr_amd/build.py --profile
    prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "osmo-gmr_amd", "libgmr1_hip_prof.so")
    env = dict(os.environ, GMR1_HIP_DBG_STOP=str(st), GMR1_HIP_LIBRARY=os.environ.get("GMR1_HIP_LIBRARY", prof))
    r = subprocess.run([sys.executable, "bench.py", "--no-cpu", "--steps", "60", "--warmup", "5"] + sys.argv[1:],
                       capture_output=True, text=True, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(st, "failed", r.stderr[-500:]); continue
    ms = json.loads(line[0])["roofline"]["kernel_ms"]
    print(f"stop {st}: {ms:.4f} ms  (+{ms - prev:.4f})", flush=True)
    prev = ms
