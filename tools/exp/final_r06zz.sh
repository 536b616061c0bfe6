#!/bin/bash
# The round's last numbers behind the channelizer changes: the -m gpu suite, the default line, the channelizer's line and kernels.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r06zz; mkdir -p $o
timeout -k 10 1000 python3 -m pytest tests -q -m gpu > $o/gpu_tests.log 2>&1; tail -3 $o/gpu_tests.log
python3 bench.py > $o/bench_default.json 2> $o/bench_default.err
python3 bench.py --workload chan > $o/bench_chan.json 2> $o/bench_chan.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_chan -- python3 bench.py --workload chan --no-cpu --steps 20 > /dev/null 2> $o/stats_chan.err
cp $(ls $o/stats_chan/*/*kernel_stats.csv | head -1) $o/kernel_stats_chan.csv; rm -rf $o/stats_chan
python3 - <<'PY'
import json
for f in ("bench_default", "bench_chan"):
    d = json.loads([l for l in open(f"gpurun_out/r06zz/{f}.json").read().splitlines() if l.startswith("{")][0])
    print(f, d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"])
    for k, v in (d.get("side") or {}).items():
        if isinstance(v, dict): print("  ", k, v.get("ms"), v.get("frac"))
PY
head -4 $o/kernel_stats_chan.csv | cut -c1-150
