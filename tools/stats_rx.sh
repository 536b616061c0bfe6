#!/bin/bash
# kernel statistics of the receive-loop bench (run on the GPU box from the repo root): tools/stats_rx.sh [tag] [extra bench flags]
tag=${1:-rx}; shift
o=gpurun_out/stats_$tag
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o -- python3 bench.py --workload rx --no-cpu --steps 10 "$@" > $o/bench.json 2> $o/err.log
f=$(find $o -name '*kernel_stats.csv' | tail -1)
if [ -n "$f" ]; then cp $f gpurun_out/kernel_stats_$tag.csv; cut -c1-180 gpurun_out/kernel_stats_$tag.csv | sed -n 1,14p; else echo "no kernel_stats.csv"; tail -5 $o/err.log; fi
