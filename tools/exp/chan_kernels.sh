#!/bin/bash
# Kernel durations of the channelizer step (rocprofv3 --stats): product build, and the profiling build with one output per
# lane in the resampler (GMR1_HIP_RESAMP_R=1).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/chan_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/chan_prof -o r2 -- python3 $R/bench.py --workload chan --steps 20 --warmup 3 --no-cpu --no-extras > $R/gpurun_out/chan_prof/r2.json 2>/dev/null
export GMR1_HIP_LIBRARY=$R/osmo-gmr_amd/libgmr1_hip_prof.so
export GMR1_HIP_RESAMP_R=1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/chan_prof -o r1 -- python3 $R/bench.py --workload chan --steps 20 --warmup 3 --no-cpu --no-extras > $R/gpurun_out/chan_prof/r1.json 2>/dev/null
for f in r2 r1; do echo "== $f"; cut -c1-140 $R/gpurun_out/chan_prof/${f}_kernel_stats.csv | head -4; done
