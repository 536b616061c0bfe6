#!/bin/bash
# Channelizer A/B on one box: libraries in LIBS (names after libgmr1_hip_, "product" = libgmr1_hip.so), alternating; the
# step's time from bench.py and the kernels' from rocprofv3 --stats (last library of LIBS and the first).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-chan_ab}.txt
: > $OUT
lib_path() { if [ $1 = product ]; then echo $GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip.so; else echo $GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip_$1.so; fi; }
for rep in 1 2 3; do for lib in ${LIBS:-product c0}; do
	export GMR1_HIP_LIBRARY=$(lib_path $lib)
	timeout -k 10 200 python3 bench.py --workload chan --no-cpu --no-extras > gpurun_out/cab_$lib.json 2>gpurun_out/cab_$lib.err || { echo "$lib failed" >> $OUT; tail -3 gpurun_out/cab_$lib.err >> $OUT; cat $OUT; exit 1; }
	python3 -c "
import json
d=json.loads(open('gpurun_out/cab_$lib.json').read().strip().splitlines()[-1]); print('$lib', round(d['ms_per_step'],4), round(d['roofline']['frac'],4))" >> $OUT
done; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in ${LIBS:-product c0}; do
	export GMR1_HIP_LIBRARY=$(lib_path $lib)
	mkdir -p $R/gpurun_out/chan_prof
	timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/chan_prof -o $lib -- python3 $R/bench.py --workload chan --steps 20 --warmup 3 --no-cpu --no-extras > $R/gpurun_out/chan_prof/$lib.json 2>/dev/null || { echo "rocprof $lib failed" >> $OUT; cat $OUT; exit 1; }
	echo "== $lib" >> $OUT; cut -c1-140 $R/gpurun_out/chan_prof/${lib}_kernel_stats.csv | head -4 >> $OUT
done
cat $OUT
