#!/bin/bash
# Receive loop A/B on one box: libraries given in LIBS (names after libgmr1_hip_, "product" = libgmr1_hip.so), alternating;
# then the stamped walk of the profiling build.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-loop_ab}.txt
: > $OUT
for rep in 1 2 3; do for lib in ${LIBS:-l0 product}; do
	if [ $lib = product ]; then export GMR1_HIP_LIBRARY=$GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip.so; else export GMR1_HIP_LIBRARY=$GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip_$lib.so; fi
	timeout -k 10 200 python3 bench.py --workload rx ${RX_ARGS} --no-cpu --no-extras > gpurun_out/lab_$lib.json 2>gpurun_out/lab_$lib.err || { echo "$lib failed" >> $OUT; tail -3 gpurun_out/lab_$lib.err >> $OUT; exit 1; }
	python3 -c "
import json
d=json.loads(open('gpurun_out/lab_$lib.json').read().strip().splitlines()[-1]); print('$lib', round(d['ms_per_step'],4), d['phases_ms'])" >> $OUT
done; done
unset GMR1_HIP_LIBRARY
if [ -z "$NO_STAMPS" ]; then timeout -k 10 200 python3 tools/loop_stamps.py >> $OUT 2>&1; fi
cat $OUT
