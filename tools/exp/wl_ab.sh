#!/bin/bash
# Any bench workload A/B on one box: WL=fcch LIBS="product pl" TAG=x tools/exp/wl_ab.sh  (libraries named after libgmr1_hip_, alternating)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-wl_ab}.txt
: > $OUT
lib_path() { if [ $1 = product ]; then echo $GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip.so; else echo $GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip_$1.so; fi; }
for rep in 1 2 3; do for lib in ${LIBS:-product}; do
	export GMR1_HIP_LIBRARY=$(lib_path $lib)
	timeout -k 10 300 python3 bench.py --workload ${WL:-fcch} --no-cpu ${WL_ARGS} > gpurun_out/wab_$lib.json 2>gpurun_out/wab_$lib.err || { echo "$lib failed" >> $OUT; tail -3 gpurun_out/wab_$lib.err >> $OUT; cat $OUT; exit 1; }
	python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/wab_$lib.json').read().splitlines() if l.startswith('{')][0]); print('$lib', round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d.get('checks'))" >> $OUT
done; done
cat $OUT
