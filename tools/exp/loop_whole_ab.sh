#!/bin/bash
# Receive loop: the whole walk in one launch (flag-triggered CCCH batches) against one launch per time slice
# (profiling build, GMR1_HIP_LOOP_SLICED=1), same box, alternating.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export GMR1_HIP_LIBRARY=$GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip_prof.so
[ -f $GMR1_HIP_LIBRARY ] || python3 osmo-gmr_amd/build.py --profile > /dev/null
for rep in 1 2 3; do for mode in whole sliced; do
	if [ $mode = sliced ]; then export GMR1_HIP_LOOP_SLICED=1; else unset GMR1_HIP_LOOP_SLICED; fi
	python3 bench.py --workload rx ${RX_ARGS} --no-cpu --no-extras > gpurun_out/lw_$mode.json 2>/dev/null
	python3 -c "
import json
d=json.loads(open('gpurun_out/lw_$mode.json').read().strip().splitlines()[-1]); print('$mode', round(d['ms_per_step'],4), d['phases_ms']['chain_ms'])"
done; done
