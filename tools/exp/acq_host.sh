#!/bin/bash
# Host-side stamps of the acquisition (profiling build, GMR1_HIP_RX_TIMING): where its wall time goes; both forms of the
# chain (glue by the sweeps' last threads / one launch per step).
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export GMR1_HIP_LIBRARY=$GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip_prof.so
[ -f $GMR1_HIP_LIBRARY ] || python3 osmo-gmr_amd/build.py --profile > /dev/null
export GMR1_HIP_RX_TIMING=1
for mode in fused; do
	if [ $mode = unfused ]; then export GMR1_HIP_ACQ_UNFUSED=1; else unset GMR1_HIP_ACQ_UNFUSED; fi
	timeout -k 10 300 python3 bench.py --workload rx ${ACQ_ARGS} --steps 20 --warmup 3 --no-cpu --no-extras > gpurun_out/acq_host_$mode.json 2> gpurun_out/acq_host_$mode.err
	echo "== $mode"
	grep -E "^acquire|^frame loop" gpurun_out/acq_host_$mode.err | tail -6
	python3 -c "
import json,sys
d=json.loads(open('gpurun_out/acq_host_$mode.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['phases_ms'])"
done
