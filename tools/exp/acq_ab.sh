#!/bin/bash
# Acquisition chain of the receive loop: glue done by the sweeps' last threads (product) vs one launch per step
# (profiling build, GMR1_HIP_ACQ_UNFUSED=1).  Prints phases_ms of the rx bench line for both.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export GMR1_HIP_LIBRARY=$GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip_prof.so
[ -f $GMR1_HIP_LIBRARY ] || python3 osmo-gmr_amd/build.py --profile > /dev/null
for mode in fused unfused; do
	if [ $mode = unfused ]; then export GMR1_HIP_ACQ_UNFUSED=1; else unset GMR1_HIP_ACQ_UNFUSED; fi
	timeout -k 10 300 python3 bench.py --workload rx --steps 30 --warmup 5 --no-cpu --no-extras > gpurun_out/acq_$mode.json
	python3 - <<PY
import json
d = json.loads(open("gpurun_out/acq_$mode.json").read().strip().splitlines()[-1])
print("$mode", d["ms_per_step"], d.get("phases_ms"), d.get("checks", {}).get("identical_to_oracle"))
PY
done
