#!/bin/bash
# Timeline of one acquisition chain of the receive loop: kernel and copy begin / end times from rocprofv3's traces.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/acq_tl
export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/acq_tl -o tl -- python3 bench.py --workload rx ${ACQ_ARGS} --steps 6 --warmup 3 --no-cpu --no-extras > gpurun_out/acq_tl/bench.json 2> gpurun_out/acq_tl/err.log
python3 tools/exp/acq_timeline.py gpurun_out/acq_tl
