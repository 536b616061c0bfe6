#!/bin/bash
# SQ counters of the configs[4]-from-samples kernels (GPU box, repo root): tools/pmc_nt3.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
t=${1:-pmc_nt3}
o=gpurun_out/$t
rm -rf $o; mkdir -p $o
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $o/a -- python3 bench.py --workload nt3 --steps 3 --warmup 1 --no-cpu --preroll-s 0 > $o/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES --kernel-trace --output-format csv -d $o/b -- python3 bench.py --workload nt3 --steps 3 --warmup 1 --no-cpu --preroll-s 0 > $o/b.log 2>&1
{ for k in k_rx4g_tch3 "k_rx4g<" k_facch3; do python3 tools/pmc_summary.py $o/a "$k"; python3 tools/pmc_summary.py $o/b "$k"; done; } > gpurun_out/${t}_kernels.txt 2>&1
rm -rf $o
cat gpurun_out/${t}_kernels.txt | head -60
