# SQ counters of the kernels of one bench workload (GPU box, repo root): bash tools/exp/pmc_any.sh <workload> <kernel name fragment> [bench.py arguments]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
w=$1; k=$2; shift; shift
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_$w -- python3 bench.py --workload $w --no-cpu --steps 3 --warmup 1 "$@" > gpurun_out/pmc_$w.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_$w $k
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d gpurun_out/pmc2_$w -- python3 bench.py --workload $w --no-cpu --steps 3 --warmup 1 "$@" > gpurun_out/pmc2_$w.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc2_$w $k
