# SQ counters of the FCCH sweep's kernels (GPU box, repo root)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_fcch -- python3 bench.py --workload fcch --no-cpu --steps 5 --warmup 2 > gpurun_out/pmc_fcch.log 2>&1
for k in k_fcch_corr k_fcch_stats k_fcch_fine; do python3 tools/pmc_summary.py gpurun_out/pmc_fcch $k; done
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d gpurun_out/pmc_fcch2 -- python3 bench.py --workload fcch --no-cpu --steps 5 --warmup 2 > gpurun_out/pmc_fcch2.log 2>&1
for k in k_fcch_corr k_fcch_stats k_fcch_fine; do python3 tools/pmc_summary.py gpurun_out/pmc_fcch2 $k; done
