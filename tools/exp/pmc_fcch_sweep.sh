# round 5: SQ counters, MFMA counters and fabric traffic of the one-pass FCCH sweep's kernels (GPU box, repo root)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { d=gpurun_out/pmcf_$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d -- python3 bench.py --workload fcch --no-cpu --steps 5 --warmup 2 > $d.log 2>&1; for k in k_fcch_sweep k_fcch_energy; do python3 tools/pmc_summary.py $d $k; done; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
run b SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES
run c FETCH_SIZE
run d WRITE_SIZE
