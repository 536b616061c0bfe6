cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_chan -- python3 bench.py --workload chan --no-cpu --steps 5 --warmup 2 > gpurun_out/pmc_chan.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_chan k_resamp
python3 tools/pmc_summary.py gpurun_out/pmc_chan k_pfb64
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_chan2 -- python3 bench.py --workload chan --no-cpu --steps 5 --warmup 2 > gpurun_out/pmc_chan2.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_chan2 k_resamp
python3 tools/pmc_summary.py gpurun_out/pmc_chan2 k_pfb64
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st_chan -- python3 bench.py --workload chan --no-cpu --steps 10 > /dev/null 2>&1
cut -d, -f1-4 $(ls gpurun_out/st_chan/*/*kernel_stats.csv | head -1) | head -4
