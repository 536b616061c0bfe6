# SQ counters of the headline kernel (run on the GPU box from the repo root): tools/pmc_rx4.sh <tag> [extra bench.py arguments, e.g. --layout planar]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; shift
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --no-cpu --no-extras --preroll-s 0.05 --steps 20 "$@" > gpurun_out/pmc_$tag.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_$tag k_rx4 > gpurun_out/pmc_$tag.txt
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc2_$tag -- python3 bench.py --no-cpu --no-extras --preroll-s 0.05 --steps 20 "$@" > gpurun_out/pmc2_$tag.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc2_$tag k_rx4 >> gpurun_out/pmc_$tag.txt
cat gpurun_out/pmc_$tag.txt
