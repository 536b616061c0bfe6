# AMBE decoder measurement set (GPU box, repo root): bash tools/measure_ambe.sh <tag>  -> gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# the switches below exist only in the profiling build (python osmo-gmr_amd/build.py --profile)
export GMR1_HIP_LIBRARY=$GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip_prof.so
[ -f $GMR1_HIP_LIBRARY ] || python3 osmo-gmr_amd/build.py --profile > /dev/null
t=$1
o=gpurun_out/$t
mkdir -p $o
python3 bench.py --workload ambe > $o/bench_ambe.json 2> $o/bench_ambe.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python3 bench.py --workload ambe --no-cpu --steps 20 > /dev/null 2> $o/stats.err
cp $(ls $o/stats/*/*kernel_stats.csv | head -1) $o/kernel_stats_ambe.csv
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $o/pmc_a -- python3 bench.py --workload ambe --steps 3 --warmup 1 --no-cpu --preroll-s 0 > $o/pmc_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $o/pmc_b -- python3 bench.py --workload ambe --steps 3 --warmup 1 --no-cpu --preroll-s 0 > $o/pmc_b.log 2>&1
{ python3 tools/pmc_summary.py $o/pmc_a "k_ambe("; python3 tools/pmc_summary.py $o/pmc_b "k_ambe("; } > $o/pmc_k_ambe.txt
: > $o/phase_times_ambe.txt
for d in 0 1 2 4 3 7; do GMR1_HIP_AMBE_DBG=$d python3 bench.py --workload ambe --steps 10 --warmup 3 --no-cpu | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('GMR1_HIP_AMBE_DBG=$d (1: no noise path, 2: no oscillator bank, 4: no parameter decode):', round(d['roofline']['kernel_ms'],2), 'ms')" >> $o/phase_times_ambe.txt; done
rm -rf $o/stats $o/pmc_a $o/pmc_b
cat $o/phase_times_ambe.txt $o/pmc_k_ambe.txt; cat $o/kernel_stats_ambe.csv | cut -c1-150 | head -5; cat $o/bench_ambe.json
