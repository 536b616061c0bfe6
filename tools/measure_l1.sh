# configs[4] workloads (layer 1 only, and from samples) with kernel stats and SQ counters (GPU box, repo root): bash tools/measure_l1.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
t=$1
o=gpurun_out/$t
mkdir -p $o
python3 bench.py --workload tch3 > $o/bench_tch3.json 2> $o/bench_tch3.err
python3 bench.py --workload nt3 > $o/bench_nt3.json 2> $o/bench_nt3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_nt3 -- python3 bench.py --workload nt3 --no-cpu --steps 20 > /dev/null 2> $o/stats_nt3.err
cp $(ls $o/stats_nt3/*/*kernel_stats.csv | head -1) $o/kernel_stats_nt3_1M.csv
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $o/pmc_nt3 -- python3 bench.py --workload nt3 --steps 3 --warmup 1 --no-cpu --preroll-s 0 > $o/pmc_nt3.log 2>&1
{ python3 tools/pmc_summary.py $o/pmc_nt3 k_rx4g; python3 tools/pmc_summary.py $o/pmc_nt3 k_tch3; python3 tools/pmc_summary.py $o/pmc_nt3 "k_rx<"; python3 tools/pmc_summary.py $o/pmc_nt3 k_facch3; } > $o/pmc_nt3_kernels.txt
rm -rf $o/stats_nt3 $o/pmc_nt3
for f in bench_tch3 bench_nt3; do python3 -c "
import json; d=json.load(open('$o/$f.json')); print('$f', d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d.get('checks'))"; done
head -6 $o/kernel_stats_nt3_1M.csv | cut -c1-130; grep -A4 "k_tch3" $o/pmc_nt3_kernels.txt | head -8
