# Every number DESIGN.md quotes for a round, from one build in one gpurun call (GPU box, repo root): bash tools/final_measure.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
t=$1
o=gpurun_out/$t
mkdir -p $o
python3 bench.py > $o/bench_100k.json 2> $o/bench_100k.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_bench -- python3 bench.py --no-cpu > $o/bench_100k_under_profiler.json 2> $o/stats_bench.err
cp $(ls $o/stats_bench/*/*kernel_stats.csv | head -1) $o/kernel_stats_bench100k.csv
bash tools/measure_traffic.sh $t > $o/traffic.log 2>&1
cp gpurun_out/hbm_traffic.json $o/hbm_traffic.json
bash tools/pmc_rx4.sh $t > /dev/null 2>&1; cp gpurun_out/pmc_$t.txt $o/pmc_sq_k_rx4.txt
python3 tools/phase_times.py > $o/phase_times.txt 2>&1
for w in nt3 tch3 fcch rx chan ambe; do python3 bench.py --workload $w > $o/bench_$w.json 2> $o/bench_$w.err; done
python3 bench.py --workload rx --arfcns 512 --seconds 20 --no-cpu > $o/bench_rx_512x20s.json 2> $o/bench_rx_512.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_nt3 -- python3 bench.py --workload nt3 --no-cpu --steps 20 > /dev/null 2> $o/stats_nt3.err
cp $(ls $o/stats_nt3/*/*kernel_stats.csv | head -1) $o/kernel_stats_nt3_1M.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_rx -- python3 bench.py --workload rx --no-cpu --no-shard --steps 10 > /dev/null 2> $o/stats_rx.err
cp $(ls $o/stats_rx/*/*kernel_stats.csv | head -1) $o/kernel_stats_rx_64x60s.csv
python3 bench.py --workload nt3 --no-cpu --nt3-two-launches > $o/bench_nt3_two_launches.json 2> $o/bench_nt3_two.err
bash tools/pmc_nt3.sh ${t}_pmc_nt3 > /dev/null 2>&1; cp gpurun_out/${t}_pmc_nt3_kernels.txt $o/pmc_nt3_kernels.txt
python3 tools/loop_stamps.py > $o/loop_stamps.txt 2>&1
python3 tools/time_legacy.py > $o/legacy_one_burst_calls.json 2> $o/legacy.err
GMR1_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 20 --warmup 5 --shard-arfcns 16 --shard-seconds 20 > $o/bench_gloo2_one_gpu.json 2> $o/bench_gloo2.err
rm -rf $o/stats_bench $o/stats_nt3 $o/stats_rx
ls -la $o | head -40
for f in bench_100k bench_nt3 bench_tch3 bench_fcch bench_rx bench_chan bench_ambe bench_rx_512x20s; do echo "== $f"; python3 -c "
import json,sys
d=json.load(open('$o/$f.json'))
print(d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'roofline', (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('kernel_ms'))
"; done
