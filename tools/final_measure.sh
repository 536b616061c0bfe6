# Every number DESIGN.md quotes for a round, from one build (GPU box, repo root), in two gpurun calls of at most 20 minutes each:
#   bash tools/final_measure.sh <tag>     traffic, bench lines, kernel statistics
#   bash tools/final_measure_b.sh <tag>   SQ counters, phase times, the -m gpu suite, the N > 1 paths rehearsed on one GPU
# Default decoder mode = ACC (the library's default); the generic mode is taken beside it on the same box.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
t=$1
o=gpurun_out/$t
mkdir -p $o
# 1. fabric traffic first (PMC passes), so that the bench lines below carry it: headline kernel in both layouts, then the nt3 step
bash tools/measure_traffic.sh $t > $o/traffic.log 2>&1
cp gpurun_out/hbm_traffic.json $o/hbm_traffic.json; cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json
bash tools/traffic_nt3.sh $t > $o/traffic_nt3.log 2>&1
cp gpurun_out/hbm_traffic_nt3.json $o/hbm_traffic_nt3.json; cp gpurun_out/hbm_traffic_nt3.json profiles/hbm_traffic_nt3.json
echo "traffic done"
# 2. the bench lines
python3 bench.py > $o/bench_100k.json 2> $o/bench_100k.err
python3 bench.py --conv-decoder generic > $o/bench_100k_generic.json 2> $o/bench_100k_generic.err
for w in nt3 tch3 fcch rx chan ambe; do python3 bench.py --workload $w > $o/bench_$w.json 2> $o/bench_$w.err; done
for w in nt3 tch3 rx; do python3 bench.py --workload $w --conv-decoder generic --no-cpu > $o/bench_${w}_generic.json 2> $o/bench_${w}_generic.err; done
python3 bench.py --workload rx --arfcns 512 --seconds 20 --no-cpu > $o/bench_rx_512x20s.json 2> $o/bench_rx_512.err
python3 bench.py --workload nt3 --no-cpu --nt3-two-launches > $o/bench_nt3_two_launches.json 2> $o/bench_nt3_two.err
echo "benches done"
# 3. rocprofv3 kernel statistics of the same commands
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_bench -- python3 bench.py --no-cpu > $o/bench_100k_under_profiler.json 2> $o/stats_bench.err
cp $(ls $o/stats_bench/*/*kernel_stats.csv | head -1) $o/kernel_stats_bench100k.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_nt3 -- python3 bench.py --workload nt3 --no-cpu --steps 20 > /dev/null 2> $o/stats_nt3.err
cp $(ls $o/stats_nt3/*/*kernel_stats.csv | head -1) $o/kernel_stats_nt3_1M.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_rx -- python3 bench.py --workload rx --no-cpu --no-shard --steps 10 > /dev/null 2> $o/stats_rx.err
cp $(ls $o/stats_rx/*/*kernel_stats.csv | head -1) $o/kernel_stats_rx_64x60s.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_fcch -- python3 bench.py --workload fcch --no-cpu --steps 20 > /dev/null 2> $o/stats_fcch.err
cp $(ls $o/stats_fcch/*/*kernel_stats.csv | head -1) $o/kernel_stats_fcch.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_chan -- python3 bench.py --workload chan --no-cpu --steps 20 > /dev/null 2> $o/stats_chan.err
cp $(ls $o/stats_chan/*/*kernel_stats.csv | head -1) $o/kernel_stats_chan.csv
rm -rf $o/stats_bench $o/stats_nt3 $o/stats_rx $o/stats_fcch $o/stats_chan
echo "stats done"
ls -la $o | head -60
for f in bench_100k bench_100k_generic bench_nt3 bench_nt3_generic bench_tch3 bench_tch3_generic bench_fcch bench_rx bench_rx_generic bench_chan bench_ambe bench_rx_512x20s bench_nt3_two_launches; do echo "== $f"; python3 -c "
import json,sys
d=json.loads([l for l in open('$o/$f.json').read().splitlines() if l.startswith('{')][0])
r=d.get('roofline') or {}
print(d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'roofline', r.get('frac'), r.get('kernel_ms'), 'traffic', r.get('traffic'))
if 'roofline_planar' in d: print('  planar', d['roofline_planar']['kernel_ms'], d['roofline_planar']['frac'], d['roofline_planar']['traffic'], d['roofline_planar']['outputs_bit_identical_to_interleaved']); print('  other', d['other_decoder'])
if 'roofline_valu' in d: print('  valu', {k: v for k, v in d['roofline_valu'].items() if k in ('achieved', 'frac', 'valu_busy', 'hbm_frac_if_valu_were_100pct_busy')})
"; done
