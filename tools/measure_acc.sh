# Every layer-1 workload in BOTH Viterbi decoder modes on ONE box (boxes of the pool differ by a few per cent, so the two
# modes are only comparable within one call): bash tools/measure_acc.sh <tag>   (GPU box, repo root)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
t=$1
o=gpurun_out/$t
mkdir -p $o
for m in generic acc; do
  python3 bench.py --conv-decoder $m > $o/${m}_bench_100k.json 2> $o/${m}_bench_100k.err
  for w in nt3 tch3 rx; do python3 bench.py --workload $w --conv-decoder $m > $o/${m}_bench_$w.json 2> $o/${m}_bench_$w.err; done
  echo "$m benches done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_bench -- python3 bench.py --no-cpu --conv-decoder acc > /dev/null 2> $o/stats_bench.err
cp $(ls $o/stats_bench/*/*kernel_stats.csv | head -1) $o/acc_kernel_stats_bench100k.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_nt3 -- python3 bench.py --workload nt3 --no-cpu --steps 20 --conv-decoder acc > /dev/null 2> $o/stats_nt3.err
cp $(ls $o/stats_nt3/*/*kernel_stats.csv | head -1) $o/acc_kernel_stats_nt3_1M.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_rx -- python3 bench.py --workload rx --no-cpu --no-shard --steps 10 --conv-decoder acc > /dev/null 2> $o/stats_rx.err
cp $(ls $o/stats_rx/*/*kernel_stats.csv | head -1) $o/acc_kernel_stats_rx_64x60s.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_tch3 -- python3 bench.py --workload tch3 --no-cpu --steps 20 --conv-decoder acc > /dev/null 2> $o/stats_tch3.err
cp $(ls $o/stats_tch3/*/*kernel_stats.csv | head -1) $o/acc_kernel_stats_tch3_1M.csv
rm -rf $o/stats_bench $o/stats_nt3 $o/stats_rx $o/stats_tch3
for m in generic acc; do for f in bench_100k bench_nt3 bench_tch3 bench_rx; do echo "== $m $f"; python3 -c "
import json
d=json.load(open('$o/${m}_$f.json'))
print(d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'roofline', (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('kernel_ms'), d.get('checks'))
"; done; done
