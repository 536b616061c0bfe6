# The headline kernel in both sample layouts and both decoder modes on one box: fabric traffic (PMC), SQ instruction
# counters, per-phase times (profiling build).   bash tools/planar_measure.sh <tag>   (GPU box, repo root)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
t=$1
o=gpurun_out/$t
mkdir -p $o
python3 bench.py --no-cpu > $o/bench_100k_nocpu.json 2> $o/bench.err
bash tools/measure_traffic.sh $t > $o/traffic.log 2>&1
cp gpurun_out/hbm_traffic.json $o/hbm_traffic.json
for lay in interleaved planar; do for m in acc generic; do
  bash tools/pmc_rx4.sh ${t}_${lay}_$m --layout $lay --conv-decoder $m > /dev/null 2>&1
  cp gpurun_out/pmc_${t}_${lay}_$m.txt $o/pmc_sq_${lay}_$m.txt
  python3 tools/phase_times.py --no-extras --layout $lay --conv-decoder $m > $o/phase_times_${lay}_$m.txt 2>&1
  echo "$lay $m done"
done; done
rm -rf gpurun_out/traffic_${t}_* gpurun_out/pmc_${t}_* gpurun_out/pmc2_${t}_*
cat $o/hbm_traffic.json; tail -n 8 $o/phase_times_*.txt
