# Every number DESIGN.md quotes for a round, from one build in one gpurun call (GPU box, repo root): bash tools/final_measure.sh <tag>
# Default decoder mode = ACC (the library's default); the generic mode is taken beside it on the same box.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
t=$1
o=gpurun_out/$t
mkdir -p $o
# 4. SQ counters and per-phase times of the headline kernel (both layouts), SQ counters of the nt3 kernels
for lay in interleaved planar; do
  bash tools/pmc_rx4.sh ${t}_$lay --layout $lay > /dev/null 2>&1; cp gpurun_out/pmc_${t}_$lay.txt $o/pmc_sq_k_rx4_$lay.txt
  python3 tools/phase_times.py --no-extras --layout $lay > $o/phase_times_$lay.txt 2>&1
done
python3 tools/valu_summary.py $o/pmc_sq_k_rx4_interleaved.txt $t --out valu_k_rx4.json > $o/valu_k_rx4.log 2>&1     # -> profiles/valu_k_rx4.json
bash tools/pmc_nt3.sh ${t}_pmc_nt3 > /dev/null 2>&1; cp gpurun_out/${t}_pmc_nt3_kernels.txt $o/pmc_nt3_kernels.txt
python3 tools/valu_summary.py $o/pmc_nt3_kernels.txt $t --kernel k_rx4g_tch3 --waves-per-simd 7 --out valu_k_rx4g_tch3.json > $o/valu_k_rx4g_tch3.log 2>&1
bash tools/exp/pmc_any.sh tch3 k_tch3 > $o/pmc_tch3_kernels.txt 2>&1
python3 tools/valu_summary.py $o/pmc_tch3_kernels.txt $t --kernel "k_tch3<" --waves-per-simd 8 --out valu_k_tch3.json > $o/valu_k_tch3.log 2>&1
bash tools/exp/pmc_fcch_sweep.sh > $o/pmc_fcch_sweep.txt 2>&1
python3 tools/loop_stamps.py > $o/loop_stamps.txt 2>&1
python3 tools/time_legacy.py > $o/legacy_one_burst_calls.json 2> $o/legacy.err
echo "pmc done"
timeout -k 10 900 python3 -m pytest tests -q -m gpu > $o/gpu_tests.log 2>&1; tail -3 $o/gpu_tests.log
# 5. the N > 1 paths rehearsed on the one GPU: two gloo ranks (headline with the sharded receive loop, configs[4] both ways), RCCL with one rank
GMR1_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 20 --warmup 5 --shard-arfcns 16 --shard-seconds 20 > $o/bench_gloo2_one_gpu.json 2> $o/bench_gloo2.err
GMR1_BENCH_BACKEND=gloo python3 bench.py --workload nt3 --gpus 2 --steps 20 --warmup 5 --no-cpu > $o/bench_nt3_gloo2_one_gpu.json 2> $o/bench_nt3_gloo2.err
GMR1_BENCH_BACKEND=gloo python3 bench.py --workload tch3 --gpus 2 --steps 20 --warmup 5 --no-cpu > $o/bench_tch3_gloo2_one_gpu.json 2> $o/bench_tch3_gloo2.err
GMR1_BENCH_FORCE_GROUP=1 python3 bench.py --no-cpu --steps 20 --shard-arfcns 16 --shard-seconds 20 > $o/bench_rccl_one_rank.json 2> $o/bench_rccl1.err
rm -rf gpurun_out/traffic_${t}_* gpurun_out/pmc_${t}_* gpurun_out/pmc2_${t}_*
ls -la $o | head -80
