# instruction-cache counters of the headline kernel (product and profiling build): does a 50 KB kernel live in the 64 KB cache two CUs share?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in libgmr1_hip.so libgmr1_hip_prof.so libgmr1_hip_base.so; do
  echo "== $lib"
  GMR1_HIP_LIBRARY=$GRAFT_REPO_ROOT/osmo-gmr_amd/$lib rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_ic_$lib -- python3 bench.py --no-cpu --no-extras --preroll-s 0.05 --steps 10 > gpurun_out/pmc_ic_$lib.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_ic_$lib "k_rx4<"
  rm -rf gpurun_out/pmc_ic_$lib
done
