# VALU / SALU / LDS instructions of k_ambe with phases cut off (GPU box, repo root): bash tools/phases_ambe.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# the switches below exist only in the profiling build (python osmo-gmr_amd/build.py --profile)
export GMR1_HIP_LIBRARY=$GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip_prof.so
[ -f $GMR1_HIP_LIBRARY ] || python3 osmo-gmr_amd/build.py --profile > /dev/null
tag=$1
out=gpurun_out/phases_ambe_$tag.txt
: > $out
for d in 0 1 2 4 7; do
  GMR1_HIP_AMBE_DBG=$d rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/pha_${tag}_$d -- python3 bench.py --workload ambe --steps 2 --warmup 1 --no-cpu --preroll-s 0 > gpurun_out/pha_${tag}_$d.log 2>&1
  echo "== GMR1_HIP_AMBE_DBG=$d (1: no noise path, 2: no oscillator bank, 4: no parameter decode)" >> $out
  python3 tools/pmc_summary.py gpurun_out/pha_${tag}_$d "k_ambe(" | grep -E "INSTS|CYCLES" >> $out
  rm -rf gpurun_out/pha_${tag}_$d
done
cat $out
