# VALU / SALU / LDS instructions and wave cycles of the headline kernel cut off after each phase (GPU box, repo root):
#   bash tools/phases.sh <tag> [bench.py arguments, e.g. --layout planar]     -> gpurun_out/phases_<tag>.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# the switches below exist only in the profiling build (python osmo-gmr_amd/build.py --profile)
export GMR1_HIP_LIBRARY=$GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip_prof.so
[ -f $GMR1_HIP_LIBRARY ] || python3 osmo-gmr_amd/build.py --profile > /dev/null
tag=$1; shift
out=gpurun_out/phases_$tag.txt
: > $out
for st in 2 3 5 6 7 0; do
  GMR1_HIP_DBG_STOP=$st rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/ph_${tag}_$st -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras --preroll-s 0 "$@" > gpurun_out/ph_${tag}_$st.log 2>&1
  echo "== stop $st" >> $out
  python3 tools/pmc_summary.py gpurun_out/ph_${tag}_$st k_rx4 | grep -E "INSTS|CYCLES|WAIT" >> $out
done
cat $out
