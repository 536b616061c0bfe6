# VALU / SALU / LDS instructions per wave of the NT3 speech demodulator (k_rx4g<8,4>) cut off after each phase
# (GPU box, repo root; needs a library built with -DGMR1_HIP_PROFILE):  bash tools/phases_nt3_pmc.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# the switches below exist only in the profiling build (python osmo-gmr_amd/build.py --profile)
export GMR1_HIP_LIBRARY=$GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip_prof.so
[ -f $GMR1_HIP_LIBRARY ] || python3 osmo-gmr_amd/build.py --profile > /dev/null
tag=$1
out=gpurun_out/phases_nt3_$tag.txt
: > $out
for st in 2 3 5 0; do
  GMR1_HIP_DBG_STOP=$st rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/phn_${tag}_$st -- python3 tools/phases_nt3.py $st 200000 > gpurun_out/phn_${tag}_$st.log 2>&1
  echo "== stop $st" >> $out
  python3 tools/pmc_summary.py gpurun_out/phn_${tag}_$st k_rx4g | grep -E "INSTS|CYCLES|grid" >> $out
  rm -rf gpurun_out/phn_${tag}_$st
done
cat $out
