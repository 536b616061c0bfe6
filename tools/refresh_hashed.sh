# Re-take what bench.py reads hash-gated (run on the GPU box from the repo root after the last change to the kernel sources):
# fabric traffic of the headline kernel and of the nt3 step, the headline kernel's SQ counters.   bash tools/refresh_hashed.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
t=$1
o=gpurun_out/$t
mkdir -p $o
bash tools/measure_traffic.sh $t > $o/traffic.log 2>&1
cp gpurun_out/hbm_traffic.json $o/hbm_traffic.json; cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json
bash tools/traffic_nt3.sh $t > $o/traffic_nt3.log 2>&1
cp gpurun_out/hbm_traffic_nt3.json $o/hbm_traffic_nt3.json; cp gpurun_out/hbm_traffic_nt3.json profiles/hbm_traffic_nt3.json
bash tools/pmc_rx4.sh ${t}_interleaved --layout interleaved > /dev/null 2>&1; cp gpurun_out/pmc_${t}_interleaved.txt $o/pmc_sq_k_rx4_interleaved.txt
python3 tools/valu_summary.py $o/pmc_sq_k_rx4_interleaved.txt $t > $o/valu_k_rx4.log 2>&1
cp profiles/valu_k_rx4.json $o/valu_k_rx4.json
rm -rf gpurun_out/traffic_${t}_* gpurun_out/pmc_${t}_* gpurun_out/pmc2_${t}_*
ls $o
