cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for st in 2 3 5 6 7 0; do
  GMR1_HIP_DBG_STOP=$st rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/ph_$st -- python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/ph_$st.log 2>&1
  echo "== stop $st"; python3 tools/pmc_summary.py gpurun_out/ph_$st k_rx4 | grep -E "VALU|SALU|INSTS_LDS|WAVE_CYCLES"
  grep -h "k_rx4" gpurun_out/ph_$st/*/*kernel_trace.csv | awk -F, '{d=$(NF-0); }END{}' 
done
