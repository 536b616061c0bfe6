# headline kernel at 6 / 5 / 4 waves per SIMD (libgmr1_hip_prof / _w5 / _w4: build.py --variant wN -DGMR1_EXP_RX4_WAVES=N), whole
# kernel (stop 0) and without pass 2's second read (stop 100: wrong results, the ceiling of a build that never re-reads)
cd $GRAFT_REPO_ROOT
run() { GMR1_HIP_LIBRARY=$GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip_$1.so GMR1_HIP_DBG_STOP=$2 python3 bench.py --no-cpu --no-extras --steps 100 $3 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('$1 stop $2 $3: %.4f ms  %.1f %%  crc_pass %.4f  missed %s' % (d['roofline']['kernel_ms'], 100*d['roofline']['frac'], d['checks']['crc_pass_frac'], d['checks'].get('mis_speculated_picks_per_launch')))"; }
for r in 1 2; do
for v in ${VARIANTS:-prof w5 w4}; do
  for st in ${STOPS:-0 100}; do run $v $st; done
done
done
